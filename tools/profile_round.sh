#!/bin/bash
# Evidence set of a round, on the GPU box (gpurun): rocprofv3 kernel stats of the benchmark, per-layer tables, MFMA-busy
# counters and HBM traffic counters (separate --pmc passes, as MI355X_MICROARCH.md prescribes).
#   usage: tools/profile_round.sh <tag>      -> gpurun_out/<tag>_*
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-rXX}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_bench -o b -- python3 $R/bench.py --cpu-frames 0 --predict-calls 0 --no-split-mode --no-configs --steps 20 --warmup 5 > $O/${TAG}_bench_under_rocprof.json 2> $O/${TAG}_prof_bench.log
cp $O/${TAG}_prof_bench/b_kernel_stats.csv $O/${TAG}_bench_kernel_stats.csv
for DT in 0 3; do
  rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_prof_layers_$DT -o lay -- python3 $R/tools/layer_profile.py run --plan $O/${TAG}_plan_$DT.json --tuning 24=0 --compute-dtype $DT > $O/${TAG}_prof_layers_$DT.log 2>&1
  python3 $R/tools/layer_profile.py report --plan $O/${TAG}_plan_$DT.json --trace $O/${TAG}_prof_layers_$DT/lay_kernel_trace.csv > $O/${TAG}_conv_layers_dtype$DT.md
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $O/${TAG}_pmc_busy_$DT -o p -- python3 $R/tools/layer_profile.py run --plan $O/${TAG}_plan1_$DT.json --tuning 24=0 --iters 1 --compute-dtype $DT > $O/${TAG}_pmc_busy_$DT.log 2>&1
  python3 $R/tools/pmc_summary.py $O/${TAG}_plan1_$DT.json $O/${TAG}_pmc_busy_$DT/p_counter_collection.csv > $O/${TAG}_conv_mfma_busy_dtype$DT.md
done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch -o p -- python3 $R/tools/layer_profile.py run --plan $O/${TAG}_plan1_f.json --tuning 24=0 --iters 1 > $O/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write -o p -- python3 $R/tools/layer_profile.py run --plan $O/${TAG}_plan1_w.json --tuning 24=0 --iters 1 > $O/${TAG}_pmc_write.log 2>&1
python3 $R/tools/traffic_report.py $O/${TAG}_plan1_f.json $O/${TAG}_pmc_fetch/p_counter_collection.csv $O/${TAG}_pmc_write/p_counter_collection.csv $O/${TAG}_conv_hbm_traffic.json > $O/${TAG}_traffic.txt 2>&1
tail -3 $O/${TAG}_traffic.txt
# the fp16 data path (BASELINE configs[4] stand-in): per-layer table at 1024x1024 batch 8
rocprofv3 --kernel-trace --output-format csv -d $O/${TAG}_prof_layers_f16 -o lay -- python3 $R/tools/layer_profile.py run --plan $O/${TAG}_plan_f16.json --tuning 24=0 --compute-dtype 2 --height 1024 --width 1024 --batch 8 > $O/${TAG}_prof_layers_f16.log 2>&1
python3 $R/tools/layer_profile.py report --plan $O/${TAG}_plan_f16.json --trace $O/${TAG}_prof_layers_f16/lay_kernel_trace.csv > $O/${TAG}_conv_layers_f16_1024.md
