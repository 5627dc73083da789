#!/bin/bash
# HBM traffic of the convolution family only (the two --pmc passes of tools/profile_round.sh) -> gpurun_out/<tag>_conv_hbm_traffic.json
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; TAG=${1:-rXX}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch -o p -- python3 $R/tools/layer_profile.py run --plan $O/${TAG}_plan1_f.json --tuning 24=0 --iters 1 > $O/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_pmc_write -o p -- python3 $R/tools/layer_profile.py run --plan $O/${TAG}_plan1_w.json --tuning 24=0 --iters 1 > $O/${TAG}_pmc_write.log 2>&1
python3 $R/tools/traffic_report.py $O/${TAG}_plan1_f.json $O/${TAG}_pmc_fetch/p_counter_collection.csv $O/${TAG}_pmc_write/p_counter_collection.csv $O/${TAG}_conv_hbm_traffic.json > $O/${TAG}_traffic.txt 2>&1
tail -3 $O/${TAG}_traffic.txt
rm -rf $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write
