#!/bin/bash
# The bench lines quoted in DESIGN.md / profiles/README.md, on the GPU box (gpurun):  tools/bench_lines.sh <tag>  -> gpurun_out/<tag>_bench*.json
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-rXX}
O=$R/gpurun_out
cd $R
# usage: tools/bench_lines.sh <tag> [traffic]: with "traffic", profiles/conv_hbm_traffic.json is first replaced by gpurun_out/<tag>_conv_hbm_traffic.json
# (tools/profile_round.sh <tag> ran before), so that the default line carries roofline.traffic of the sources it runs on
if [ "${2:-}" = traffic ] && [ -f $O/${TAG}_conv_hbm_traffic.json ]; then cp $O/${TAG}_conv_hbm_traffic.json $R/profiles/conv_hbm_traffic.json; fi
run() { out=$1; shift; python3 bench.py "$@" 2> $O/${TAG}_$out.err | tail -1 > $O/${TAG}_$out.json; python3 - $O/${TAG}_$out.json <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print(sys.argv[1].split("/")[-1], round(j["value"], 1), j["unit"], round(j["ms_per_step"], 2), "ms", "frac", round(j["roofline"]["frac"], 3))
PY
}
run bench --steps 20 --warmup 5
Q="--cpu-frames 0 --predict-calls 0 --no-split-mode --no-configs"
run bench_bf16x3 --steps 20 --warmup 5 --dtype f32-bf16x3 $Q
run bench_hostio --steps 20 --warmup 5 --host-io $Q
run bench_b1 --steps 100 --warmup 20 --batch 1 $Q
run bench_f16_1024 --steps 20 --warmup 5 --dtype f16 --height 1024 --width 1024 --batch 8 $Q
run bench_f32_1024 --steps 10 --warmup 3 --height 1024 --width 1024 --batch 8 $Q
run bench_f16_b16 --steps 20 --warmup 5 --dtype f16 $Q
QUBER_DIST_BACKEND=gloo run bench_2rank_gloo_rehearsal --gpus 2 --steps 10 --warmup 3 $Q
for b in 1 2 4; do run config2_1280x720_b${b}_graph --steps 40 --warmup 10 --height 720 --width 1280 --instances 30 --batch $b --graph $Q; done
