#!/bin/bash
# A/B of option key 24 (side lanes, batches <= 16; exact fp32 / bf16x3 <= 12) on the product library:  tools/lanes_key_ab.sh <tag>  -> gpurun_out/<tag>_lanes_key_ab.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; TAG=${1:-rXX}; O=$R/gpurun_out
cd $R
Q="--cpu-frames 0 --predict-calls 0 --no-split-mode --no-configs --steps 20 --warmup 5"
run() { python3 bench.py $Q "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],3), 'ms', round(j['value'],1), 'masks/s')"; }
{
for cfgl in "--batch 16" "--batch 12" "--batch 8" "--batch 4" "--batch 1" "--batch 16 --dtype f32-bf16x3" "--batch 16 --dtype f16" "--batch 4 --height 720 --width 1280 --instances 30" "--batch 8 --dtype f32-bf16x3" "--batch 8 --dtype f16 --height 1024 --width 1024" "--batch 4 --dtype f16 --height 1024 --width 1024" "--batch 8 --dtype f16"; do
  for rep in 1 2; do
    echo "== $cfgl, key 24 = 0 (one stream)"; run $cfgl --tuning 24=0
    echo "== $cfgl, default (lanes at batches <= 16, exact fp32 / bf16x3 <= 12)"; run $cfgl
  done
done
} > $O/${TAG}_lanes_key_ab.txt 2>&1
cat $O/${TAG}_lanes_key_ab.txt
