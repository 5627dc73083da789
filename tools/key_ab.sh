#!/bin/bash
# Same-box A/B of one launch-time tuning key on the bench lines: tools/key_ab.sh <out tag> <reps> <key> <value a> <value b>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; REPS=$2; KEY=$3; shift 3
Q="--cpu-frames 0 --predict-calls 0 --no-split-mode --no-configs"
ms() { python3 $R/bench.py "$@" $Q 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],3))"; }
cd $R
for rep in $(seq $REPS); do
for v in "$@"; do
  T="--tuning $KEY=$v"
  echo "$rep key$KEY=$v b16 $(ms --steps 20 --warmup 5 $T) b2 $(ms --steps 60 --warmup 10 --batch 2 $T) b1 $(ms --steps 100 --warmup 20 --batch 1 $T) 720p_b1 $(ms --steps 40 --warmup 10 --height 720 --width 1280 --instances 30 --batch 1 $T) 720p_b4 $(ms --steps 20 --warmup 5 --height 720 --width 1280 --instances 30 --batch 4 $T) x3 $(ms --steps 20 --warmup 5 --dtype f32-bf16x3 $T)" | tee -a gpurun_out/${TAG}_key_ab.txt
done; done
