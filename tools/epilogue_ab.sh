#!/bin/bash
# Same-box A/B of library variants built beside the product library (quber_amd/libquber_hip_<name>.so; "new" = the product library):
#   tools/epilogue_ab.sh <out tag> <reps> name1 name2 ...   -> gpurun_out/<tag>_ab.txt
#   ms per step: batch 16, batch 1, 1280x720 batch 1; with F16=1 also the fp16 data path at 1024x1024 batch 8 and 640x480 batch 16
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; REPS=$2; shift 2
Q="--cpu-frames 0 --predict-calls 0 --no-split-mode --no-configs"
ms() { python3 $R/bench.py "$@" $Q 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],3))"; }
cd $R
for rep in $(seq $REPS); do
for v in "$@"; do
  if [ $v = new ]; then unset QUBER_LIB; else export QUBER_LIB=$R/quber_amd/libquber_hip_$v.so; fi
  L="$rep $v b16 $(ms --steps 20 --warmup 5) b1 $(ms --steps 100 --warmup 20 --batch 1) 720p $(ms --steps 40 --warmup 10 --height 720 --width 1280 --instances 30 --batch 1)"
  if [ -n "$F16" ]; then L="$L f16_1024 $(ms --steps 20 --warmup 5 --dtype f16 --height 1024 --width 1024 --batch 8) f16_b16 $(ms --steps 20 --warmup 5 --dtype f16) x3 $(ms --steps 20 --warmup 5 --dtype f32-bf16x3)"; fi
  echo "$L" | tee -a gpurun_out/${TAG}_ab.txt
done; done
