#!/bin/bash
# Same-box A/B of library variants, per convolution stage of the 16-frame step (bench.py's roofline.conv_stages, ms):
#   tools/stage_ab.sh <out tag> <reps> name1 name2 ...   ("new" = the product library, else quber_amd/libquber_hip_<name>.so)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; REPS=$2; shift 2
cd $R
for rep in $(seq $REPS); do
for v in "$@"; do
  if [ $v = new ]; then unset QUBER_LIB; else export QUBER_LIB=$R/quber_amd/libquber_hip_$v.so; fi
  python3 bench.py --steps 20 --warmup 5 --cpu-frames 0 --predict-calls 0 --no-split-mode --no-configs 2>/dev/null | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); cs=j['roofline']['conv_stages']; hs=j.get('roofline',{})
print('$rep $v', round(j['ms_per_step'],3), ' '.join('%s %.3f' % (k, v['ms']) for k, v in cs.items()))" | tee -a gpurun_out/${TAG}_stage_ab.txt
done; done
