#!/bin/bash
# End-to-end A/B of the single-kernel Winograd layers: the default bench step with the layers off (25=0), on (default: inputs
# up to 160 channels: the default of option key 27), and with wider layers admitted (27=...), for the WF_SPLIT builds given.  usage (GPU box): tools/wino_fused_ab.sh "9 7"
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
B="python bench.py --steps 12 --warmup 4 --cpu-frames 0 --predict-calls 0 --no-split-mode"
for sp in ${1:-9}; do
  DL=$(tools/diag_build.sh wfsplit$sp WFX=-DWF_SPLIT=$sp) || exit 1      # scratch copy: the product library is never touched
  export QUBER_LIB=$DL
  for t in "25=0" "25=1" "25=1,27=160" "25=1,27=320" "25=1,27=512"; do
    echo "WF_SPLIT=$sp tuning $t: $($B --tuning $t 2>/dev/null | python3 -c 'import json,sys; j=json.loads(sys.stdin.readlines()[-1]); print(round(j["value"],1), "masks/s", round(j["ms_per_step"],2), "ms", "frac", round(j["roofline"]["frac"],3))')"
  done
done
unset QUBER_LIB
