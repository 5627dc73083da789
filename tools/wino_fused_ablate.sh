#!/bin/bash
# Where the single-kernel Winograd layer's time goes: rebuild wino_fused.hip without one ingredient at a time
# (results are then wrong, only the time is of interest) and time one layer.   usage (GPU box): tools/wino_fused_ablate.sh "<layer filter>"
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for skip in ${SKIPS:-0 1 2 3 4 7 16 17 18 19 20 23}; do
  DL=$(tools/diag_build.sh wfskip$skip WFX=-DWF_SKIP=$skip) || exit 1     # scratch copy: the product library is never touched
  echo "WF_SKIP=$skip"
  QUBER_LIB=$DL python tools/wino_fused_bench.py 16 "$1" 2>/dev/null | tail -n +3 | cut -d'|' -f2,4,5,7
done
