#!/bin/bash
# Where the single-kernel Winograd layer's time goes: rebuild wino_fused.hip without one ingredient at a time
# (results are then wrong, only the time is of interest) and time one layer.   usage (GPU box): tools/wino_fused_ablate.sh "<layer filter>"
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for skip in ${SKIPS:-0 1 2 3 4 7 16 17 18 19 20 23}; do
  rm -f quber_amd/csrc/wino_fused.o
  make -C quber_amd/csrc WFX=-DWF_SKIP=$skip > /dev/null 2>&1
  echo "WF_SKIP=$skip"
  python tools/wino_fused_bench.py 16 "$1" 2>/dev/null | tail -n +3 | cut -d'|' -f2,4,5,7
done
rm -f quber_amd/csrc/wino_fused.o
make -C quber_amd/csrc > /dev/null 2>&1
