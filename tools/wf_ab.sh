#!/bin/bash
# A/B of the single-kernel Winograd layer's build-time switches (wino_fused.hip WF_OPT) on the stand-alone layer bench, then the
# in-kernel phase stamps of the default build.   usage (GPU box): tools/wf_ab.sh "<variants, e.g. 0 1 3 7>" "<layer filter>" [stamps]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
# (every variant is built in a scratch copy of the sources, tools/diag_build.sh; the product library is never touched)
for o in $1; do
  DL=$(tools/diag_build.sh wfopt$o WFX="-DWF_OPT=$o") || exit 1
  echo "== WF_OPT=$o"
  QUBER_LIB=$DL python tools/wino_fused_bench.py 16 "$2" 2>&1 | tail -n +3 | cut -d'|' -f2,4,5,6,7,8
done
if [ -n "$3" ]; then
  DL=$(tools/diag_build.sh wfstamps WFX=-DWF_STAMPS) || exit 1
  QUBER_LIB=$DL python tools/wf_stamps.py 120 160 128 128
  QUBER_LIB=$DL python tools/wf_stamps.py 120 160 64 64 32
  QUBER_LIB=$DL python tools/wf_stamps.py 240 320 32 32 32
  QUBER_LIB=$DL python tools/wf_stamps.py 120 160 128 32
fi
