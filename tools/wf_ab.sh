#!/bin/bash
# A/B of the single-kernel Winograd layer's build-time switches (wino_fused.hip WF_OPT) on the stand-alone layer bench, then the
# in-kernel phase stamps of the default build.   usage (GPU box): tools/wf_ab.sh "<variants, e.g. 0 1 3 7>" "<layer filter>" [stamps]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for o in $1; do
  rm -f quber_amd/csrc/wino_fused.o
  make -C quber_amd/csrc WFX="-DWF_OPT=$o" > /dev/null 2>&1 || { echo "build failed: WF_OPT=$o"; exit 1; }
  echo "== WF_OPT=$o"
  python tools/wino_fused_bench.py 16 "$2" 2>&1 | tail -n +3 | cut -d'|' -f2,4,5,6,7,8
done
if [ -n "$3" ]; then
  rm -f quber_amd/csrc/wino_fused.o quber_amd/csrc/plan.o
  make -C quber_amd/csrc WFX="-DWF_STAMPS" > /dev/null 2>&1 || { echo "build failed: stamps"; exit 1; }
  python tools/wf_stamps.py 120 160 128 128
  python tools/wf_stamps.py 120 160 64 64 32
  python tools/wf_stamps.py 240 320 32 32 32
  python tools/wf_stamps.py 120 160 128 32
fi
rm -f quber_amd/csrc/wino_fused.o quber_amd/csrc/plan.o
make -C quber_amd/csrc > /dev/null 2>&1
