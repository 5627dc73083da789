# batch-1 per-layer tables under forced (tile, split) settings of the implicit GEMM: which choice is best per layer?
# usage: b1_tile_sweep.sh <tag> <HxW> "<tuning> ..."   -> gpurun_out/<tag>_b1_forced_<HxW>_<tuning>.md
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=$1; S=$2; shift 2
H=${S%x*}; W=${S#*x}
cd /tmp && export TMPDIR=/tmp
for T in "$@"; do
  N=$(echo "$T" | tr '=,' '__')
  rm -rf $O/b1_forced
  rocprofv3 --kernel-trace --output-format csv -d $O/b1_forced -o lay -- python3 $R/tools/layer_profile.py run --plan $O/b1_forced_plan.json --batch ${BATCH:-1} --iters ${ITERS:-4} --height $H --width $W --tuning "24=0${T:+,$T}" ${DTYPE:+--compute-dtype $DTYPE} > $O/b1_forced.log 2>&1 || { echo "run failed $T"; continue; }
  python3 $R/tools/layer_profile.py report --plan $O/b1_forced_plan.json --trace $O/b1_forced/lay_kernel_trace.csv > $O/${TAG}_b1_forced_${S}_$N.md || echo "report failed $T"
done
