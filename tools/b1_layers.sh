# per-layer tables of a batch-1 forward (the reference's own call pattern); usage: b1_layers.sh <tag> [HxW ...]
# per size: <tag>_b1_layers_<HxW>.md (side lanes off: launches in plan order, one row per convolution op) and
# <tag>_b1_kernels_<HxW>.txt (side lanes on, as predict() runs: span, busy time, idle gaps, every launch of the last forward)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-b1}; shift
cd /tmp && export TMPDIR=/tmp
for S in "${@:-480x640}"; do
  H=${S%x*}; W=${S#*x}
  rm -rf $O/b1_prof_$S $O/b1_prof_nolanes_$S
  rocprofv3 --kernel-trace --output-format csv -d $O/b1_prof_nolanes_$S -o lay -- python3 $R/tools/layer_profile.py run --plan $O/b1_plan_$S.json --batch 1 --iters 5 --height $H --width $W --tuning 24=0 > $O/b1_prof_$S.log 2>&1 || exit 1
  python3 $R/tools/layer_profile.py report --plan $O/b1_plan_$S.json --trace $O/b1_prof_nolanes_$S/lay_kernel_trace.csv > $O/${TAG}_b1_layers_$S.md || exit 1
  rocprofv3 --kernel-trace --output-format csv -d $O/b1_prof_$S -o lay -- python3 $R/tools/layer_profile.py run --plan $O/b1_plan_$S.json --batch 1 --iters 5 --height $H --width $W >> $O/b1_prof_$S.log 2>&1 || exit 1
  python3 $R/tools/b1_gaps.py $O/b1_prof_$S/lay_kernel_trace.csv > $O/${TAG}_b1_kernels_$S.txt || exit 1
done
