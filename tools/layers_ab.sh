#!/bin/bash
# per-layer convolution times (rocprofv3 kernel trace) under tuning sets.  usage (GPU box): tools/layers_ab.sh <dtype> <h> <w> <batch> "30=0" "30=1" ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
DT=$1; H=$2; W=$3; B=$4; shift 4
for T in "$@"; do
  N=d${DT}_$(echo $T | tr '=,' '__')
  rocprofv3 --kernel-trace --output-format csv -d $O/w_$N -o lay -- python3 $R/tools/layer_profile.py run --plan $O/w_plan_$N.json --compute-dtype $DT --height $H --width $W --batch $B --tuning $T > $O/w_$N.log 2>&1
  python3 $R/tools/layer_profile.py report --plan $O/w_plan_$N.json --trace $O/w_$N/lay_kernel_trace.csv > $O/w_layers_$N.md
  rm -rf $O/w_$N
  echo "$T: $(tail -1 $O/w_layers_$N.md)"
done
