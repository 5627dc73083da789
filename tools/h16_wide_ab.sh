#!/bin/bash
# fp16 data path at 1024x1024 batch 8: per-layer convolution times (rocprofv3 kernel trace) under tuning sets. usage (GPU box): tools/h16_wide_ab.sh "30=0" "30=1" ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for T in "$@"; do
  N=$(echo $T | tr '=,' '__')
  rocprofv3 --kernel-trace --output-format csv -d $O/w_$N -o lay -- python3 $R/tools/layer_profile.py run --plan $O/w_plan_$N.json --compute-dtype 2 --height 1024 --width 1024 --batch 8 --tuning $T > $O/w_$N.log 2>&1
  python3 $R/tools/layer_profile.py report --plan $O/w_plan_$N.json --trace $O/w_$N/lay_kernel_trace.csv > $O/w_layers_$N.md
  rm -rf $O/w_$N
  echo "$T: $(tail -1 $O/w_layers_$N.md)"
done
