set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for T in "31=0" "31=1" "31=1,32=256"; do
  N=$(echo $T | tr '=,' '__')
  rocprofv3 --kernel-trace --output-format csv -d $O/w_$N -o lay -- python3 $R/tools/layer_profile.py run --plan $O/w_plan_$N.json --compute-dtype 2 --height 1024 --width 1024 --batch 8 --tuning $T > $O/w_$N.log 2>&1
  python3 $R/tools/layer_profile.py report --plan $O/w_plan_$N.json --trace $O/w_$N/lay_kernel_trace.csv > $O/w_layers_$N.md
  rm -rf $O/w_$N
  tail -1 $O/w_layers_$N.md
done
cd $R
for T in "30=0" "30=1" "30=3" "31=1" ; do
  timeout -k 10 200 python bench.py --dtype f16 --height 1024 --width 1024 --batch 8 --cpu-frames 0 --tuning $T > $O/b16_$T.json 2> $O/b16_$T.err
  python - <<PY
import json
d=json.loads(open("$O/b16_$T.json").read().strip().splitlines()[-1]); print("$T", d["value"], d["ms_per_step"])
PY
done
