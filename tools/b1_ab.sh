# batch-1 bench line under tuning sets; usage: b1_ab.sh "" "14=8" "14=8,15=128" ...
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for T in "$@"; do
  timeout -k 10 300 python3 $R/bench.py --batch 1 --steps 100 --warmup 20 --cpu-frames 0 --predict-calls 0 --no-split-mode ${T:+--tuning $T} > $O/b1_ab.json 2> $O/b1_ab.err
  python3 -c "
import json; d=json.load(open('$O/b1_ab.json')); r=d['roofline']['conv_stages']; print('[$T]', round(d['ms_per_step'],3), {k:round(v['ms'],3) for k,v in r.items()})"
done
