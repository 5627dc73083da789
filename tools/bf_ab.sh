# bf16x3 bench line under tuning sets; usage: bf_ab.sh "" "37=4" ...
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for T in "$@"; do
  timeout -k 10 300 python3 $R/bench.py --dtype f32-bf16x3 --steps 20 --warmup 5 --cpu-frames 0 --predict-calls 0 --no-split-mode ${T:+--tuning $T} > $O/bf_ab.json 2> $O/bf_ab.err
  python3 -c "
import json; d=json.load(open('$O/bf_ab.json')); r=d['roofline']['conv_stages']; print('[$T]', round(d['value'],1), round(d['ms_per_step'],3), {k:(round(v['ms'],2), v['launches']) for k,v in r.items()})"
done
