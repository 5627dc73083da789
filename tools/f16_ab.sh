# fp16 1024x1024 batch 8 bench line, optionally with --tuning; usage: f16_ab.sh <tag> [tuning]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
T=${2:-}
timeout -k 10 300 python3 $R/bench.py --dtype f16 --height 1024 --width 1024 --batch 8 --steps 20 --warmup 5 --cpu-frames 0 --predict-calls 0 ${T:+--tuning $T} > $O/$1.json 2> $O/$1.err
python3 -c "
import json,sys; d=json.load(open('$O/$1.json')); r=d['roofline']['conv_stages']; print('$1', round(d['value'],1), round(d['ms_per_step'],3), {k:round(v['ms'],3) for k,v in r.items()})"
