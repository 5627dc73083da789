# batch-1 step eager against hipGraph replay, 640x480 and 1280x720
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
Q="--steps 200 --warmup 30 --cpu-frames 0 --predict-calls 0 --no-split-mode --no-configs"
for S in "480 640 20" "720 1280 30"; do set -- $S
  for G in "" "--graph"; do
    python3 $R/bench.py --batch 1 --height $1 --width $2 --instances $3 $Q $G ${T:+--tuning $T} 2> $O/b1g.err | tail -1 > $O/b1g.json
    python3 -c "
import json; d=json.load(open('$O/b1g.json')); print('$1x$2 [$G] [$T]', round(d['ms_per_step'],3), 'ms', d['step_ms_min_median_max_rank0'])"
  done
done
