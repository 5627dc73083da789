# the step at several batches / frame sizes under tuning sets; usage: batch_ab.sh "" "42=0" ...
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
Q="--cpu-frames 0 --predict-calls 0 --no-split-mode --no-configs"
for T in "$@"; do
  L="[$T]"
  for C in "1 480 640 20 100 20" "2 480 640 20 60 10" "4 480 640 20 40 10" "8 480 640 20 20 5" "16 480 640 20 20 5" "1 720 1280 30 60 10" "2 720 1280 30 40 10"; do set -- $C
    v=$(python3 $R/bench.py $Q --batch $1 --height $2 --width $3 --instances $4 --steps $5 --warmup $6 ${T:+--tuning $T} 2>/dev/null | tail -1 | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'],3))")
    L="$L b$1@$3x$2 $v"
  done
  echo $L
done
