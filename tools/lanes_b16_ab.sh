#!/bin/bash
# A/B on the GPU box: side lanes (key 24) allowed up to batch 16 instead of the product library's LANE_BATCH - a scratch build with LANE_BATCH raised, loaded through QUBER_LIB.
#   usage: tools/lanes_b16_ab.sh <tag>   -> gpurun_out/<tag>_lanes_b16_ab.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; TAG=${1:-rXX}; O=$R/gpurun_out
D=/tmp/quber_diag_lanes16
rm -rf $D && mkdir -p $D/quber_amd/csrc $D/include && cp $R/include/*.h $D/include/ && cp $R/quber_amd/csrc/*.hip $R/quber_amd/csrc/*.h $R/quber_amd/csrc/Makefile $D/quber_amd/csrc/ || exit 1
sed -i 's/constexpr int LANE_BATCH = [0-9]*;/constexpr int LANE_BATCH = 16;/' $D/quber_amd/csrc/plan.hip
grep -q "LANE_BATCH = 16" $D/quber_amd/csrc/plan.hip || exit 1
make -C $D/quber_amd/csrc -j16 > $D/build.log 2>&1 || { tail -20 $D/build.log; exit 1; }
cd $R
Q="--cpu-frames 0 --predict-calls 0 --no-split-mode --no-configs --steps 20 --warmup 5"
{
run() { python3 bench.py $Q "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['value'])"; }
for cfgl in "--batch 16" "--batch 16 --dtype f32-bf16x3" "--batch 16 --dtype f16" "--batch 12"; do
  for rep in 1 2; do
    echo "== default library, $cfgl"; run $cfgl
    echo "== lanes up to batch 16, $cfgl"; QUBER_LIB=$D/quber_amd/libquber_hip.so run $cfgl
  done
done
} > $O/${TAG}_lanes_b16_ab.txt 2>&1
cat $O/${TAG}_lanes_b16_ab.txt
