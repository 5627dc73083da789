#!/bin/bash
# What a short-K residual 1x1 layer spends its time on: timing with / without residual and affine, then two counter passes.
#   usage (GPU box): tools/short_k_pmc.sh <tag> [layer]  -> gpurun_out/<tag>_short_k.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-rXX}; L=${2:-res2}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
{
for ra in "1 1" "0 1" "1 0" "0 0"; do python3 $R/tools/short_k_probe.py $L 50 $ra; done
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_ANY"; do
  i=$((i+1)); rm -rf $O/sk_pmc$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/sk_pmc$i -o p -- python3 $R/tools/short_k_probe.py $L 5 > $O/sk_pmc$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/sk_pmc$i.log; }
  python3 $R/tools/kernel_pmc.py $O/sk_pmc$i/p_counter_collection.csv conv_igemm
  rm -rf $O/sk_pmc$i
done
} > $O/${TAG}_short_k_$L.txt 2>&1
cat $O/${TAG}_short_k_$L.txt
