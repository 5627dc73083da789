#!/bin/bash
# PMC passes over one layer of tools/h8_bench.py: counters per dispatch of the kernels whose name contains FILTER.
# usage (GPU box): tools/h8_pmc.sh <tag> <layer substring> [kernel name filter = conv_h8]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; LAYER=$2; FLT=${3:-conv_h8}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
           "SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/${TAG}_pmc$i -o p -- python3 $R/tools/h8_bench.py --only "$LAYER" --iters 3 > $O/${TAG}_pmc$i.log 2>&1
  python3 $R/tools/kernel_pmc.py $O/${TAG}_pmc$i/p_counter_collection.csv "$FLT" >> $O/${TAG}_pmc.txt 2>&1
  rm -rf $O/${TAG}_pmc$i
done
cat $O/${TAG}_pmc.txt > /dev/null
