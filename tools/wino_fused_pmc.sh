#!/bin/bash
# PMC passes over the single-kernel Winograd layer on one layer shape.  usage (GPU box): tools/wino_fused_pmc.sh <tag> "<layer filter>"
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; FLT=$2; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"; do     # (a pass with TA_* counters hung the profiler: not collected)
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/${TAG}_pmc$i -o p -- python3 $R/tools/wino_fused_bench.py 16 "$FLT" > $O/${TAG}_pmc$i.log 2>&1
  python3 $R/tools/kernel_pmc.py $O/${TAG}_pmc$i/p_counter_collection.csv wino_fused
done
