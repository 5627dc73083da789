#!/bin/bash
# Is the LDS path the limiter of the single-kernel Winograd layers?    Counter passes + phase stamps on two layers.
#   usage (GPU box): tools/wino_fused_lds_pmc.sh <tag>  -> gpurun_out/<tag>_wf_lds_pmc.txt, <tag>_wf_stamps.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-rXX}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
: > $O/${TAG}_wf_lds_pmc.txt
for L in "head 128>128 @120x160" "stem.conv2 32>32"; do
  i=0
  for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rm -rf $O/wf_lds_pmc$i
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/wf_lds_pmc$i -o p -- python3 $R/tools/wino_fused_bench.py 16 "$L" > $O/wf_lds_pmc$i.log 2>&1 || { echo "pmc pass $i failed"; tail -5 $O/wf_lds_pmc$i.log; exit 1; }
    echo "== $L, pass $i" >> $O/${TAG}_wf_lds_pmc.txt
    python3 $R/tools/kernel_pmc.py $O/wf_lds_pmc$i/p_counter_collection.csv wino_fused >> $O/${TAG}_wf_lds_pmc.txt
  done
done
cd $R
DL=$(tools/diag_build.sh wfstamps WFX=-DWF_STAMPS) || exit 1
{ QUBER_LIB=$DL python3 tools/wf_stamps.py 120 160 128 128; QUBER_LIB=$DL python3 tools/wf_stamps.py 240 320 32 32 32; } > $O/${TAG}_wf_stamps.txt 2>&1
cat $O/${TAG}_wf_lds_pmc.txt; grep -v amdgpu.ids $O/${TAG}_wf_stamps.txt
