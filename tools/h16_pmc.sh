#!/bin/bash
# PMC passes over the fp16 data path at 1024x1024 batch 8: counters summed over the big 3x3 launches (> 0.5 ms) of the 128x128 / 128x256 kernel.
# usage (GPU box): tools/h16_pmc.sh <tag> [tuning]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; TUN=${2:-31=0}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/${TAG}_pmc$i -o p -- python3 $R/tools/layer_profile.py run --plan $O/${TAG}_plan.json --iters 1 --compute-dtype 2 --height 1024 --width 1024 --batch 8 --tuning $TUN > $O/${TAG}_pmc$i.log 2>&1
  python3 - $O/${TAG}_pmc$i/p_counter_collection.csv <<'PY'
import collections, csv, sys
d = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "conv_igemm_f32<128, 128, 2, 2, 0, 4>" not in n and "conv_igemm_f32<128, 256, 2, 2, 0, 4>" not in n: continue
    x = d.setdefault(r["Dispatch_Id"], {"ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
    x[r["Counter_Name"]] = x.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
v = [x for x in d.values() if x["ns"] > 500000]
print(f"{len(v)} launches > 0.5 ms, total {sum(x['ns'] for x in v) / 1e6:.3f} ms")
for k in sorted(v[0]):
    if k != "ns": print(f"  {k:36s} {sum(x.get(k, 0.0) for x in v):18.0f}")
PY
  rm -rf $O/${TAG}_pmc$i
done
