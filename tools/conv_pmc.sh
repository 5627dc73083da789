#!/bin/bash
# PMC passes over the network's convolution kernels: counters summed over the launches > MINUS us whose kernel name contains FILTER.
# usage (GPU box): tools/conv_pmc.sh <tag> <tuning> <compute dtype> <height> <width> <batch> <kernel name filter> [min us]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; TUN=$2; DT=$3; H=$4; W=$5; B=$6; FLT=$7; MINUS=${8:-300}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/${TAG}_pmc$i -o p -- python3 $R/tools/layer_profile.py run --plan $O/${TAG}_plan.json --iters 1 --compute-dtype $DT --height $H --width $W --batch $B --tuning $TUN > $O/${TAG}_pmc$i.log 2>&1
  python3 - $O/${TAG}_pmc$i/p_counter_collection.csv "$FLT" $MINUS <<'PY'
import collections, csv, sys
d = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] not in r["Kernel_Name"]: continue
    x = d.setdefault(r["Dispatch_Id"], {"ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
    x[r["Counter_Name"]] = x.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
v = [x for x in d.values() if x["ns"] > 1000 * int(sys.argv[3])]
print(f"{len(v)} launches of *{sys.argv[2]}* > {sys.argv[3]} us, total {sum(x['ns'] for x in v) / 1e6:.3f} ms")
for k in sorted(v[0]):
    if k != "ns": print(f"  {k:36s} {sum(x.get(k, 0.0) for x in v):18.0f}")
PY
  rm -rf $O/${TAG}_pmc$i
done
