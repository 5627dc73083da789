#!/usr/bin/env python3
"""rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT)
joined with the launch plan -> per-layer MFMA utilisation table (markdown)."""
import collections
import csv
import json
import sys

plan_json, counters_csv = sys.argv[1:3]
meta = json.load(open(plan_json))
convs = [p for p in meta["plan"] if p[1] == "conv"]
B = meta["batch"]
d = collections.OrderedDict()
for r in csv.DictReader(open(counters_csv)):
    x = d.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"], "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                                        "t0": int(r["Start_Timestamp"])})
    x[r["Counter_Name"]] = float(r["Counter_Value"])
GEMM = ("conv_igemm", "conv_h8", "conv_x8")     # the implicit-GEMM kernels (csrc/conv_igemm.hip, conv_persist.hip, conv_h8 / x8)
# one group of GEMM launches per convolution op (a Winograd op whose groups share one input transform launches several)
disp = sorted(d.values(), key=lambda v: v["t0"])
groups, i = [], 0
while i < len(disp):
    n_ = disp[i]["name"]
    if "wino_input" in n_:
        j = i + 1
        while j < len(disp) and "wino_output" not in disp[j]["name"]:
            j += 1
        groups.append([v for v in disp[i:j + 1] if any(t in v["name"] for t in GEMM)])
        i = j + 1
    elif "wino_fused" in n_ or "stem_conv1" in n_:
        groups.append([disp[i]])
        i += 1
    elif any(t in n_ for t in GEMM):
        groups.append([disp[i]])
        i += 1
    else:
        i += 1
rows = []
for g in groups[-len(convs):]:
    v = {"ns": sum(x["ns"] for x in g)}
    for k in ("GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_BUSY_CYCLES"):
        v[k] = sum(x.get(k, 0.0) for x in g)
    rows.append(v)
print("| layer | ms | TFLOP/s | MFMA busy (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs)) | eff. clock GHz | mean waves per CU | LDS bank conflicts |")
print("|---|---|---|---|---|---|---|")
tb = ta = tf = tn = 0.0
for c, v in zip(convs, rows):
    cyc = v["GRBM_GUI_ACTIVE"] / 8
    busy = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024)
    waves = v["SQ_WAVE_CYCLES"] * 4 / cyc / 256
    fl = c[2] * B
    tb += v["SQ_VALU_MFMA_BUSY_CYCLES"]; ta += cyc * 1024; tf += fl; tn += v["ns"]
    if fl / 1e9 > 60 * B / 16 or "stem" in c[0]:
        print("| %s | %.3f | %.1f | %.2f | %.2f | %.1f | %d |" % (c[0].replace("backbone.", "b.").replace("ins_embed_head.", "h."),
              v["ns"] / 1e6, fl / v["ns"] / 1e3, busy, cyc / v["ns"], waves, v["SQ_LDS_BANK_CONFLICT"]))
print("| **all %d convolution launches** | %.3f | %.1f | %.2f | | | |" % (len(convs), tn / 1e6, tf / tn / 1e3, tb / ta))
