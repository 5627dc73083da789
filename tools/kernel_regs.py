#!/usr/bin/env python3
"""Register / LDS / occupancy table of the kernels of one HIP source for gfx950 (hipcc -Rpass-analysis=kernel-resource-usage).
usage: tools/kernel_regs.py quber_amd/csrc/conv_persist.hip [filter substring]"""
import os
import re
import subprocess
import sys

src = os.path.abspath(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-mllvm", "-simplifycfg-sink-common=false", *(["-fno-slp-vectorize"] if "wino_fused" in src else []),
                      "--cuda-device-only", "-c", os.path.basename(src), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                     cwd=os.path.dirname(src), capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s+\[-Rpass", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        name = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"quber::(\(anonymous namespace\)::)?|\(quber::ConvP\)|void ", "", name)}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
print("| kernel | VGPRs | AGPRs | scratch B/lane | VGPR spill | waves/SIMD | LDS B/block |\n|---|---|---|---|---|---|---|")
for r in rows:
    if flt in r["name"]:
        print(f"| {r['name']} | {r.get('VGPRs')} | {r.get('AGPRs')} | {r.get('ScratchSize [bytes/lane]')} | {r.get('VGPRs Spill')} | "
              f"{r.get('Occupancy [waves/SIMD]')} | {r.get('LDS Size [bytes/block]')} |")
