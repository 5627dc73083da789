#!/usr/bin/env python3
"""Which kernels wait for their loads one at a time?  Compiles every csrc/*.hip to gfx950 assembly (no GPU needed) and, per kernel,
counts the loads that are followed by `s_waitcnt vmcnt(0)` before the next load is issued - the shape `if (valid) { load; use }` in a loop
compiles to (profiles/r17_epilogue.md).  usage: tools/isa_serial_loads.py [file.hip ...]   (default: all of quber_amd/csrc)"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "quber_amd", "csrc")
EXTRA = {"conv_igemm": ["-mllvm", "-simplifycfg-sink-common=false"], "conv_persist": ["-mllvm", "-simplifycfg-sink-common=false"],
         "conv_x8": ["-mllvm", "-simplifycfg-sink-common=false"], "conv_h8": ["-mllvm", "-simplifycfg-sink-common=false"],
         "wino_fused": ["-fno-slp-vectorize"]}


def scan(path):
    kernels, name = {}, None
    for line in open(path):
        m = re.match(r"^(_Z\S+):", line)
        if m:
            name = m.group(1)
            kernels[name] = []
            continue
        if name is None:
            continue
        t = line.strip()
        if t.startswith(("global_load", "buffer_load", "flat_load")):
            kernels[name].append("L")
        elif t.startswith("s_waitcnt") and "vmcnt(0)" in t:
            kernels[name].append("W")
        elif t.startswith(("global_store", "buffer_store")):
            kernels[name].append("S")
    for k, seq in kernels.items():
        s = "".join(seq)
        loads, serial = s.count("L"), len(re.findall(r"(?<=W)LW", s)) + (1 if s.startswith("LW") else 0)
        if loads >= 6 and serial >= 4:
            dn = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
            print(f"{os.path.basename(path):18s} loads {loads:3d}  single load then vmcnt(0) {serial:3d}  {dn[:120]}")


def main():
    srcs = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    with tempfile.TemporaryDirectory() as d:
        for src in srcs:
            stem = os.path.splitext(os.path.basename(src))[0]
            out = os.path.join(d, stem + ".s")
            cmd = ["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off"] + EXTRA.get(stem, []) + \
                  ["-I", CSRC, "-I", os.path.join(ROOT, "include"), "-S", "--offload-device-only", src, "-o", out]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode:
                print(stem, ": compile failed\n", r.stderr[-400:])
                continue
            scan(out)


if __name__ == "__main__":
    main()
