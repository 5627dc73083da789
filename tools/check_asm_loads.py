#!/usr/bin/env python3
"""Audit of a kernel's .s: no instruction may touch the destination registers of an inline-asm global load between the load
and the next counted wait (cdna_hip_programming.md 5.7 item 1).  usage: check_asm_loads.py kernel.s"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
loads = [(i, l) for i, l in enumerate(lines) if "global_load_dwordx4" in l and "off" in l]
bad = 0
for i, l in loads:
    m = re.search(r"v\[(\d+):(\d+)\]", l)
    if not m:
        continue
    regs = {f"v{r}" for r in range(int(m.group(1)), int(m.group(2)) + 1)}
    j = i + 1
    while j < len(lines) and "s_waitcnt vmcnt" not in lines[j] and "s_endpgm" not in lines[j]:
        t = lines[j]
        if "global_load" not in t and not t.strip().startswith((";", ".")):
            used = {f"v{x}" for x in re.findall(r"\bv(\d+)\b", t)}
            for a, b in re.findall(r"v\[(\d+):(\d+)\]", t):
                used |= {f"v{r}" for r in range(int(a), int(b) + 1)}
            if used & regs:
                bad += 1
                print("TOUCH", i + 1, l.strip()[:60], "->", j + 1, t.strip()[:80])
        j += 1
print("loads", len(loads), "violations", bad)
sys.exit(1 if bad else 0)
