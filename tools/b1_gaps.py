#!/usr/bin/env python3
"""Kernel list of the last forward in a rocprofv3 kernel trace: span, busy time, idle gaps, time per kernel name."""
import collections
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "stem_conv1" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
seg = rows[a:b]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("kernels", len(seg), "span us", (t1 - t0) / 1e3, "busy us", sum(map(dur, seg)) / 1e3)
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("quber::", "").split("(")[0][:90]
    agg[n][0] += 1
    agg[n][1] += dur(r)
for n, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:40]:
    print(f"{t / 1e3:9.1f} us {c:4d}  {n}")
end = int(seg[0]["End_Timestamp"])
gap = 0
for r in seg[1:]:
    s = int(r["Start_Timestamp"])
    if s > end:
        gap += s - end
    end = max(end, int(r["End_Timestamp"]))
print("idle (no kernel running) us", gap / 1e3)
for r in seg:
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {dur(r) / 1e3:8.1f}  grid {r.get('Grid_Size_X', '?') + 'x' + r.get('Grid_Size_Y', '') + 'x' + r.get('Grid_Size_Z', ''):>8} wg {r.get('Workgroup_Size_X', '?'):>4}  {r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').replace('quber::', '').split('(')[0][:100]}")
