#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc pass: tools/kernel_pmc.py <counter_collection.csv> <kernel name substring> [...more csv]
Prints, for the dispatches whose kernel name contains the substring (the first dispatch dropped as warm-up), the mean
duration and the mean of every counter collected."""
import collections
import csv
import sys

sub = sys.argv[2]
for path in [sys.argv[1]] + sys.argv[3:]:
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if sub not in r["Kernel_Name"]:
            continue
        x = d.setdefault(r["Dispatch_Id"], {"ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        x[r["Counter_Name"]] = x.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    v = list(d.values())[1:] or list(d.values())
    if not v:
        print(path, ": no dispatch of", sub)
        continue
    print(f"{path}: {len(v)} dispatches of *{sub}*, mean {sum(x['ns'] for x in v) / len(v) / 1e3:.1f} us")
    for k in sorted(v[0]):
        if k != "ns":
            print(f"  {k:44s} {sum(x.get(k, 0.0) for x in v) / len(v):16.0f}")
