#!/usr/bin/env python3
"""Joins the launch plan with two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md prescribes) into profiles/conv_hbm_traffic.json.  FETCH_SIZE is doubled: on gfx950 it reports
half the bytes of 16-byte-per-lane reads (128-B requests tallied at 64 B)."""
import collections
import csv
import json
import sys


def conv_rows(path, counter, n):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "conv_igemm" in r["Kernel_Name"]:
            d[r["Dispatch_Id"]] = float(r["Counter_Value"])
    v = list(d.values())
    assert len(v) >= n and len(v) % n == 0, (len(v), n)
    return v[-n:]


def main(plan_json, fetch_csv, write_csv, out_json):
    meta = json.load(open(plan_json))
    convs = [p for p in meta["plan"] if p[1] == "conv"]
    B = meta["batch"]
    f = conv_rows(fetch_csv, "FETCH_SIZE", len(convs))
    w = conv_rows(write_csv, "WRITE_SIZE", len(convs))
    layers = []
    for c, a, b in zip(convs, f, w):
        layers.append({"layer": c[0], "flops": c[2] * B, "fetch_bytes": a * 1024 * 2, "write_bytes": b * 1024})
    tot = sum(l["fetch_bytes"] + l["write_bytes"] for l in layers)
    out = {"batch": B, "launches": len(layers), "bytes_per_launch": tot / len(layers),
           "fetch_bytes_total": sum(l["fetch_bytes"] for l in layers), "write_bytes_total": sum(l["write_bytes"] for l in layers),
           "note": "HBM-side bytes of the conv_igemm_f32 launches of one forward; FETCH_SIZE x2 (gfx950 correction), "
                   "WRITE_SIZE as read; separate --pmc passes", "layers": layers}
    json.dump(out, open(out_json, "w"), indent=1)
    print("bytes_per_launch %.1f MB, fetch %.1f GB, write %.1f GB" % (out["bytes_per_launch"] / 1e6, out["fetch_bytes_total"] / 1e9, out["write_bytes_total"] / 1e9))


if __name__ == "__main__":
    main(*sys.argv[1:5])
