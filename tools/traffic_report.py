#!/usr/bin/env python3
"""Joins the launch plan with two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md prescribes) into profiles/conv_hbm_traffic.json.  FETCH_SIZE is doubled: on gfx950 it reports
half the bytes of 16-byte-per-lane reads (128-B requests tallied at 64 B)."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import source_digest  # noqa: E402  (bench.py reports the file only while the kernel sources still match)


FAMILY = ("conv_igemm", "conv_h8", "conv_x8", "wino_input", "wino_output", "wino_fused", "stem_conv1", "splitk_reduce", "pk_fixup")


def family_bytes(path, counter):
    """-> {kernel family: counter sum} over the trace (the profiled run does ONE forward: layer_profile.py run --iters 1)"""
    seen, out = set(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or r["Dispatch_Id"] in seen:
            continue
        for fam in FAMILY:
            if fam in r["Kernel_Name"]:
                seen.add(r["Dispatch_Id"])
                out[fam] += float(r["Counter_Value"])
    return out


def main(plan_json, fetch_csv, write_csv, out_json):
    meta = json.load(open(plan_json))
    assert meta["iters"] == 1, "profile a single forward (layer_profile.py run --iters 1)"
    convs = [p for p in meta["plan"] if p[1] == "conv"]
    f = family_bytes(fetch_csv, "FETCH_SIZE")
    w = family_bytes(write_csv, "WRITE_SIZE")
    fam = {k: {"fetch_bytes": f[k] * 1024 * 2, "write_bytes": w[k] * 1024} for k in FAMILY}
    tot = sum(v["fetch_bytes"] + v["write_bytes"] for v in fam.values())
    out = {"batch": meta["batch"], "launches": len(convs), "bytes_per_launch": tot / len(convs),
           "source_digest": source_digest(), "dtype": {0: "f32", 1: "bf16", 2: "f16", 3: "f32-bf16x3"}[meta.get("compute_dtype", 0)],
           "fetch_bytes_total": sum(v["fetch_bytes"] for v in fam.values()),
           "write_bytes_total": sum(v["write_bytes"] for v in fam.values()),
           "note": "HBM-side bytes of one forward's convolution ops (GEMM launches, Winograd input / output transforms, "
                   "split-K reduce passes) divided by the number of convolution ops; FETCH_SIZE x2 (gfx950 correction), "
                   "WRITE_SIZE as read; separate --pmc passes", "by_kernel_family": fam}
    json.dump(out, open(out_json, "w"), indent=1)
    print("bytes_per_launch %.1f MB, fetch %.1f GB, write %.1f GB" % (out["bytes_per_launch"] / 1e6, out["fetch_bytes_total"] / 1e9, out["write_bytes_total"] / 1e9))
    for k, v in fam.items():
        print("   %-14s fetch %.2f GB  write %.2f GB" % (k, v["fetch_bytes"] / 1e9, v["write_bytes"] / 1e9))


if __name__ == "__main__":
    main(*sys.argv[1:5])
