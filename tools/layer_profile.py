#!/usr/bin/env python3
"""Per-layer view of the convolution kernels.

  run    : build the engine, run a few forwards at the benchmark shape and write the launch plan to JSON
           (meant to be executed under `rocprofv3 --kernel-trace --output-format csv`)
  report : join the plan with rocprofv3's kernel_trace.csv -> per-layer time / TFLOP/s table (markdown)
"""
import argparse
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(a):
    import torch
    from quber_amd import arch, engine, synth
    eng = engine.Engine(engine.make_config(a.height, a.width, max_batch=a.batch), "cuda:0")
    eng.load_state_dict(arch.init_state_dict(seed=0))
    b = synth.make_batch(7, a.batch, a.height, a.width, 20)
    masks, bgr, depth = (torch.from_numpy(b[k]).cuda() for k in ("masks", "rgb", "depth"))
    offs = eng.encode(masks)
    for _ in range(a.iters):
        eng.forward(bgr, depth, offs)
    torch.cuda.synchronize()
    json.dump({"batch": a.batch, "iters": a.iters, "plan": eng.plan()}, open(a.plan, "w"))


def report(a):
    meta = json.load(open(a.plan))
    plan, B = meta["plan"], meta["batch"]
    convs = [p for p in plan if p[1] == "conv"]
    rows = [r for r in csv.DictReader(open(a.trace)) if "conv_igemm" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    n = len(convs)
    assert len(rows) % n == 0 and len(rows) >= n, (len(rows), n)
    last = rows[-n:]
    tot_t = tot_f = 0.0
    out = ["| # | layer (first weight key) | tile | GFLOP (batch %d) | ms | TFLOP/s |" % B, "|---|---|---|---|---|---|"]
    for i, (c, r) in enumerate(zip(convs, last)):
        ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        fl = c[2] * B
        tile = r["Kernel_Name"].split("<")[1].split(">")[0].replace(" ", "")
        out.append("| %d | %s | %s | %.1f | %.3f | %.1f |" % (i, c[0].replace("backbone.", "b.").replace("ins_embed_head.", "h."), tile, fl / 1e9, ms, fl / ms / 1e9))
        tot_t += ms
        tot_f += fl
    out.append("| | **all convolutions** | | %.1f | %.3f | %.1f |" % (tot_f / 1e9, tot_t, tot_f / tot_t / 1e9))
    print("\n".join(out))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["run", "report"])
    ap.add_argument("--plan", default="gpurun_out/plan.json")
    ap.add_argument("--trace")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--iters", type=int, default=3)
    a = ap.parse_args()
    run(a) if a.mode == "run" else report(a)
