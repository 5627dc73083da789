#!/usr/bin/env python3
"""The implicit GEMM with the LEAN loader (key 30 = 1) against the per-thread tap arithmetic (0), fp16 data path and fp32-tensor modes:
same bits on every head output, at a ragged size and at the benchmark sizes; then ms per forward of both.  GPU box only."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from quber_amd import arch, engine, synth

def run(h, w, b, dt, timing, modes=(0, 1)):
    qc = engine.make_config(h, w, max_batch=b)
    qc.compute_dtype = dt
    e = engine.Engine(qc, "cuda:0")
    e.load_state_dict(arch.init_state_dict(seed=0, loud_heads=True))
    d = synth.make_batch(7, b, h, w, 20)
    masks, bgr, depth = (torch.from_numpy(d[k]).cuda() for k in ("masks", "rgb", "depth"))
    offs = e.encode(masks)
    outs = {}
    for mode in modes:
        e.set_option(30, mode)
        o = e.forward(bgr, depth, offs)
        torch.cuda.synchronize()
        outs[mode] = o.clone() if torch.is_tensor(o) else {k: v.clone() for k, v in o.items()}
    a, c = outs[modes[0]], outs[modes[1]]
    if torch.is_tensor(a):
        same = torch.equal(a, c); diff = float((a.float() - c.float()).abs().max())
    else:
        same = all(torch.equal(a[k], c[k]) for k in a); diff = max(float((a[k].float() - c[k].float()).abs().max()) for k in a)
    print(f"{h}x{w} batch {b} dtype {dt}: identical {same} (max |diff| {diff:.3g})", flush=True)
    if timing:
        for mode in modes + modes:
            e.set_option(30, mode)
            for _ in range(3): e.forward(bgr, depth, offs)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): e.forward(bgr, depth, offs)
            torch.cuda.synchronize()
            print(f"  key 30 = {mode}: {(time.perf_counter() - t0) * 100:.3f} ms per forward", flush=True)
    return same

ok = run(150, 203, 3, 2, False)
ok &= run(1024, 1024, 8, 2, True)
# fp32-tensor modes (exact fp32, bf16x3): the one-tile-per-block kernel's launches
ok &= run(150, 203, 3, 0, False)
ok &= run(150, 203, 3, 3, False)
ok &= run(480, 640, 16, 0, True)
sys.exit(0 if ok else 1)
