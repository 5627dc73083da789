#!/usr/bin/env python3
"""Micro-benchmark of the stand-alone convolution op across arithmetic modes (quber_set_tuning key 12) and forced tile
shapes (key 4) on representative GEMM-bound refiner layers.  GPU box only."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quber_amd import _lib  # noqa: E402

LAYERS = [
    ("fusion_res5.conv 1x1 4096>2048 @30x40", 16, 30, 40, 4096, 2048, 1, 1, 1),
    ("res5.shortcut 1x1 1024>2048 @30x40 x2", 32, 30, 40, 1024, 2048, 1, 1, 1),
    ("res4.conv1 1x1 1024>256 @30x40 x2", 32, 30, 40, 1024, 256, 1, 1, 1),
    ("wino-like GEMM K=256 N=256 M=19200x36", 36, 120, 160, 256, 256, 1, 1, 1),
    ("aspp 3x3 d18 2048>256 @30x40", 16, 30, 40, 2048, 256, 3, 1, 18),
    ("stem.conv3 3x3 32>64 @240x320 x2", 32, 240, 320, 32, 64, 3, 1, 1),
]
# name, key 12 (arithmetic), key 4 (forced tile)
CONFIGS = [("f32 auto", 0, 0), ("x3 auto", 3, 0), ("x3 64x64", 3, 1), ("x3 128x128", 3, 2), ("f16 auto", 2, 0)]


def main():
    lib = _lib.load()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    lib.quber_set_tuning(2, 0)          # no split-K workspace: every launch computes whole tiles
    print("| layer | " + " | ".join(f"{c[0]} TF/s" for c in CONFIGS) + " |")
    print("|---|" + "---|" * len(CONFIGS))
    for (name, B, H, W, Cin, Cout, k, s, d) in LAYERS:
        x = torch.randn(B, H, W, Cin, device="cuda")
        w = torch.randn(Cout, Cin, k, k, device="cuda") / np.sqrt(Cin * k * k)
        sc, sh = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda")
        pad = d * (k // 2)
        y = torch.empty(B, H, W, Cout, device="cuda")
        packed = torch.empty(Cout * ((k * k * Cin + 31) // 32 * 32), device="cuda")
        flops = 2.0 * B * H * W * Cin * k * k * Cout
        res = []
        for (_, dt, tile) in CONFIGS:
            lib.quber_set_tuning(12, dt)
            lib.quber_set_tuning(4, tile)
            ts = []
            for rd in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    _lib.check(lib.quber_op_conv2d(p(x), B, H, W, Cin, p(w), Cout, k, s, pad, d, p(sc), p(sh), p(None), 1,
                                                   p(packed), p(y), st))
                e1.record()
                torch.cuda.synchronize()
                if rd:
                    ts.append(e0.elapsed_time(e1) / 3)
            res.append(flops / (np.median(ts) * 1e-3) / 1e12)
        print(f"| {name} | " + " | ".join("%.1f" % r for r in res) + " |", flush=True)
    lib.quber_set_tuning(12, 0)
    lib.quber_set_tuning(4, 0)
    lib.quber_set_tuning(2, 0)


if __name__ == "__main__":
    main()
