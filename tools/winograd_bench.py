#!/usr/bin/env python3
"""Direct implicit-GEMM vs Winograd F(2x2,3x3) on the eligible 3x3 layers of the refiner (stand-alone ops)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quber_amd import _lib  # noqa: E402

lib = _lib.load()
lib.quber_set_tuning(2, 1)
lib.quber_set_tuning(7, 32)
lib.quber_set_tuning(10, 32)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
print(f"| layer ({F} frames) | GFLOP | direct ms | TF/s | F(2x2) ms | speed-up | F(4x4) ms | speed-up | F(6x6) ms | speed-up |")
print("|---|---|---|---|---|---|---|---|---|---|")
for name, ipf, H, W, Cin, Cout, *rest in [("fusion_res2 3x3 256>256 @120x160", 1, 120, 160, 256, 256),
                                   ("res5.conv2 3x3 d2 512>512 @30x40 (2 streams)", 2, 30, 40, 512, 512, 2),
                                   ("res5.conv2 3x3 d4 512>512 @30x40 (2 streams)", 2, 30, 40, 512, 512, 4),
                                   ("res5.conv2 3x3 d8 512>512 @30x40 (2 streams)", 2, 30, 40, 512, 512, 8),
                                   ("aspp 3x3 d6 2048>256 @30x40", 1, 30, 40, 2048, 256, 6),
                                   ("aspp 3x3 d12 2048>256 @30x40", 1, 30, 40, 2048, 256, 12),
                                   ("aspp 3x3 d18 2048>256 @30x40", 1, 30, 40, 2048, 256, 18),
                                   ("fusion_res3 3x3 512>512 @60x80", 1, 60, 80, 512, 512),
                                   ("res4.conv2 3x3 256>256 @30x40 (2 streams)", 2, 30, 40, 256, 256),
                                   ("decoder.res3.fuse0 3x3 320>128 @60x80", 1, 60, 80, 320, 128),
                                   ("decoder.res2.fuse0 3x3 160>128 @120x160", 1, 120, 160, 160, 128),
                                   ("head 3x3 128>128 @120x160", 1, 120, 160, 128, 128),
                                   ("heads x3 3x3 128>128 @120x160", 3, 120, 160, 128, 128),
                                   ("head 3x3 128>32 @120x160", 1, 120, 160, 128, 32),
                                   ("res3.conv2 3x3 128>128 @60x80 (2 streams)", 2, 60, 80, 128, 128),
                                   ("res2.conv2 3x3 64>64 @120x160 (2 streams)", 2, 120, 160, 64, 64),
                                   ("stem.conv3 3x3 32>64 @240x320 (2 streams)", 2, 240, 320, 32, 64)]:
    B = ipf * F
    d = rest[0] if rest else 1
    x = torch.randn(B, H, W, Cin, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") / np.sqrt(Cin * 9)
    sc, sh = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda")
    y = torch.empty(B, H, W, Cout, device="cuda")
    packed = torch.empty(Cout * 9 * Cin, device="cuda")
    fl = 2.0 * B * H * W * Cin * 9 * Cout

    def timed(fn):
        ts = []
        for rd in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                fn()
            e1.record()
            torch.cuda.synchronize()
            if rd:
                ts.append(e0.elapsed_time(e1) / 3)
        return float(np.median(ts))

    td = timed(lambda: _lib.check(lib.quber_op_conv2d(p(x), B, H, W, Cin, p(w), Cout, 3, 1, d, d, p(sc), p(sh), p(None), 1,
                                                      p(packed), p(y), st)))
    cols = []
    for m in (2, 4, 6):
        P = (m + 2) ** 2
        tiles = B * d * d * ((-(-H // d) + m - 1) // m) * ((-(-W // d) + m - 1) // m)
        u = torch.empty(P * Cout * Cin, device="cuda")
        ws = torch.empty(P * tiles * (Cin + Cout), device="cuda")
        try:
            tw = timed(lambda: _lib.check(lib.quber_op_conv3x3_winograd(p(x), B, H, W, Cin, p(w), Cout, d, m, p(sc), p(sh), 1, p(u),
                                                                        p(ws), ws.numel(), p(y), st)))
            cols.append("%.3f | %.2fx" % (tw, td / tw))
        except Exception as e:
            cols.append("n/a | ")
        del u, ws
    print("| %s | %.1f | %.3f | %.1f | %s |" % (name, fl / 1e9, td, fl / td / 1e9, " | ".join(cols)), flush=True)
