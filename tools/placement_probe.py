#!/usr/bin/env python3
"""Does the run time of a memory-bound conv launch depend on where its three streams (input, residual, output) lie
relative to each other in HBM?  Places them in one pool at controlled skews and times the stand-alone conv op."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quber_amd import _lib  # noqa: E402

lib = _lib.load()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
CASES = [("res2.conv3 1x1 64>256 +res", 32, 120, 160, 64, 256, True),
         ("res2.shortcut 1x1 64>256", 32, 120, 160, 64, 256, False),
         ("res3.conv3 1x1 128>512 +res", 32, 60, 80, 128, 512, True),
         ("res4.conv3 1x1 256>1024 +res", 32, 30, 40, 256, 1024, True)]
pool = torch.empty(3 << 30, dtype=torch.uint8, device="cuda")
torch.manual_seed(0)
base = (pool.data_ptr() + (1 << 21) - 1) >> 21 << 21
for (name, B, H, W, Cin, Cout, res) in CASES:
    nx, ny = B * H * W * Cin * 4, B * H * W * Cout * 4
    w = torch.randn(Cout, Cin, 1, 1, device="cuda") / np.sqrt(Cin)
    sc, sh = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda")
    packed = torch.empty(Cout * ((Cin + 31) // 32 * 32), device="cuda")
    up = lambda v: (v + (1 << 21) - 1) >> 21 << 21
    print(name, "bytes moved %.0f MB" % ((nx + ny * (2 if res else 1)) / 1e6), flush=True)
    for skew in (0, 256, 1024, 4096, 4096 + 256, 16384, 65536 + 4096, (1 << 20) + 4096 * 3 + 256):
        px = base
        pr = up(px + nx) + skew
        py = up(pr + ny) + 2 * skew
        assert py + ny < pool.data_ptr() + pool.numel()
        ts = []
        for rd in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                _lib.check(lib.quber_op_conv2d(C.c_void_p(px), B, H, W, Cin, C.c_void_p(w.data_ptr()), Cout, 1, 1, 0, 1,
                                               C.c_void_p(sc.data_ptr()), C.c_void_p(sh.data_ptr()),
                                               C.c_void_p(pr if res else 0), 1, C.c_void_p(packed.data_ptr()), C.c_void_p(py), st))
            e1.record()
            torch.cuda.synchronize()
            if rd:
                ts.append(e0.elapsed_time(e1) / 4)
        t = float(np.median(ts))
        print("   skew %8d B: %.3f ms  %.2f TB/s" % (skew, t, (nx + ny * (2 if res else 1)) / t / 1e9), flush=True)
