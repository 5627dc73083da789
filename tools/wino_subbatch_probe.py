#!/usr/bin/env python3
"""Does a Winograd layer run faster when its V / M intermediates stay inside the 256 MiB Infinity Cache?
Times the stand-alone Winograd op (ms) on F frames at once against the same F frames in passes whose V | M footprint stays
below a budget (quber_set_tuning key 20); the passes reuse one workspace, so the intermediates of a pass are rewritten in
place before they are evicted.
usage: tools/wino_subbatch_probe.py [frames=16]"""
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
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
m = 4
P = (m + 2) ** 2


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


BUDGETS = (0, 512, 256, 192, 128, 96, 64, 32)
print(f"| layer ({F} frames, F(4x4)) | V+M MiB per frame | " + " | ".join("whole batch" if b == 0 else f"<= {b} MiB" for b in BUDGETS) + " |")
print("|---|---|" + "---|" * len(BUDGETS))
for name, ipf, H, W, Cin, Cout, d in [("fusion_res2 3x3 256>256 @120x160", 1, 120, 160, 256, 256, 1),
                                      ("fusion_res3 3x3 512>512 @60x80", 1, 60, 80, 512, 512, 1),
                                      ("head 3x3 128>128 @120x160", 1, 120, 160, 128, 128, 1),
                                      ("decoder.res2.fuse0 3x3 160>128 @120x160", 1, 120, 160, 160, 128, 1),
                                      ("head 3x3 128>32 @120x160", 1, 120, 160, 128, 32, 1),
                                      ("res3.conv2 3x3 128>128 @60x80 (2 streams)", 2, 60, 80, 128, 128, 1),
                                      ("res4.conv2 3x3 256>256 @30x40 (2 streams)", 2, 30, 40, 256, 256, 1),
                                      ("res5.conv2 3x3 d2 512>512 @30x40 (2 streams)", 2, 30, 40, 512, 512, 2),
                                      ("aspp 3x3 d6 2048>256 @30x40", 1, 30, 40, 2048, 256, 6)]:
    B = ipf * F
    x = torch.randn(B, H, W, Cin, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") / np.sqrt(Cin * 9)
    sc, sh = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda")
    y = torch.empty(B, H, W, Cout, device="cuda")
    tpf = d * d * ((-(-H // d) + m - 1) // m) * ((-(-W // d) + m - 1) // m)      # tiles per frame
    u = torch.empty(P * Cout * Cin, device="cuda")
    ws = torch.empty(P * B * tpf * (Cin + Cout), device="cuda")
    cols = []
    ref = None
    for mb in BUDGETS:
        lib.quber_set_tuning(20, mb)
        t = timed(lambda: _lib.check(lib.quber_op_conv3x3_winograd(p(x), B, H, W, Cin, p(w), Cout, d, m, p(sc), p(sh), 1, p(u), p(ws),
                                                                   ws.numel(), p(y), st)))
        if ref is None:
            ref = y.clone()
        else:
            assert float((ref - y).abs().max()) <= 1e-5 * float(ref.abs().max()), "chunked result differs"   # (split-K may re-associate)
        cols.append("%.3f" % t)
    lib.quber_set_tuning(20, 0)
    print("| %s | %.0f | %s |" % (name, P * tpf * (Cin + Cout) * 4 / 1048576, " | ".join(cols)), flush=True)
    del x, y, ws, u
