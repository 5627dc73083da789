#!/usr/bin/env python3
"""F(4x4,3x3) Winograd layers of the refiner: the three-kernel pipeline (winograd.hip) against the single-kernel form
(wino_fused.hip), stand-alone ops on the layer shapes of the benchmarked plan.  Times are of the layer alone (filters
transformed once, before); `diff` = max |fused - pipeline| / max |pipeline|, `vs direct` the same against the direct kernel.
usage: tools/wino_fused_bench.py [frames=16] [filter substring]"""
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
for kv in os.environ.get("QUBER_TUNE", "").split(","):       # e.g. QUBER_TUNE=27=0
    if "=" in kv:
        lib.quber_set_tuning(*(int(x) for x in kv.split("=")))
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
flt = sys.argv[2] if len(sys.argv) > 2 else ""
flts = [f for f in flt.split(";")]           # several substrings: "head 128>128;stem"
LAYERS = [("fusion_res2 256>256 @120x160", 1, 120, 160, 256, 256, 1),
          ("fusion_res3 512>512 @60x80", 1, 60, 80, 512, 512, 1),
          ("res3.conv2 128>128 @60x80 (2 streams)", 2, 60, 80, 128, 128, 1),
          ("res4.conv2 256>256 @30x40 (2 streams)", 2, 30, 40, 256, 256, 1),
          ("res5.conv2 d2 512>512 @30x40 (2 streams)", 2, 30, 40, 512, 512, 2),
          ("res5.conv2 d4 512>512 @30x40 (2 streams)", 2, 30, 40, 512, 512, 4),
          ("aspp d6 2048>256 @30x40", 1, 30, 40, 2048, 256, 6),
          ("decoder.res3.fuse0 320>128 @60x80", 1, 60, 80, 320, 128, 1),
          ("decoder.res2.fuse0 160>128 @120x160", 1, 120, 160, 160, 128, 1),
          ("head 128>128 @120x160", 1, 120, 160, 128, 128, 1),
          ("heads x3 128>128 @120x160", 3, 120, 160, 128, 128, 1),
          ("res2.conv2 64>64 @120x160 (2 streams)", 2, 120, 160, 64, 64, 1),
          ("stem.conv2 32>32 @240x320 (2 streams)", 2, 240, 320, 32, 32, 1),
          ("stem.conv3 32>64 @240x320 (2 streams)", 2, 240, 320, 32, 64, 1),
          ("head 128>32 @120x160", 1, 120, 160, 128, 32, 1),
          ("head x3 128>32 @120x160", 3, 120, 160, 128, 32, 1)]
print(f"| layer ({F} frames) | GFLOP | pipeline ms | fused ms | speed-up | fused TFLOP/s (executed) | diff | fused vs direct | pipeline vs direct | direct kernel ms |")
print("|---|---|---|---|---|---|---|---|---|---|")


def timed(fn):  # noqa: E302
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


for name, ipf, H, W, Cin, Cout, d in LAYERS:
    if not any(f in name for f in flts):
        continue
    B = ipf * F
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(B, H, W, Cin, device="cuda", generator=g)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) / np.sqrt(Cin * 9)
    sc, sh = torch.rand(Cout, device="cuda", generator=g) + 0.5, torch.randn(Cout, device="cuda", generator=g)
    packed = torch.empty(Cout * 9 * Cin, device="cuda")
    yd = torch.empty(B, H, W, Cout, device="cuda")
    direct = lambda: _lib.check(lib.quber_op_conv2d(p(x), B, H, W, Cin, p(w), Cout, 3, 1, d, d, p(sc), p(sh), p(None), 1, p(packed), p(yd), st))
    direct()
    td = timed(direct)
    tiles = B * d * d * ((-(-H // d) + 3) // 4) * ((-(-W // d) + 3) // 4)
    u = torch.empty(36 * Cout * Cin, device="cuda")
    ws = torch.empty(max(36 * tiles * (Cin + Cout), 36 * Cout * Cin + 2 * B * Cin), device="cuda")
    ys, ts = [], []
    for fused in (0, 1):
        lib.quber_set_tuning(25, fused)
        y = torch.full((B, H, W, Cout), float("nan"), device="cuda")
        run = lambda: _lib.check(lib.quber_op_conv3x3_winograd(p(x), B, H, W, Cin, p(w), Cout, d, 4, p(sc), p(sh), 1, p(u), p(ws),
                                                               ws.numel(), p(y), st))
        run()
        lib.quber_set_tuning(26, 1)          # time the layer alone: the transformed filters of the call above are reused
        ts.append(timed(run))
        lib.quber_set_tuning(26, 0)
        ys.append(y)
    lib.quber_set_tuning(25, 1)
    scale = max(1.0, yd.abs().max().item())
    diff = (ys[1] - ys[0]).abs().max().item() / scale
    e1 = (ys[1] - yd).abs().max().item() / scale
    e0 = (ys[0] - yd).abs().max().item() / scale
    fl = 2.0 * B * H * W * Cin * 9 * Cout
    ex = 2.0 * 36 * tiles * Cin * Cout
    print("| %s | %.1f | %.3f | %.3f | %.2fx | %.1f | %.1e | %.1e | %.1e | %.3f |" % (name, fl / 1e9, ts[0], ts[1], ts[0] / ts[1], ex / ts[1] / 1e9,
                                                                                   diff, e1, e0, td), flush=True)
    del x, w, u, ws, ys, yd
