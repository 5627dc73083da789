#!/usr/bin/env python3
"""A/B of the persistent convolution launch (tuning key 13) against the one-tile-per-block launch on the stand-alone conv
op, interleaved rounds in one process, with the split-K workspace allocated for both.  GPU box only.
usage: persist_bench.py [dtypes, e.g. 0,3] [tuning knobs, e.g. 15=0,14=16]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quber_amd import _lib  # noqa: E402

LAYERS = [
    # name, B, H, W, Cin, Cout, k, stride, dil, residual
    ("wino GEMM fusion_res2 K=256 N=256 M=36x19200", 36, 120, 160, 256, 256, 1, 1, 1, False),
    ("wino GEMM head K=128 N=128 M=36x19200", 36, 120, 160, 128, 128, 1, 1, 1, False),
    ("wino GEMM res5.conv2 K=512 N=512 M=2x16x1200", 32, 30, 40, 512, 512, 1, 1, 1, False),
    ("wino GEMM res4.conv2 K=256 N=256 M=2x36x1200", 72, 30, 40, 256, 256, 1, 1, 1, False),
    ("fusion_res5.conv 1x1 4096>2048 @30x40", 16, 30, 40, 4096, 2048, 1, 1, 1, False),
    ("res5.shortcut 1x1 1024>2048 @30x40 x2", 32, 30, 40, 1024, 2048, 1, 1, 1, False),
    ("res5.conv3 1x1 512>2048 +res @30x40 x2", 32, 30, 40, 512, 2048, 1, 1, 1, True),
    ("res5.conv1 1x1 2048>512 @30x40 x2", 32, 30, 40, 2048, 512, 1, 1, 1, False),
    ("res4.conv1 1x1 1024>256 @30x40 x2", 32, 30, 40, 1024, 256, 1, 1, 1, False),
    ("res4.conv3 1x1 256>1024 +res @30x40 x2", 32, 30, 40, 256, 1024, 1, 1, 1, True),
    ("res3.conv3 1x1 128>512 +res @60x80 x2", 32, 60, 80, 128, 512, 1, 1, 1, True),
    ("res2.conv3 1x1 64>256 +res @120x160 x2", 32, 120, 160, 64, 256, 1, 1, 1, True),
    ("res2.conv2 3x3 64>64 @120x160 x2", 32, 120, 160, 64, 64, 3, 1, 1, False),
    ("stem.conv3 3x3 32>64 @240x320 x2", 32, 240, 320, 32, 64, 3, 1, 1, False),
    ("fusion_res2.conv0 3x3 256>256 @120x160", 16, 120, 160, 256, 256, 3, 1, 1, False),
]


def main():
    dts = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "3"])]
    lib = _lib.load()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    lib.quber_set_tuning(2, 1)
    lib.quber_set_tuning(15, 0)
    for kv in (sys.argv[2].split(",") if len(sys.argv) > 2 else []):
        lib.quber_set_tuning(int(kv.split("=")[0]), int(kv.split("=")[1]))
    print("| layer | " + " | ".join(f"dt{d} tile-per-block TF/s | dt{d} persistent TF/s" for d in dts) + " |")
    print("|---|" + "---|---|" * len(dts))
    for (name, B, H, W, Cin, Cout, k, s, d, res) in LAYERS:
        x = torch.randn(B, H, W, Cin, device="cuda")
        w = torch.randn(Cout, Cin, k, k, device="cuda") / np.sqrt(Cin * k * k)
        sc, sh = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda")
        pad = d * (k // 2)
        y = torch.empty(B, H, W, Cout, device="cuda")
        r = torch.randn(B, H, W, Cout, device="cuda") if res else None
        packed = torch.empty(Cout * ((k * k * Cin + 31) // 32 * 32), device="cuda")
        flops = 2.0 * B * H * W * Cin * k * k * Cout
        cells = []
        for dt in dts:
            lib.quber_set_tuning(12, dt)
            ts = {0: [], 1: []}
            outs = {}
            for rd in range(6):
                for persist in (0, 1):
                    lib.quber_set_tuning(13, persist)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(3):
                        _lib.check(lib.quber_op_conv2d(p(x), B, H, W, Cin, p(w), Cout, k, s, pad, d, p(sc), p(sh), p(r), 1,
                                                       p(packed), p(y), st))
                    e1.record()
                    torch.cuda.synchronize()
                    if rd:
                        ts[persist].append(e0.elapsed_time(e1) / 3)
                    elif persist not in outs:
                        outs[persist] = y.clone()
            diff = (outs[0] - outs[1]).abs().max().item() / max(1.0, outs[0].abs().max().item())
            cells += ["%.1f" % (flops / (np.median(ts[0]) * 1e-3) / 1e12),
                      "%.1f (diff %.1e)" % (flops / (np.median(ts[1]) * 1e-3) / 1e12, diff)]
        print(f"| {name} | " + " | ".join(cells) + " |", flush=True)
    lib.quber_set_tuning(12, 0)
    lib.quber_set_tuning(13, 1)
    lib.quber_set_tuning(2, 0)


if __name__ == "__main__":
    main()
