#!/usr/bin/env python3
"""A/B micro-benchmark of the convolution main-loop variants on representative refiner layers
(interleaved rounds in one process, random data; cdna_hip_programming.md rule 24/25)."""
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
    ("fusion_res2.conv0 3x3 256>256 @120x160", 16, 120, 160, 256, 256, 3, 1, 1, False),
    ("res5.conv2 3x3 d2 512>512 @30x40 (2 streams)", 32, 30, 40, 512, 512, 3, 1, 2, False),
    ("res2.conv3 1x1 64>256 +res @120x160 (2 streams)", 32, 120, 160, 64, 256, 1, 1, 1, True),
    ("res4.conv3 1x1 256>1024 +res @30x40 (2 streams)", 32, 30, 40, 256, 1024, 1, 1, 1, True),
    ("fusion_res5.conv 1x1 4096>2048 @30x40", 16, 30, 40, 4096, 2048, 1, 1, 1, False),
    ("head 3x3 128>128 @120x160", 16, 120, 160, 128, 128, 3, 1, 1, False),
    ("aspp 3x3 d12 2048>256 @30x40", 16, 30, 40, 2048, 256, 3, 1, 12, False),
]


def main():
    variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2,3,4,5,6,7".split(","))]
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    lib = _lib.load()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    print("| layer | " + " | ".join(f"v{v} TF/s" for v in variants) + " |")
    print("|---|" + "---|" * len(variants))
    for (name, B, H, W, Cin, Cout, k, s, d, res) in LAYERS:
        x = torch.randn(B, H, W, Cin, device="cuda")
        w = torch.randn(Cout, Cin, k, k, device="cuda") / np.sqrt(Cin * k * k)
        sc, sh = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda")
        pad = d * (k // 2)
        y = torch.empty(B, H, W, Cout, device="cuda")
        r = torch.randn(B, H, W, Cout, device="cuda") if res else None
        packed = torch.empty(Cout * ((k * k * Cin + 31) // 32 * 32), device="cuda")
        flops = 2.0 * B * H * W * Cin * k * k * Cout
        best = {v: [] for v in variants}
        ref = None
        for rd in range(rounds + 1):
            for v in variants:
                lib.quber_set_tuning(2, v)     # variant 1 = with the split-K / tail-split workspace
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    _lib.check(lib.quber_op_conv2d(p(x), B, H, W, Cin, p(w), Cout, k, s, pad, d, p(sc), p(sh), p(r), 1,
                                                   p(packed), p(y), st))
                e1.record()
                torch.cuda.synchronize()
                if rd == 0:
                    if ref is None:
                        ref = y.clone()
                    else:
                        assert v >= 8 or torch.allclose(y, ref, rtol=1e-4, atol=1e-4), f"variant {v} differs on {name}"
                else:
                    best[v].append(e0.elapsed_time(e1) / 3)
        print(f"| {name} | " + " | ".join("%.1f" % (flops / (np.median(best[v]) * 1e-3) / 1e12) for v in variants) + " |")
    lib.quber_set_tuning(2, 0)


if __name__ == "__main__":
    main()
