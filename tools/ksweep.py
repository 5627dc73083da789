#!/usr/bin/env python3
"""Per-tile overhead of the implicit GEMM: time of a 1x1 convolution (M = B*H*W rows, N output channels) as a function of K,
tile-per-block against persistent launch.  Fits t = tiles/slots * (a + b * nk).  GPU box only.
usage: ksweep.py [N] [B] [tuning knobs]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quber_amd import _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
H, W = 120, 160
lib = _lib.load()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
lib.quber_set_tuning(2, 1)
lib.quber_set_tuning(15, 0)
lib.quber_set_tuning(5, 0)
for kv in (sys.argv[3].split(",") if len(sys.argv) > 3 else []):
    lib.quber_set_tuning(int(kv.split("=")[0]), int(kv.split("=")[1]))
M = B * H * W
tiles = ((M + 127) // 128) * ((N + 127) // 128)
print(f"M {M} N {N}: {tiles} tiles of 128x128 = {tiles / 768:.2f} rounds of 768")
rows = []
for K in (32, 64, 128, 256, 512, 1024, 2048):
    x = torch.randn(B, H, W, K, device="cuda")
    w = torch.randn(N, K, 1, 1, device="cuda") / np.sqrt(K)
    y = torch.empty(B, H, W, N, device="cuda")
    packed = torch.empty(N * K, device="cuda")
    lib.quber_set_tuning(4, 2)       # 128x128 tiles
    ts = {0: [], 1: []}
    for rd in range(6):
        for persist in (0, 1):
            lib.quber_set_tuning(13, persist)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                _lib.check(lib.quber_op_conv2d(p(x), B, H, W, K, p(w), N, 1, 1, 0, 1, p(None), p(None), p(None), 0, p(packed), p(y), st))
            e1.record()
            torch.cuda.synchronize()
            if rd:
                ts[persist].append(e0.elapsed_time(e1) / 3)
    t0, t1 = float(np.median(ts[0])), float(np.median(ts[1]))
    fl = 2.0 * M * N * K
    rows.append((K // 32, t0, t1))
    print(f"K {K:5d} nk {K // 32:3d}: tile-per-block {t0 * 1e3:8.1f} us {fl / t0 / 1e9:6.1f} TF/s | persistent {t1 * 1e3:8.1f} us {fl / t1 / 1e9:6.1f} TF/s"
          f" | per tile-round {t0 * 1e3 / (tiles / 768):6.2f} / {t1 * 1e3 / (tiles / 768):6.2f} us", flush=True)
nk = np.array([r[0] for r in rows], float)
for name, col in (("tile-per-block", 1), ("persistent", 2)):
    t = np.array([r[col] for r in rows]) * 1e3 / (tiles / 768)
    b, a = np.polyfit(nk, t, 1)
    print(f"{name}: per tile-round {a:.2f} us + {b:.3f} us per K-slice (pure MFMA at 2.4 GHz: {3 * 4096 / 2400:.3f} us per slice of 3 resident blocks)")
lib.quber_set_tuning(4, 0); lib.quber_set_tuning(13, 1); lib.quber_set_tuning(2, 0); lib.quber_set_tuning(5, 1)
