#!/usr/bin/env python3
"""The dilated ASPP layer (3x3, dilation 18, 2048 -> 256 channels on a 30x40 map, 16 frames) on the stand-alone op: tile shapes
(key 4) x padded-filter-row skipping (key 11), split-K workspace on (key 2).  GPU box only."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from quber_amd import _lib
lib = _lib.load()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
B, H, W, Cin, Cout, k, d = 16, 30, 40, 2048, 256, 3, int(sys.argv[1]) if len(sys.argv) > 1 else 18
x = torch.randn(B, H, W, Cin, device="cuda"); w = torch.randn(Cout, Cin, k, k, device="cuda") / np.sqrt(Cin * k * k)
sc, sh = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda")
packed = torch.empty(Cout * k * k * Cin, device="cuda")
fl = 2.0 * B * H * W * Cin * k * k * Cout
lib.quber_set_tuning(2, 1)
ref = None
for skip in (1, 0):
    lib.quber_set_tuning(11, skip)
    for tile in (0, 2, 3, 1):
        for persist in ((0, 1) if tile in (0, 2) and not skip else (0,)):
            lib.quber_set_tuning(4, tile); lib.quber_set_tuning(13, persist)
            y = torch.empty(B, H, W, Cout, device="cuda")
            run = lambda: lib.quber_op_conv2d(p(x), B, H, W, Cin, p(w), Cout, k, 1, d, d, p(sc), p(sh), p(None), 1, p(packed), p(y), st)
            for _ in range(3):
                assert run() == 0, lib.quber_last_error()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            if ref is None:
                ref = y.clone()
            print(f"skip {skip} tile {tile} persistent {persist}: {ms:.3f} ms  {fl / ms / 1e9:.0f} algorithmic TFLOP/s  max |diff| {float((y - ref).abs().max()):.2e}", flush=True)
lib.quber_set_tuning(4, 0); lib.quber_set_tuning(11, 0); lib.quber_set_tuning(13, 1); lib.quber_set_tuning(2, 0)
