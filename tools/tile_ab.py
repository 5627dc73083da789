#!/usr/bin/env python3
"""Stand-alone convolution op with the tile shape forced (tuning key 4: 1 = 64x64, 3 = 128x64, 2 = 128x128) on the layers
with at most 64 output channels and on a residual 1x1; one-tile-per-block kernel (key 13 = 0).  GPU box only."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from quber_amd import _lib
lib = _lib.load()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
LAYERS = [("stem.conv3 3x3 32>64 @240x320 x2", 32, 240, 320, 32, 64, 3, 1, 1, False),
          ("res2.conv2 3x3 64>64 @120x160 x2", 32, 120, 160, 64, 64, 3, 1, 1, False),
          ("res2.conv1 1x1 256>64 @120x160 x2", 32, 120, 160, 256, 64, 1, 1, 1, False),
          ("decoder.project 1x1 256>48-ish (64) @120x160", 16, 120, 160, 256, 64, 1, 1, 1, False),
          ("res2.conv3 1x1 64>256 +res @120x160 x2", 32, 120, 160, 64, 256, 1, 1, 1, True),
          ("res3.conv3 1x1 128>512 +res @60x80 x2", 32, 60, 80, 128, 512, 1, 1, 1, True),
          ("res4.conv3 1x1 256>1024 +res @30x40 x2", 32, 30, 40, 256, 1024, 1, 1, 1, True),
          ("res5.conv3 1x1 512>2048 +res @30x40 x2", 32, 30, 40, 512, 2048, 1, 1, 1, True)]
lib.quber_set_tuning(13, 0)
for (name, B, H, W, Cin, Cout, k, s, d, res) in LAYERS:
    x = torch.randn(B, H, W, Cin, device="cuda"); w = torch.randn(Cout, Cin, k, k, device="cuda") / np.sqrt(Cin * k * k)
    sc, sh = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda")
    y = torch.empty(B, H, W, Cout, device="cuda"); r = torch.randn(B, H, W, Cout, device="cuda") if res else None
    packed = torch.empty(Cout * ((k * k * Cin + 31) // 32 * 32), device="cuda")
    fl = 2.0 * B * H * W * Cin * k * k * Cout
    out = []
    ref = None
    for tile in (1, 3, 2):
        lib.quber_set_tuning(4, tile)
        ts = []
        for rd in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                _lib.check(lib.quber_op_conv2d(p(x), B, H, W, Cin, p(w), Cout, k, s, d * (k // 2), d, p(sc), p(sh), p(r), 1, p(packed), p(y), st))
            e1.record(); torch.cuda.synchronize()
            if rd: ts.append(e0.elapsed_time(e1) / 3)
        if ref is None: ref = y.clone()
        out.append("tile %d: %.1f TF/s (%.3f ms, diff %.1e)" % (tile, fl / np.median(ts) / 1e9, np.median(ts), (y - ref).abs().max().item()))
    print(name, "|", " | ".join(out), flush=True)
lib.quber_set_tuning(4, 0); lib.quber_set_tuning(13, 1)
