#!/usr/bin/env python3
"""The dilated ASPP layer (3x3, 2048 -> 256 channels, 16 frames) on the stand-alone op with padded filter rows skipped, with and
without the padded filter COLUMNS skipped as well (option key 43, ConvP::zones).  usage: tools/zone_ab.py [H W [dil [frames]]]   GPU box only."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from quber_amd import _lib
lib = _lib.load()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
a = [int(v) for v in sys.argv[1:]]
H, W = (a[0], a[1]) if len(a) >= 2 else (30, 40)
d = a[2] if len(a) > 2 else 18
B = a[3] if len(a) > 3 else 16
Cin, Cout = 2048, 256
x = torch.randn(B, H, W, Cin, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") / np.sqrt(Cin * 9)
sc, sh = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda")
packed = torch.empty(Cout * 9 * Cin, device="cuda")
fl = 2.0 * B * H * W * Cin * 9 * Cout
lib.quber_set_tuning(2, int(os.environ.get("WS", "1"))); lib.quber_set_tuning(11, 1)
ref = None
for rep in range(2):
    for zones in (0, 1):
        lib.quber_set_tuning(43, zones)
        y = torch.empty(B, H, W, Cout, device="cuda")
        run = lambda: lib.quber_op_conv2d(p(x), B, H, W, Cin, p(w), Cout, 3, 1, d, d, p(sc), p(sh), p(None), 1, p(packed), p(y), st)
        for _ in range(3):
            assert run() == 0, lib.quber_last_error()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        if ref is None:
            ref = y.clone()
        print(f"{B} x {H}x{W} d={d} columns skipped {zones}: {ms:.3f} ms  {fl / ms / 1e9:.0f} algorithmic TFLOP/s  equal to the first {bool(torch.equal(y, ref))} (max |diff| {float((y - ref).abs().max()):.1e})", flush=True)
lib.quber_set_tuning(43, 1); lib.quber_set_tuning(11, 0); lib.quber_set_tuning(2, 0)
