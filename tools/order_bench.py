#!/usr/bin/env python3
"""Tile-order experiment: time + (under rocprofv3 --pmc FETCH_SIZE) traffic of a few layers per order knob."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quber_amd import _lib
from tools.conv_bench import LAYERS
lib = _lib.load()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
orders = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 1, 2]
for (name, B, H, W, Cin, Cout, k, s, d, res) in LAYERS:
    x = torch.randn(B, H, W, Cin, device="cuda"); w = torch.randn(Cout, Cin, k, k, device="cuda") / np.sqrt(Cin * k * k)
    y = torch.empty(B, H, W, Cout, device="cuda"); packed = torch.empty(Cout * ((k * k * Cin + 31) // 32 * 32), device="cuda")
    r = torch.randn(B, H, W, Cout, device="cuda") if res else None
    flops = 2.0 * B * H * W * Cin * k * k * Cout
    out = []
    for o in orders:
        lib.quber_set_tuning(1, o)
        ts = []
        for it in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(lib.quber_op_conv2d(p(x), B, H, W, Cin, p(w), Cout, k, s, d * (k // 2), d, None, None, p(r), 0, p(packed), p(y), st))
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        out.append("order %d: %.1f TF/s" % (o, flops / (min(ts) * 1e-3) / 1e12))
    print(name, "|", " | ".join(out), flush=True)
