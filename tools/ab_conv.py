#!/usr/bin/env python3
"""A/B two builds of libquber_hip.so on the stand-alone conv op, interleaved rounds in one process
(cdna_hip_programming.md rule 24).  usage: ab_conv.py <other libquber_hip.so>"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quber_amd import _lib  # noqa: E402
from tools.conv_bench import LAYERS  # noqa: E402

libs = {"old": C.CDLL(sys.argv[1]), "new": C.CDLL(_lib.LIB_PATH)}
if len(sys.argv) > 2:       # extra "key=value" tuning knobs applied to the first library only
    for kv in sys.argv[2:]:
        k, v = kv.split("=")
        libs["old"].quber_set_tuning(int(k), int(v))
for lib in libs.values():
    lib.quber_op_conv2d.restype = C.c_int
    lib.quber_op_conv2d.argtypes = _lib.SIGNATURES["quber_op_conv2d"][1]
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
for (name, B, H, W, Cin, Cout, k, s, d, res) in LAYERS:
    x = torch.randn(B, H, W, Cin, device="cuda")
    w = torch.randn(Cout, Cin, k, k, device="cuda") / np.sqrt(Cin * k * k)
    sc, sh = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda")
    y = torch.empty(B, H, W, Cout, device="cuda")
    packed = torch.empty(Cout * ((k * k * Cin + 31) // 32 * 32), device="cuda")
    r = torch.randn(B, H, W, Cout, device="cuda") if res else None
    fl = 2.0 * B * H * W * Cin * k * k * Cout
    ts = {n: [] for n in libs}
    for rd in range(7):
        for n, lib in libs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                assert lib.quber_op_conv2d(p(x), B, H, W, Cin, p(w), Cout, k, s, d * (k // 2), d, p(sc), p(sh), p(r), 1,
                                           p(packed), p(y), st) == 0
            e1.record()
            torch.cuda.synchronize()
            if rd:
                ts[n].append(e0.elapsed_time(e1) / 3)
    print(name, "|", " | ".join("%s %.1f TF/s" % (n, fl / np.median(v) / 1e9) for n, v in ts.items()), flush=True)
