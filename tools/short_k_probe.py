#!/usr/bin/env python3
"""One short-K residual 1x1 layer, stand-alone, for counter passes and timing: tools/short_k_probe.py [layer] [iters] [res 0|1] [affine 0|1] [tuning] [sets]
(sets > 1: that many (x, residual, y) tensor sets used in rotation, so that no launch finds its operands in the 256 MB infinity cache)
layers: res2 (64 -> 256 @120x160 x16), res3 (128 -> 512 @60x80 x16), res4 (256 -> 1024 @30x40 x16).  GPU box only."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quber_amd import _lib  # noqa: E402

SHAPES = {"res2": (16, 120, 160, 64, 256), "res3": (16, 60, 80, 128, 512), "res4": (16, 30, 40, 256, 1024), "res2c1": (16, 120, 160, 256, 64)}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "res2"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    with_res = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    affine = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    B, H, W, Cin, Cout = SHAPES[name]
    lib = _lib.load()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    sets = int(sys.argv[6]) if len(sys.argv) > 6 else 1
    x = torch.randn(B, H, W, Cin, device="cuda")
    w = torch.randn(Cout, Cin, 1, 1, device="cuda") / np.sqrt(Cin)
    sc = torch.rand(Cout, device="cuda") + 0.5 if affine else None
    sh = torch.randn(Cout, device="cuda") if affine else None
    r = torch.randn(B, H, W, Cout, device="cuda") if with_res else None
    y = torch.empty(B, H, W, Cout, device="cuda")
    packed = torch.empty(Cout * Cin, device="cuda")
    for kv in (sys.argv[5].split(",") if len(sys.argv) > 5 and sys.argv[5] else []):
        lib.quber_set_tuning(int(kv.split("=")[0]), int(kv.split("=")[1]))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    xs = [x] + [torch.randn_like(x) for _ in range(sets - 1)]
    rs = [r] + [torch.randn_like(r) if with_res else None for _ in range(sets - 1)]
    ys = [y] + [torch.empty_like(y) for _ in range(sets - 1)]
    for it in range(iters + 3):
        if it == 3:
            ev[0].record()
        k = it % sets
        _lib.check(lib.quber_op_conv2d(p(xs[k]), B, H, W, Cin, p(w), Cout, 1, 1, 0, 1, p(sc), p(sh), p(rs[k]), 1, p(packed), p(ys[k]), st))
    ev[1].record()
    torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) / iters * 1e3
    traffic = 4.0 * B * H * W * (Cin + Cout * (2 if with_res else 1))
    print(f"{name} res={with_res} affine={affine} sets={sets}: {us:.1f} us, {2e-6 * B * H * W * Cin * Cout / us:.1f} TFLOP/s, {traffic / us * 1e-6:.2f} TB/s of algorithmic traffic")


if __name__ == "__main__":
    main()
