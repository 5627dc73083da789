#!/usr/bin/env python3
"""Randomised cross-check of the convolution launch structures on the stand-alone op: persistent launches for every tile
shape (tuning keys 13 = 2, 15 = 0) against the one-tile-per-block kernel (13 = 0) on random geometries - ragged M and N,
strides, dilations, K tails, residuals, both fp32-class arithmetic modes - and the dual-input 1x1 op against a float64
einsum.  GPU box only.  usage: conv_fuzz.py [cases] [seed]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quber_amd import _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
lib = _lib.load()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
lib.quber_set_tuning(2, 1)
lib.quber_set_tuning(15, 0)
worst, bad, worst64, n64, bad5, n_lean = 0.0, 0, 0.0, 0, 0, 0
for case in range(N):
    k = int(rng.choice([1, 1, 3]))
    Cin = int(rng.choice([4, 8, 32, 36, 64, 96, 128, 164, 256, 512, 1024]))
    if k > 1 and Cin < 8:
        Cin = 8            # (the op refuses fewer: the loader's tap stepping assumes >= 8 channels per tap)
    Cout = int(rng.choice([4, 32, 48, 64, 100, 128, 132, 256, 512, 1024]))
    stride = int(rng.choice([1, 1, 2]))
    dil = int(rng.choice([1, 1, 2, 6])) if k == 3 and stride == 1 else 1
    H, W = int(rng.integers(1, 70)), int(rng.integers(1, 90))
    B = int(rng.integers(1, 24))
    while B * H * W * max(Cin, Cout) > 3.0e8:
        B = max(1, B // 2); H = max(1, H // 2)
    res, relu, dt = bool(rng.integers(2)), int(rng.integers(2)), int(rng.choice([0, 0, 3]))
    pad = dil * (k // 2)
    OH, OW = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1, (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    g = torch.Generator(device="cuda").manual_seed(case)
    x = torch.randn(B, H, W, Cin, device="cuda", generator=g)
    w = torch.randn(Cout, Cin, k, k, device="cuda", generator=g) / np.sqrt(Cin * k * k)
    sc, sh = torch.rand(Cout, device="cuda", generator=g) + 0.5, torch.randn(Cout, device="cuda", generator=g)
    r = torch.randn(B, OH, OW, Cout, device="cuda", generator=g) if res else None
    packed = torch.empty(Cout * ((k * k * Cin + 31) // 32 * 32), device="cuda")
    outs = []
    lib.quber_set_tuning(12, dt)
    for mode in (0, 2):
        lib.quber_set_tuning(13, mode)
        y = torch.full((B, OH, OW, Cout), float("nan"), device="cuda")
        _lib.check(lib.quber_op_conv2d(p(x), B, H, W, Cin, p(w), Cout, k, stride, pad, dil, p(sc), p(sh), p(r), relu, p(packed), p(y), st))
        outs.append(y)
    # the one-tile-per-block kernel again with per-thread tap arithmetic (key 30 = 0): the LEAN loader (block-uniform taps, buffer
    # loads; taken when Cin % 32 == 0) must give the same bits
    lib.quber_set_tuning(13, 0)
    lib.quber_set_tuning(30, 0)
    y0 = torch.full((B, OH, OW, Cout), float("nan"), device="cuda")
    _lib.check(lib.quber_op_conv2d(p(x), B, H, W, Cin, p(w), Cout, k, stride, pad, dil, p(sc), p(sh), p(r), relu, p(packed), p(y0), st))
    lib.quber_set_tuning(30, 1)
    torch.cuda.synchronize()
    n_lean += int(Cin % 32 == 0)
    if not torch.equal(y0, outs[0]):
        bad5 += 1
        print("LEAN LOADER MISMATCH", dict(B=B, H=H, W=W, Cin=Cin, Cout=Cout, k=k, stride=stride, dil=dil, res=res, relu=relu, dt=dt),
              float((y0 - outs[0]).abs().max()), flush=True)
    if 2.0 * B * OH * OW * Cout * Cin * k * k < 4.0e9:          # small enough for a float64 reference on the device
        ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), None, stride, pad, dil).permute(0, 2, 3, 1)
        ref = ref * sc.double() + sh.double()
        if res:
            ref = ref + r.double()
        if relu:
            ref = ref.relu()
        e64 = float((outs[0].double() - ref).abs().max()) / max(1.0, float(ref.abs().max()))
        worst64 = max(worst64, e64)
        n64 += 1
        if not (e64 < 4e-6):
            bad += 1
            print("MISMATCH vs float64", dict(B=B, H=H, W=W, Cin=Cin, Cout=Cout, k=k, stride=stride, dil=dil, res=res, relu=relu, dt=dt), e64, flush=True)
    ok = bool(torch.isfinite(outs[1]).all())
    err = float((outs[0] - outs[1]).abs().max()) / max(1.0, float(outs[0].abs().max())) if ok else float("inf")
    worst = max(worst, err)
    if not ok or err > 4e-6:
        bad += 1
        print("MISMATCH", dict(B=B, H=H, W=W, Cin=Cin, Cout=Cout, k=k, stride=stride, dil=dil, res=res, relu=relu, dt=dt), err, flush=True)
print(f"conv2d: {N} random cases, persistent vs one-tile-per-block: {bad} mismatches, worst relative difference {worst:.2e}; "
      f"{n64} of them also against a float64 convolution: worst relative error {worst64:.2e}; "
      f"LEAN loader vs per-thread tap arithmetic ({n_lean} eligible cases): {bad5} bit mismatches")

# dual-input 1x1 against float64
worst2, bad2 = 0.0, 0
ones_cache = {}
for case in range(N // 3):
    mid, cin = int(rng.choice([32, 64, 128, 256, 512])), int(rng.choice([32, 64, 256, 512, 1024]))
    cout = int(rng.choice([64, 128, 256, 512, 1024, 2048]))
    stride = int(rng.choice([1, 2]))
    oh, ow, B = int(rng.integers(1, 40)), int(rng.integers(1, 50)), int(rng.integers(1, 17))
    h2, w2 = (oh - 1) * stride + 1 + int(rng.integers(0, stride)), (ow - 1) * stride + 1 + int(rng.integers(0, stride))
    dt = int(rng.choice([0, 3]))
    g = torch.Generator(device="cuda").manual_seed(1000 + case)
    y = torch.randn(B, oh, ow, mid, device="cuda", generator=g)
    x = torch.randn(B, h2, w2, cin, device="cuda", generator=g)
    w = torch.randn(cout, mid + cin, device="cuda", generator=g) / np.sqrt(mid + cin)
    sh = torch.randn(cout, device="cuda", generator=g)
    ones = torch.ones(cout, device="cuda")
    out = torch.full((B, oh, ow, cout), float("nan"), device="cuda")
    lib.quber_set_tuning(12, dt)
    lib.quber_set_tuning(13, 1)
    _lib.check(lib.quber_op_conv1x1_dual(p(y), p(x), B, oh, ow, mid, h2, w2, cin, stride, p(w), p(sh), p(ones), cout, 1, p(out), st))
    ref = (torch.einsum("bhwc,oc->bhwo", y.double(), w[:, :mid].double()) +
           torch.einsum("bhwc,oc->bhwo", x[:, ::stride, ::stride][:, :oh, :ow].double(), w[:, mid:].double()) + sh.double()).relu()
    err = float((out.double() - ref).abs().max()) / max(1.0, float(ref.abs().max()))
    worst2 = max(worst2, err)
    if not (err < 3e-6):
        bad2 += 1
        print("DUAL MISMATCH", dict(B=B, oh=oh, ow=ow, mid=mid, cin=cin, cout=cout, stride=stride, dt=dt), err, flush=True)
print(f"conv1x1_dual: {N // 3} random cases against float64: {bad2} mismatches, worst relative error {worst2:.2e}")

# Winograd F(m x m, 3x3) op (grouped GEMM through the persistent launch) against the direct kernel on random geometries
worst3, bad3, n3 = {2: 0.0, 4: 0.0, 6: 0.0}, 0, 0
lib.quber_set_tuning(12, 0)
lib.quber_set_tuning(13, 1)
for case in range(N // 5):
    Cin = int(rng.choice([64, 128, 256, 512, 1024]))
    Cout = int(rng.choice([32, 64, 128, 256, 512]))
    dil = int(rng.choice([1, 1, 1, 2, 3, 6]))
    H, W, B = int(rng.integers(3, 64)), int(rng.integers(3, 80)), int(rng.integers(1, 9))
    m = int(rng.choice([2, 4, 6]))
    P = (m + 2) ** 2
    tiles = B * dil * dil * ((-(-H // dil) + m - 1) // m) * ((-(-W // dil) + m - 1) // m)
    if P * tiles * (Cin + Cout) > 4.0e8:
        continue
    g = torch.Generator(device="cuda").manual_seed(5000 + case)
    x = torch.randn(B, H, W, Cin, device="cuda", generator=g)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) / np.sqrt(Cin * 9)
    sc, sh = torch.rand(Cout, device="cuda", generator=g) + 0.5, torch.randn(Cout, device="cuda", generator=g)
    u = torch.empty(P * Cout * Cin, device="cuda")
    ws = torch.empty(P * tiles * (Cin + Cout), device="cuda")
    y = torch.full((B, H, W, Cout), float("nan"), device="cuda")
    rc = lib.quber_op_conv3x3_winograd(p(x), B, H, W, Cin, p(w), Cout, dil, m, p(sc), p(sh), 1, p(u), p(ws), ws.numel(), p(y), st)
    if rc != 0:
        continue            # geometry the path does not take (refused loudly)
    packed = torch.empty(Cout * 9 * Cin, device="cuda")
    yd = torch.empty_like(y)
    _lib.check(lib.quber_op_conv2d(p(x), B, H, W, Cin, p(w), Cout, 3, 1, dil, dil, p(sc), p(sh), p(None), 1, p(packed), p(yd), st))
    err = float((y - yd).abs().max()) / max(1.0, float(yd.abs().max())) if bool(torch.isfinite(y).all()) else float("inf")
    n3 += 1
    worst3[m] = max(worst3[m], err)
    if not (err < {2: 1e-5, 4: 2e-5, 6: 1e-4}[m]):
        bad3 += 1
        print("WINOGRAD MISMATCH", dict(B=B, H=H, W=W, Cin=Cin, Cout=Cout, dil=dil, m=m), err, flush=True)
print(f"conv3x3_winograd: {n3} random cases against the direct kernel: {bad3} mismatches, worst relative difference "
      f"F(2x2) {worst3[2]:.1e}, F(4x4) {worst3[4]:.1e}, F(6x6) {worst3[6]:.1e}")

# the single-kernel Winograd layer (wino_fused.hip: both block shapes) against the direct kernel AND the three-kernel pipeline
worst4, worst4p, bad4, n4 = 0.0, 0.0, 0, 0
for case in range(N // 4):
    Cin = int(rng.choice([32, 64, 96, 128, 160]))
    Cout = int(rng.choice([32, 64, 96, 128, 192, 256]))
    dil = int(rng.choice([1, 1, 1, 2, 3, 5]))
    H, W, B = int(rng.integers(3, 70)), int(rng.integers(3, 90)), int(rng.integers(1, 9))
    tiles = B * dil * dil * ((-(-H // dil) + 3) // 4) * ((-(-W // dil) + 3) // 4)
    g = torch.Generator(device="cuda").manual_seed(9000 + case)
    x = torch.randn(B, H, W, Cin, device="cuda", generator=g)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) / np.sqrt(Cin * 9)
    affine = bool(rng.integers(2))
    sc = torch.rand(Cout, device="cuda", generator=g) + 0.5 if affine else None
    sh = torch.randn(Cout, device="cuda", generator=g) if affine else None
    relu = int(rng.integers(2))
    u = torch.empty(36 * Cout * Cin, device="cuda")
    ws = torch.empty(36 * tiles * (Cin + Cout) + 36 * Cout * Cin, device="cuda")
    ys = []
    for fused in (0, 1):
        lib.quber_set_tuning(25, fused)
        y = torch.full((B, H, W, Cout), float("nan"), device="cuda")
        _lib.check(lib.quber_op_conv3x3_winograd(p(x), B, H, W, Cin, p(w), Cout, dil, 4, p(sc), p(sh), relu, p(u), p(ws), ws.numel(), p(y), st))
        ys.append(y)
    lib.quber_set_tuning(25, 1)
    packed = torch.empty(Cout * 9 * Cin, device="cuda")
    yd = torch.empty_like(ys[1])
    _lib.check(lib.quber_op_conv2d(p(x), B, H, W, Cin, p(w), Cout, 3, 1, dil, dil, p(sc), p(sh), p(None), relu, p(packed), p(yd), st))
    scale = max(1.0, float(yd.abs().max()))
    ok = bool(torch.isfinite(ys[1]).all()) and not torch.equal(ys[0], ys[1])
    err = float((ys[1] - yd).abs().max()) / scale if ok else float("inf")
    errp = float((ys[1] - ys[0]).abs().max()) / scale if ok else float("inf")
    n4 += 1
    worst4, worst4p = max(worst4, err), max(worst4p, errp)
    if not (err < 2e-5 and errp < 1e-5):
        bad4 += 1
        print("SINGLE-KERNEL WINOGRAD MISMATCH", dict(B=B, H=H, W=W, Cin=Cin, Cout=Cout, dil=dil, affine=affine, relu=relu), err, errp, flush=True)
print(f"conv3x3_winograd, single kernel: {n4} random cases: {bad4} mismatches, worst relative difference to the direct kernel {worst4:.1e}, "
      f"to the three-kernel pipeline {worst4p:.1e}")
lib.quber_set_tuning(12, 0); lib.quber_set_tuning(13, 1); lib.quber_set_tuning(15, 256); lib.quber_set_tuning(2, 0); lib.quber_set_tuning(25, 1)
sys.exit(1 if bad or bad2 or bad3 or bad4 or bad5 else 0)
