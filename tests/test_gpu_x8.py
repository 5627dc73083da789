"""GPU: the bf16x3 256 x 128-tile LDS-DMA kernel (csrc/conv_x8.hip: weights pre-split into three bf16 planes at plan time,
activations split fragment by fragment in registers) through the C ABI, against the kernel it replaces on the wide 1x1 launches
(conv_igemm.hip DT 3, option key 35 = 0) - the same six partial products in the same order, so BIT FOR BIT where that kernel runs
bf16x3 itself - and against float64.

Layers: the bottleneck / fusion 1x1 convolutions of maskrefiner/modeling/backbone/resnet.py:395-449, 472-485 and the position GEMMs
of the wide Winograd layers, in the fp32-equivalent bf16x3 mode (quber_config.compute_dtype 3).

What this file is: KERNEL AGAINST KERNEL (plus a float64 convolution of the same operands) - it shows that conv_x8.hip and the
128-tile kernels are the same arithmetic, not that either equals the reference path.  The tie to the oracle
(oracle/network_torch.py, the restatement of model.py / resnet.py) is at network level: the `*-bf16x3` ids of
tests/test_gpu_loud_parity.py::test_benchmarked_plan_float64_anchor and ::test_benchmarked_plan_taps_heads_and_instances run the plan that contains
these launches and hold every tap and head to the 1e-4 / float64-anchor bars."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from quber_amd import _lib

pytestmark = pytest.mark.gpu

CASES = [
    # B, H, W, cin, cout, stride, affine, residual, relu, the 128-tile path runs bf16x3 too (it keeps exact fp32 for K <= 256 on 64 x 64 tiles)
    (2, 40, 52, 64, 128, 1, True, False, True, False),        # two K-slices = one accumulation chunk, ragged last pixel tile
    (1, 48, 64, 256, 320, 1, True, True, True, False),        # ragged channel tile, residual
    (2, 48, 64, 288, 256, 2, True, False, False, True),       # strided 1x1, an odd number of K-slices (the last chunk is one slice)
    (2, 64, 64, 512, 256, 1, False, False, False, True),      # no affine (a Winograd position GEMM's epilogue)
    (1, 30, 40, 2048, 256, 1, True, False, True, True),       # long K
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(str(int(v)) for v in c))
def test_x8_equals_the_128_tile_bf16x3_kernel(case):
    B, H, W, cin, cout, stride, affine, residual, relu, same_arithmetic = case
    lib = _lib.load()
    g = torch.Generator().manual_seed(99 + cin + cout)
    x = torch.randn((B, H, W, cin), generator=g)
    w = torch.randn((cout, cin, 1, 1), generator=g) / cin ** 0.5
    scale = (0.5 + torch.rand(cout, generator=g)) if affine else None
    shift = torch.randn(cout, generator=g) * 0.3 if affine else None
    oh, ow = (H - 1) // stride + 1, (W - 1) // stride + 1
    res = torch.randn((B, oh, ow, cout), generator=g) if residual else None
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), None, stride).permute(0, 2, 3, 1)
    if affine:
        ref = ref * scale.double() + shift.double()
    if residual:
        ref = ref + res.double()
    if relu:
        ref = ref.clamp_min(0)
    dev = lambda t_: t_.cuda() if t_ is not None else None
    p = lambda t_: C.c_void_p(t_.data_ptr()) if t_ is not None else None
    xd, wd, sd, hd, rd = dev(x), dev(w), dev(scale), dev(shift), dev(res)
    scratch = torch.empty(cout * cin, device="cuda")
    outs = {}
    try:
        lib.quber_set_tuning(12, 3)            # the stand-alone op in the bf16x3 mode
        for mode in (2, 0):
            lib.quber_set_tuning(35, mode)
            y = torch.full((B, oh, ow, cout), float("nan"), device="cuda")
            _lib.check(lib.quber_op_conv2d(p(xd), B, H, W, cin, p(wd), cout, 1, stride, 0, 1, p(sd), p(hd), p(rd), int(relu), p(scratch), p(y),
                                           C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            torch.cuda.synchronize()
            outs[mode] = y
    finally:
        lib.quber_set_tuning(35, 1)
        lib.quber_set_tuning(12, 0)
    assert torch.isfinite(outs[2]).all()
    bound = 2e-6 * max(1.0, float(ref.abs().max()))           # bf16x3 = fp32-equivalent: the bar of the exact mode's kernels
    assert float((outs[2].double().cpu() - ref).abs().max()) < bound
    assert float((outs[0].double().cpu() - ref).abs().max()) < bound
    if same_arithmetic:
        assert torch.equal(outs[2], outs[0])


@pytest.mark.parametrize("case", [
    # B, oh, ow, mid, cin, cout, stride: bottleneck conv3 + projection shortcut as ONE GEMM over two inputs (conv_persist.hip launch_conv_dual)
    # ... , the 128-tile kernel sums K in the same chunks (with few tiles or K >= 32 slices it shares the K of its ragged round between blocks: other partial sums)
    (4, 60, 80, 128, 256, 512, 2, True),        # res3.0: strided second input, K = 384
    (3, 15, 21, 256, 512, 1024, 2, False),      # ragged M, odd input size
    (2, 30, 40, 512, 1024, 2048, 1, False),     # res5.0 geometry, two frames: K = 1536
    (8, 30, 40, 512, 1024, 2048, 1, False),     # ... (K of 48 slices: the persistent 128-tile kernel shares the K of its ragged round between blocks)
], ids=lambda c: "x".join(str(int(v)) for v in c))
def test_x8_dual_input_equals_the_128_tile_bf16x3_kernel(case):
    """The dual-input form of the kernel: K-slices [0, mid / 32) from the bottleneck's conv2 output, the rest from the block's input sampled at
    the stride - bit for bit the 128-tile persistent kernel's bf16x3 result (same six products, same order, same two-level accumulation), and
    float64-close."""
    B, oh, ow, mid, cin, cout, stride, same_arithmetic = case
    lib = _lib.load()
    g = torch.Generator().manual_seed(17)
    h2, w2 = (oh - 1) * stride + 1, (ow - 1) * stride + 1
    y = torch.randn(B, oh, ow, mid, generator=g)
    x = torch.randn(B, h2, w2, cin, generator=g)
    w = torch.randn(cout, mid + cin, generator=g) / np.sqrt(mid + cin)
    sh = torch.randn(cout, generator=g)
    ref = (torch.einsum("bhwc,oc->bhwo", y.double(), w[:, :mid].double()) +
           torch.einsum("bhwc,oc->bhwo", x[:, ::stride, ::stride].double(), w[:, mid:].double()) + sh.double()).relu()
    p = lambda t: C.c_void_p(t.data_ptr())
    yd, xd, wd, shd, ones = y.cuda(), x.cuda(), w.cuda(), sh.cuda(), torch.ones(cout, device="cuda")
    outs = {}
    lib.quber_set_tuning(2, 1)
    lib.quber_set_tuning(12, 3)
    lib.quber_set_tuning(15, 0)
    try:
        for mode in (2, 0):
            lib.quber_set_tuning(35, mode)
            out = torch.full((B, oh, ow, cout), float("nan"), device="cuda")
            _lib.check(lib.quber_op_conv1x1_dual(p(yd), p(xd), B, oh, ow, mid, h2, w2, cin, stride, p(wd), p(shd), p(ones), cout, 1,
                                                 p(out), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            torch.cuda.synchronize()
            outs[mode] = out
    finally:
        lib.quber_set_tuning(35, 1)
        lib.quber_set_tuning(15, 256)
        lib.quber_set_tuning(12, 0)
        lib.quber_set_tuning(2, 0)
    bound = 2e-6 * max(1.0, float(ref.abs().max()))
    assert float((outs[2].double().cpu() - ref).abs().max()) < bound
    assert float((outs[0].double().cpu() - ref).abs().max()) < bound
    if same_arithmetic:
        assert torch.equal(outs[2], outs[0])


def test_x8_runs_the_wide_gemms_of_the_bf16x3_network():
    """bf16x3 network with the kernel on and off: the stage profile shows its launches (1x1 layers and Winograd position GEMMs), and
    the logits agree far inside the mode's 1e-4 bar (the persistent kernels it replaces share the K of ragged tiles between blocks:
    another association of the same fp32 sums)."""
    from quber_amd import arch, engine, synth
    from oracle import encode_np
    h, w, b = 256, 320, 4
    qc = engine.make_config(h, w, max_batch=b)
    qc.compute_dtype = 3
    e = engine.Engine(qc, "cuda:0")
    e.load_state_dict(arch.init_state_dict(seed=11, loud_heads=True))
    batch = synth.make_batch(5, b, h, w, 6)
    offs = np.stack([encode_np.encode_initial_masks(m) for m in batch["masks"]])
    bgr, dep, off = torch.from_numpy(batch["rgb"]).cuda(), torch.from_numpy(batch["depth"]).cuda(), torch.from_numpy(offs).cuda()
    outs, stages = {}, {}
    for mode in (0, 2):
        e.set_option(35, mode)
        e.profile_begin()
        outs[mode] = e.forward(bgr, dep, off).clone()
        stages[mode] = e.profile_end()
    e.close()
    assert torch.isfinite(outs[2]).all()
    assert "conv_gemm_x8" not in stages[0] and "wino_gemm_x8" not in stages[0]
    assert stages[2]["conv_gemm_x8"]["launches"] >= 10 and stages[2]["wino_gemm_x8"]["launches"] >= 4
    d = (outs[0] - outs[2]).abs()
    assert float(d.max()) < 2e-5 * max(1.0, float(outs[0].abs().max())), float(d.max())
