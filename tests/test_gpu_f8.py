"""GPU: the exact-fp32 256 x 128-tile LDS-DMA kernel (csrc/conv_f8.hip) through the C ABI (quber_op_conv2d, quber_forward)
against the kernel it replaces on the wide 1x1 launches (conv_igemm.hip, option key 33 = 0) - BIT FOR BIT: same MFMA, same
k order inside a K-slice, same two-level accumulation - and against a float64 convolution.

Layers it runs in the network: the bottleneck / fusion 1x1 convolutions of maskrefiner/modeling/backbone/resnet.py:395-449,
472-485 and the position GEMMs of the wide Winograd layers (csrc/winograd.hip)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from quber_amd import _lib

pytestmark = pytest.mark.gpu

CASES = [
    # B, H, W, cin, cout, stride, affine, residual, relu
    (2, 40, 52, 64, 128, 1, True, False, True),        # two K-slices, ragged last pixel tile
    (1, 48, 64, 256, 320, 1, True, True, True),        # ragged channel tile, residual
    (2, 48, 64, 128, 256, 2, True, False, False),      # strided 1x1 (first block of a stage)
    (1, 64, 64, 96, 128, 1, False, False, False),      # no affine (a Winograd position GEMM's epilogue)
    (1, 30, 40, 2048, 256, 1, True, False, True),      # long K
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(str(int(v)) for v in c))
def test_f8_equals_the_128_tile_kernel_bit_for_bit(case):
    B, H, W, cin, cout, stride, affine, residual, relu = case
    lib = _lib.load()
    g = torch.Generator().manual_seed(77 + cin + cout)
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
        for mode in (2, 0):
            lib.quber_set_tuning(33, mode)
            y = torch.full((B, oh, ow, cout), float("nan"), device="cuda")
            _lib.check(lib.quber_op_conv2d(p(xd), B, H, W, cin, p(wd), cout, 1, stride, 0, 1, p(sd), p(hd), p(rd), int(relu), p(scratch), p(y),
                                           C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            torch.cuda.synchronize()
            outs[mode] = y
    finally:
        lib.quber_set_tuning(33, 1)
    assert torch.isfinite(outs[2]).all()
    assert torch.equal(outs[2], outs[0])
    assert float((outs[2].double().cpu() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))


def test_f8_runs_the_wide_gemms_of_the_network():
    """Exact fp32 network with the kernel on and off: the stage profile shows its launches (1x1 layers and Winograd position
    GEMMs), and the logits agree to the last bits (GroupNorm sums are fp64 atomics in both: their order is the only freedom)."""
    from quber_amd import arch, engine, synth
    from oracle import encode_np
    h, w, b = 256, 320, 4
    e = engine.Engine(engine.make_config(h, w, max_batch=b), "cuda:0")
    e.load_state_dict(arch.init_state_dict(seed=11, loud_heads=True))
    batch = synth.make_batch(5, b, h, w, 6)
    offs = np.stack([encode_np.encode_initial_masks(m) for m in batch["masks"]])
    bgr, dep, off = torch.from_numpy(batch["rgb"]).cuda(), torch.from_numpy(batch["depth"]).cuda(), torch.from_numpy(offs).cuda()
    outs, stages = {}, {}
    for mode in (0, 2):
        e.set_option(33, mode)
        e.profile_begin()
        outs[mode] = e.forward(bgr, dep, off).clone()
        stages[mode] = e.profile_end()
    e.close()
    assert torch.isfinite(outs[2]).all()
    assert "conv_gemm_f8" not in stages[0] and "wino_gemm_f8" not in stages[0]
    assert stages[2]["conv_gemm_f8"]["launches"] >= 10 and stages[2]["wino_gemm_f8"]["launches"] >= 4
    d = (outs[0] - outs[2]).abs()
    assert float(d.max()) < 2e-5 * max(1.0, float(outs[0].abs().max())), float(d.max())
