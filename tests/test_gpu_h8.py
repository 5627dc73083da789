"""GPU: the LDS-DMA convolution kernels of the fp16 data path (csrc/conv_h8.hip: the 256 x 256 / 256 x 128-tile DMA-gather kernels, and
the patch kernels of the undilated 3x3 layers - conv_h8p / h8w / h8s, option key 38) through the C ABI
(quber_op_conv2d_f16), against a float32 CPU convolution of the same fp16 operands (torch, the oracle's arithmetic for
one layer: oracle/network_torch.py runs the reference's F.conv2d) and against the 128 x 128 kernel it replaces
(conv_igemm.hip, option key 31 = 0).

Layers this kernel runs in the network: maskrefiner/modeling/backbone/resnet.py:395-449 (res4 / res5 bottlenecks),
:472-485 (fusion convolutions, bias + GroupNorm).  Tolerance: the result is rounded to fp16 once (2^-11 relative) after an
fp32 accumulation over K <= 4608 fp16 products; 3e-3 of the layer's output scale covers both."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from quber_amd import _lib

pytestmark = pytest.mark.gpu


def pack(w, kmode):
    """OIHW -> [cout][K] in the kernels' K order (csrc/plan.hip emit_conv)."""
    co, ci, kh, kw = w.shape
    t = w.permute(0, 2, 3, 1).reshape(co, kh * kw, ci)               # [co][tap][ci]
    if kmode == 0:                                                    # tap-major, rows zero-filled to a multiple of 64 (one K-tile)
        k = kh * kw * ci
        return F.pad(t.reshape(co, k), (0, (k + 63) // 64 * 64 - k)).contiguous()
    return t.reshape(co, kh * kw, ci // 64, 64).permute(0, 2, 1, 3).reshape(co, kh * kw * ci).contiguous()


def run(lib, x, wp, cout, k, stride, pad, dil, kmode, scale, shift, res, relu, groups):
    B, H, W, cin = x.shape
    oh = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
    ow = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    y = torch.full((B, oh, ow, cout), float("nan"), dtype=torch.float16, device="cuda")
    sums = torch.zeros((B, groups, 2), dtype=torch.float64, device="cuda") if groups else None
    p = lambda t_: C.c_void_p(t_.data_ptr()) if t_ is not None else None
    _lib.check(lib.quber_op_conv2d_f16(p(x), B, H, W, cin, p(wp), cout, k, stride, pad, dil, kmode, p(scale), p(shift), p(res),
                                       int(relu), p(sums), groups, p(y), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    return y, sums


CASES = [
    # B, H, W, cin, cout, k, stride, pad, dil, kmode, affine, residual, relu, norm groups, key 31
    (2, 40, 52, 64, 256, 3, 1, 1, 1, 1, True, False, True, 0, 1),        # ragged last pixel tile, padded borders
    (3, 40, 52, 128, 256, 3, 1, 2, 2, 1, True, False, False, 32, 1),     # dilated; GroupNorm sums, tiles across image boundaries
    (2, 33, 47, 128, 512, 1, 1, 0, 1, 0, True, True, True, 0, 1),        # 1x1 + residual + ReLU (bottleneck conv3)
    (2, 48, 64, 256, 256, 1, 2, 0, 1, 0, True, False, True, 0, 1),       # strided 1x1 (first block of a stage)
    (1, 36, 44, 64, 320, 3, 1, 1, 1, 1, True, False, False, 0, 1),       # ragged channel tile
    (2, 32, 40, 128, 128, 3, 1, 1, 1, 1, True, False, False, 32, 1),     # 128 channels (the 256 x 128-tile kernel): 4 channels per norm group
    (3, 37, 45, 64, 128, 3, 1, 2, 2, 1, True, True, True, 0, 1),         # 128 channels, dilated, residual, tiles across image boundaries
    (1, 50, 70, 320, 128, 1, 1, 0, 1, 0, True, False, True, 32, 1),      # 128 channels, 1x1, five K-tiles
    (2, 32, 40, 128, 192, 3, 1, 1, 1, 1, True, False, False, 0, 2),      # 192 channels on 256-channel tiles (key 31 = 2)
    (1, 64, 64, 512, 512, 3, 1, 4, 4, 1, True, False, True, 0, 1),       # res5-like: K = 4608
    (1, 20, 24, 2048, 256, 1, 1, 0, 1, 0, True, False, True, 32, 1),     # long 1x1 (ASPP convs.0 / fusion_res5-like)
    (2, 40, 52, 64, 64, 3, 1, 1, 1, 1, True, False, True, 0, 1),         # 64 channels (256 x 64 tiles): res2 conv2
    (3, 37, 45, 256, 64, 1, 1, 0, 1, 0, True, True, True, 16, 1),        # 64 channels, 1x1, residual, norm sums in the epilogue (4 channels per group)
    (2, 33, 47, 128, 64, 3, 1, 2, 2, 1, True, False, False, 32, 1),      # 64 channels, dilated, 2 channels per norm group (separate sums pass)
    (2, 40, 52, 32, 64, 3, 1, 1, 1, 0, True, False, True, 0, 1),         # 32 input channels: two filter taps per K-tile (stem.conv3)
    (3, 37, 45, 32, 64, 3, 1, 1, 1, 0, True, False, False, 0, 1),        # ... ragged tiles across image boundaries
    (2, 40, 52, 32, 32, 3, 1, 1, 1, 0, True, False, True, 0, 1),         # 32 > 32 channels (stem.conv2)
    # the patch kernel (3x3, stride 1, pad 1, <= 128 output channels): 8 x 32-pixel tiles, the 10 x 34 patch of a 64-channel block fetched once
    (3, 37, 45, 128, 128, 3, 1, 1, 1, 1, True, True, True, 0, 1),        # ragged tiles in both directions, residual, two channel blocks
    (2, 20, 24, 320, 128, 3, 1, 1, 1, 1, True, False, False, 32, 1),     # images narrower than a tile, five channel blocks, norm sums in the epilogue
    (1, 64, 96, 64, 128, 3, 1, 1, 1, 1, True, False, True, 0, 1),        # one channel block per tile: every patch is the NEXT tile's
    (2, 33, 70, 128, 64, 3, 1, 1, 1, 1, True, True, False, 16, 1),       # 64 output channels, residual + norm sums
    (2, 40, 52, 128, 32, 3, 1, 1, 1, 1, True, False, True, 0, 1),        # 32 output channels on half-empty 64-channel tiles (the heads' 128 > 32)
    # ... and its 256-channel form (conv_h8w_kernel): four phases, channel tiles of 256; key 38 = 2: the same layers on the DMA-gather kernel
    (2, 40, 52, 64, 256, 3, 1, 1, 1, 1, True, False, True, 0, 1, 2),
    (1, 36, 44, 64, 320, 3, 1, 1, 1, 1, True, False, False, 0, 1),       # ragged channel tile
    (3, 37, 45, 128, 512, 3, 1, 1, 1, 1, True, True, True, 0, 1),        # two channel tiles, residual
    (2, 33, 70, 256, 256, 3, 1, 1, 1, 1, True, False, False, 32, 1),     # norm sums in the epilogue
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(str(v) for v in c[:10]))
def test_h8_conv_matches_float32_convolution_and_the_128_tile_kernel(case):
    B, H, W, cin, cout, k, stride, pad, dil, kmode, affine, residual, relu, groups, key31 = case[:15]
    key38 = case[15] if len(case) > 15 else 1
    lib = _lib.load()
    g = torch.Generator().manual_seed(1234 + cin + cout + k)
    x = torch.randn((B, H, W, cin), generator=g).half()
    w = (torch.randn((cout, cin, k, k), generator=g) / (cin * k * k) ** 0.5).half()
    scale = (0.5 + torch.rand(cout, generator=g)) if affine else None
    shift = torch.randn(cout, generator=g) * 0.3 if affine else None
    oh = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
    ow = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    res = torch.randn((B, oh, ow, cout), generator=g).half() if residual else None
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), None, stride, pad, dil).permute(0, 2, 3, 1)
    if affine:
        ref = ref * scale + shift
    if residual:
        ref = ref + res.float()
    if relu:
        ref = ref.clamp_min(0)
    dev = lambda t_: t_.cuda() if t_ is not None else None
    args = (dev(x), dev(pack(w, kmode)), cout, k, stride, pad, dil, kmode, dev(scale), dev(shift), dev(res), relu, groups)
    try:
        lib.quber_set_tuning(32, 1)            # these launches are a handful of tiles
        lib.quber_set_tuning(31, key31)
        lib.quber_set_tuning(38, key38)
        y, sums = run(lib, *args)
        lib.quber_set_tuning(31, 0)
        y0, sums0 = run(lib, *args)
    finally:
        lib.quber_set_tuning(31, 1)
        lib.quber_set_tuning(38, 1)
        lib.quber_set_tuning(32, 224)
    assert torch.isfinite(y).all()
    tol = 3e-3 * max(1.0, float(ref.abs().max()))
    assert float((y.cpu().float() - ref).abs().max()) < tol
    assert float((y0.cpu().float() - ref).abs().max()) < tol
    # the two kernels differ by the order of the fp32 sum inside a K-tile only: a few fp16 roundings apart at most
    assert float((y.float() - y0.float()).abs().max()) < tol
    if groups:
        yd = y.double().reshape(B, oh * ow, groups, cout // groups)
        exp = torch.stack([yd.sum((1, 3)), (yd * yd).sum((1, 3))], -1)
        assert torch.allclose(sums, exp, rtol=2e-6, atol=1e-4)      # fp32 inside a lane (16 values), fp64 across lanes, tiles and blocks
        yd0 = y0.double().reshape(B, oh * ow, groups, cout // groups)
        assert torch.allclose(sums0, torch.stack([yd0.sum((1, 3)), (yd0 * yd0).sum((1, 3))], -1), rtol=1e-11, atol=1e-9)


@pytest.mark.parametrize("h,w,b,grouped", [(256, 320, 4, False), (1024, 1024, 1, True)], ids=["256x320x4", "1024x1024-grouped-aspp"])
def test_h8_takes_the_wide_layers_of_the_network(h, w, b, grouped):
    """The fp16 network with the kernel on and off: same plan, logits within the fp16 path's own noise of each other.  At 1024x1024
    the three dilated ASPP branches (model.py:610-651, ASPP_DILATIONS 6 / 12 / 18) are ONE grouped launch with per-group dilation;
    with the kernel off the same plan entry runs them as three launches of the 128-tile kernel."""
    from quber_amd import arch, engine, synth
    from oracle import encode_np
    qc = engine.make_config(h, w, max_batch=b)
    qc.compute_dtype = 2
    e = engine.Engine(qc, "cuda:0")
    e.load_state_dict(arch.init_state_dict(seed=11, loud_heads=True))
    batch = synth.make_batch(5, b, h, w, 6)
    offs = np.stack([encode_np.encode_initial_masks(m) for m in batch["masks"]])
    bgr, dep, off = torch.from_numpy(batch["rgb"]).cuda(), torch.from_numpy(batch["depth"]).cuda(), torch.from_numpy(offs).cuda()
    outs, stages = {}, {}
    for mode in (0, 1, 2):                  # 2: the DMA-gather kernels only (key 38 = 0)
        e.set_option(31, min(mode, 1))
        e.set_option(38, 0 if mode == 2 else 1)
        e.set_option(32, 16)
        e.profile_begin()
        outs[mode] = e.forward(bgr, dep, off).clone()
        stages[mode] = e.profile_end()
    names = [p_[0] for p_ in e.plan()]
    e.close()
    # the stem's 32-channel layers and res2 conv2 (64 channels) run on the patch kernels only
    assert stages[1]["conv_gemm_h8"]["launches"] >= stages[2]["conv_gemm_h8"]["launches"] + 4
    assert float((outs[1] - outs[2]).abs().max()) < 2e-2 * max(1.0, float(outs[2].abs().max()))
    assert torch.isfinite(outs[1]).all()
    # the kernel is on the path: the stage profile shows its launches (fusion convolutions, res4 / res5 bottlenecks), none with key 31 = 0
    assert "conv_gemm_h8" not in stages[0] and stages[1]["conv_gemm_h8"]["launches"] >= 10
    assert sum("project_conv.convs." in n and n.endswith(("convs.1", "convs.2", "convs.3")) for n in names) == (1 if grouped else 3)
    # (same ops; a grouped ASPP entry is 3 launches of the 128-tile kernel, and a conv3 + shortcut pair that its persistent dual launch
    #  does not cover at this size is 2)
    n0, n1 = stages[0]["conv_gemm"]["launches"], stages[1]["conv_gemm"]["launches"] + stages[1]["conv_gemm_h8"]["launches"]
    assert n0 - (2 if grouped else 0) - 4 <= n1 <= n0 - (2 if grouped else 0)
    d = (outs[0] - outs[1]).abs()
    scale = max(1.0, float(outs[0].abs().max()))
    assert float(d.max()) < 2e-2 * scale, float(d.max())


@pytest.mark.parametrize("h,w,b", [(256, 320, 3), (200, 264, 2)], ids=["256x320x3", "ragged-200x264x2"])
def test_patch_kernels_absorb_the_norm_in_front_of_them(h, w, b):
    """Option key 39: a patch-kernel layer that is the only reader of a GroupNorm + ReLU output (decoder fuse_conv.1, the heads' head.1:
    model.py:386-403, 610-651) normalises its LDS patches from the producer's raw output instead of reading the result of a norm pass.  Same
    arithmetic (half(max(fmaf(x, scale, bias), 0)) per (image, channel), the zero padding applied after it): the logits are the SAME BITS,
    with fewer norm passes; and with key 38 = 0 at launch time the same plan falls back to the pass + the other kernels."""
    from quber_amd import arch, engine, synth
    from oracle import encode_np
    sd = arch.init_state_dict(seed=13, loud_heads=True)
    batch = synth.make_batch(9, b, h, w, 6)
    offs = np.stack([encode_np.encode_initial_masks(m) for m in batch["masks"]])
    bgr, dep, off = torch.from_numpy(batch["rgb"]).cuda(), torch.from_numpy(batch["depth"]).cuda(), torch.from_numpy(offs).cuda()
    outs, stages = {}, {}
    for key39 in (0, 1):
        qc = engine.make_config(h, w, max_batch=b)
        qc.compute_dtype = 2
        e = engine.Engine(qc, "cuda:0")
        e.set_option(39, key39)
        e.load_state_dict(sd)
        e.set_option(32, 1)
        e.profile_begin()
        outs[key39] = e.forward(bgr, dep, off).clone()
        stages[key39] = e.profile_end()
        if key39:
            e.set_option(38, 0)                 # launch-time: no patch kernels -> the absorbed norms run as passes again
            e.profile_begin()
            outs[2] = e.forward(bgr, dep, off).clone()
            stages[2] = e.profile_end()
        e.close()
    assert torch.isfinite(outs[1]).all()
    assert stages[1]["gn_apply"]["launches"] <= stages[0]["gn_apply"]["launches"] - 4, (stages[0]["gn_apply"], stages[1]["gn_apply"])
    assert stages[2]["gn_apply"]["launches"] == stages[0]["gn_apply"]["launches"]
    assert torch.equal(outs[0], outs[1])
    assert float((outs[2] - outs[0]).abs().max()) < 2e-2 * max(1.0, float(outs[0].abs().max()))


def test_h8_random_geometries_are_deterministic_and_match():
    """Randomised launches (ragged pixel and channel tiles, dilation, stride, every epilogue variant, few and many tiles per block so
    that the DMA pipeline crosses tile boundaries): every launch three times - the LDS-DMA pipeline has no data-dependent control, so
    ANY run-to-run difference of the output tensor would be a race between a DMA and a fragment read - and against the 128-tile kernel."""
    lib = _lib.load()
    rng = np.random.default_rng(20251)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t_: C.c_void_p(t_.data_ptr()) if t_ is not None else None
    try:
        lib.quber_set_tuning(32, 1)
        for it in range(int(os.environ.get("QUBER_H8_FUZZ", "48"))):      # (profiles/r11_h8_fuzz.txt: 600 launches)
            k = int(rng.choice([1, 3]))
            cin = int(rng.choice([64, 128, 192, 256, 384])) if k == 3 else int(rng.choice([192, 256, 512, 1024]))
            cout = int(rng.choice([128, 128, 256, 256, 320, 512]))
            B, H, W = int(rng.integers(1, 4)), int(rng.integers(17, 90)), int(rng.integers(17, 90))
            if rng.random() < 0.3:
                H, W = int(rng.integers(90, 200)), int(rng.integers(90, 200))      # several tiles per block
            dil = int(rng.integers(1, 4)) if k == 3 else 1
            stride = int(rng.choice([1, 2])) if k == 1 else 1
            pad = dil if k == 3 else 0
            residual = bool(rng.random() < 0.3)
            groups = 32 if (not residual and rng.random() < 0.5 and ((H - 1) // stride + 1) * ((W - 1) // stride + 1) >= 256) else 0
            relu = bool(rng.random() < 0.5)
            g = torch.Generator().manual_seed(1000 + it)
            x = torch.randn((B, H, W, cin), generator=g).half().cuda()
            wp = pack((torch.randn((cout, cin, k, k), generator=g) / (cin * k * k) ** 0.5).half(), 1 if k == 3 else 0).cuda()
            scale, shift = (0.5 + torch.rand(cout, generator=g)).cuda(), (torch.randn(cout, generator=g) * 0.3).cuda()
            oh, ow = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1, (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
            res = torch.randn((B, oh, ow, cout), generator=g).half().cuda() if residual else None
            args = (x, wp, cout, k, stride, pad, dil, 1 if k == 3 else 0, scale, shift, res, relu, groups)
            lib.quber_set_tuning(31, 1)
            lib.quber_set_tuning(38, 1 + it % 2)          # (odd cases: the wide undilated 3x3 layers on the DMA-gather kernel)
            ys = [run(lib, *args) for _ in range(3)]
            lib.quber_set_tuning(31, 0)
            y0, _ = run(lib, *args)
            what = f"case {it}: k{k} {cin}>{cout} B{B} {H}x{W} d{dil} s{stride} res{int(residual)} gn{groups} relu{int(relu)}"
            assert torch.equal(ys[0][0], ys[1][0]) and torch.equal(ys[0][0], ys[2][0]), what
            d = float((ys[0][0].float() - y0.float()).abs().max())
            assert d < 3e-3 * max(1.0, float(y0.float().abs().max())), (what, d)
            if groups:
                yd = ys[0][0].double().reshape(B, oh * ow, groups, cout // groups)
                exp = torch.stack([yd.sum((1, 3)), (yd * yd).sum((1, 3))], -1)
                assert torch.allclose(ys[0][1], exp, rtol=2e-6, atol=1e-4), what
    finally:
        lib.quber_set_tuning(31, 1)
        lib.quber_set_tuning(38, 1)
        lib.quber_set_tuning(32, 224)
