"""Diagnostic: stand-alone fp16 convolution op (conv_h8.hip, key 31) on network-size launches, run repeatedly: bit-equal between repeats?
Close to the 128-tile kernel (key 31 = 0)?  usage: python3 tools/h8_repeat_probe.py B [B ...]"""
import sys
import torch
sys.path.insert(0, "tests")
sys.path.insert(0, ".")
from test_gpu_h8 import pack, run          # noqa: E402
from quber_amd import _lib                 # noqa: E402

lib = _lib.load()
SHAPES = [   # H, W, cin, cout, k, stride, pad, dil, kmode, residual, groups   (res3 stage of a 640x480 frame)
    ("res3.conv3 128>512 +res", 60, 80, 128, 512, 1, 1, 0, 1, 0, True, 0),
    ("res3.0 shortcut 256>512 s2", 120, 160, 256, 512, 1, 2, 0, 1, 0, False, 0),
    ("fusion_res3.conv 1024>512 gn", 60, 80, 1024, 512, 1, 1, 0, 1, 0, False, 32),
    ("res4.conv3 256>1024 +res", 30, 40, 256, 1024, 1, 1, 0, 1, 0, True, 0),
    ("res3.0.conv1 256>128 s2", 120, 160, 256, 128, 1, 2, 0, 1, 0, False, 0),
]
for B in [int(v) for v in sys.argv[1:]]:
    for name, H, W, cin, cout, k, stride, pad, dil, kmode, residual, groups in SHAPES:
        g = torch.Generator().manual_seed(7)
        x = torch.randn((B, H, W, cin), generator=g).half().cuda()
        w = (torch.randn((cout, cin, k, k), generator=g) / (cin * k * k) ** 0.5).half()
        scale, shift = (0.5 + torch.rand(cout, generator=g)).cuda(), (torch.randn(cout, generator=g) * 0.3).cuda()
        oh, ow = (H - 1) // stride + 1, (W - 1) // stride + 1
        res = torch.randn((B, oh, ow, cout), generator=g).half().cuda() if residual else None
        args = (x, pack(w, kmode).cuda(), cout, k, stride, pad, dil, kmode, scale, shift, res, True, groups)
        lib.quber_set_tuning(31, 0)
        y0, _ = run(lib, *args)
        lib.quber_set_tuning(31, 1)
        ys = [run(lib, *args)[0] for _ in range(6)]
        rep = max(float((y.float() - ys[0].float()).abs().max()) for y in ys[1:])
        d0 = float((ys[0].float() - y0.float()).abs().max())
        bad = (ys[1].float() - ys[0].float()).abs().amax((1, 2, 3)) if rep else None
        print(f"B {B} {name}: M {B * oh * ow}, max diff between repeats {rep:.3e}, vs 128-tile kernel {d0:.3e}" + (f", frames {[i for i in range(B) if float(bad[i]) > 0]}" if rep else ""), flush=True)
