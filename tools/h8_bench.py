"""Times the fp16 data path's convolution kernels on the wide layer shapes of BASELINE configs[4] (1024 x 1024, batch 8):
csrc/conv_h8.hip (key 31 = 1) against csrc/conv_igemm.hip (key 31 = 0), through quber_op_conv2d_f16.  One group per launch
(the network runs the two encoder streams as one grouped launch: twice the tiles).
    python tools/h8_bench.py [--iters 20] [--batch 8] [--size 1024]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quber_amd import _lib  # noqa: E402


def pack(w, kmode):
    co, ci, kh, kw = w.shape
    t = w.permute(0, 2, 3, 1).reshape(co, kh * kw, ci)
    if kmode == 0:
        k = kh * kw * ci
        return torch.nn.functional.pad(t.reshape(co, k), (0, (k + 63) // 64 * 64 - k)).contiguous()
    return t.reshape(co, kh * kw, ci // 64, 64).permute(0, 2, 1, 3).reshape(co, -1).contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--only", type=str, default="")
    a = ap.parse_args()
    lib = _lib.load()
    B, S = a.batch, a.size
    s4, s8, s16 = S // 4, S // 8, S // 16
    # name, H, cin, cout, k, dil, residual, gn
    layers = [
        ("fusion_res2.conv0 3x3 256>256", s4, 256, 256, 3, 1, False, True),
        ("fusion_res3.conv0 3x3 512>512", s8, 512, 512, 3, 1, False, True),
        ("res5.conv2 3x3 d2 512>512", s16, 512, 512, 3, 2, False, False),
        ("res4.conv2 3x3 256>256", s16, 256, 256, 3, 1, False, False),
        ("fusion_res5.conv 1x1 4096>2048", s16, 4096, 2048, 1, 1, False, True),
        ("fusion_res3.conv 1x1 1024>512", s8, 1024, 512, 1, 1, False, True),
        ("fusion_res2.conv 1x1 512>256", s4, 512, 256, 1, 1, False, True),
        ("res5.conv1 1x1 2048>512", s16, 2048, 512, 1, 1, False, False),
        ("res5.conv3 1x1 512>2048 +res", s16, 512, 2048, 1, 1, True, False),
        ("res4.conv3 1x1 256>1024 +res", s16, 256, 1024, 1, 1, True, False),
        ("head.0 / fusion_layers 3x3 128>128", s4, 128, 128, 3, 1, False, True),
        ("res3.conv2 3x3 128>128", s8, 128, 128, 3, 1, False, False),
        ("fusion_layers.1 3x3 128>128 (no norm sums)", s4, 128, 128, 3, 1, False, False),
        ("decoder.res3.fuse_conv.0 3x3 320>128", s8, 320, 128, 3, 1, False, True),
        ("stem.conv3 3x3 32>64", S // 2, 32, 64, 3, 1, False, False),
        ("res2.conv2 3x3 64>64", s4, 64, 64, 3, 1, False, False),
        ("res2.conv1 1x1 256>64", s4, 256, 64, 1, 1, False, False),
    ]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    print(f"| layer (batch {B}, {S}x{S} frame) | GFLOP | conv_igemm ms | TFLOP/s | conv_h8 ms | TFLOP/s | ratio |")
    print("|---|---|---|---|---|---|---|")
    for name, H, cin, cout, k, dil, residual, gn in layers:
        if a.only and a.only not in name:
            continue
        g = torch.Generator().manual_seed(1)
        x = torch.randn((B, H, H, cin), generator=g).half().cuda()
        kmode = 1 if k == 3 and cin % 64 == 0 else 0
        w = pack((torch.randn((cout, cin, k, k), generator=g) / (cin * k * k) ** 0.5).half(), kmode).cuda()
        scale, shift = torch.ones(cout).cuda(), torch.zeros(cout).cuda()
        res = torch.randn((B, H, H, cout), generator=g).half().cuda() if residual else None
        y = torch.empty((B, H, H, cout), dtype=torch.float16, device="cuda")
        sums = torch.zeros((B, 32, 2), dtype=torch.float64, device="cuda") if gn else None
        p = lambda t_: C.c_void_p(t_.data_ptr()) if t_ is not None else None
        pad = dil if k == 3 else 0
        ms = {}
        for mode in (0, 1):
            lib.quber_set_tuning(31, mode)
            call = lambda: _lib.check(lib.quber_op_conv2d_f16(p(x), B, H, H, cin, p(w), cout, k, 1, pad, dil, kmode, p(scale), p(shift), p(res), 1,
                                                              p(sums), 32 if gn else 0, p(y), st))
            for _ in range(3):
                call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                call()
            e1.record()
            torch.cuda.synchronize()
            ms[mode] = e0.elapsed_time(e1) / a.iters
        lib.quber_set_tuning(31, 1)
        gf = 2.0 * B * H * H * cin * k * k * cout / 1e9
        print(f"| {name} | {gf:.1f} | {ms[0]:.3f} | {gf / ms[0]:.0f} | {ms[1]:.3f} | {gf / ms[1]:.0f} | {ms[0] / ms[1]:.2f} |", flush=True)


if __name__ == "__main__":
    main()
