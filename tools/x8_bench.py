"""Times the bf16x3 GEMM kernels on the wide 1x1 launches of the headline configuration (batch 16, 640 x 480): csrc/conv_x8.hip
(key 35 = 2) against csrc/conv_igemm.hip + conv_persist.hip DT 3 (key 35 = 0), through quber_op_conv2d in the bf16x3 mode (key 12 = 3)
with the op workspace (key 2), so that the old path takes its persistent / split-K kernels as in the network.
    python tools/x8_bench.py [--iters 10]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quber_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    lib = _lib.load()
    lib.quber_set_tuning(2, 1)
    lib.quber_set_tuning(12, 3)
    B = 16
    layers = [
        ("fusion_res5.conv 4096>2048 @30x40", (B, 30, 40), 4096, 2048, False),
        ("res5.conv3 512>2048 +res @30x40 (x2 streams)", (2 * B, 30, 40), 512, 2048, True),
        ("res5.conv1 2048>512 @30x40 (x2)", (2 * B, 30, 40), 2048, 512, False),
        ("fusion_res2.conv 512>256 @120x160", (B, 120, 160), 512, 256, False),
        ("fusion_res3.conv 1024>512 @60x80", (B, 60, 80), 1024, 512, False),
        ("wino GEMM 256>256, 19200 tiles x 36 positions", (36, 120, 160), 256, 256, False),
        ("wino GEMM 512>512, 4800 tiles x 36 positions", (36, 60, 80), 512, 512, False),
        ("res4.conv1 1024>256 @30x40 (x2)", (2 * B, 30, 40), 1024, 256, False),
        ("res4.conv3 256>1024 +res @30x40 (x2)", (2 * B, 30, 40), 256, 1024, True),
    ]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    print("| launch (one group) | GFLOP | conv_igemm bf16x3 ms | fp32-equivalent TFLOP/s | conv_x8 ms | TFLOP/s | ratio |")
    print("|---|---|---|---|---|---|---|")
    for name, (b, h, w), cin, cout, residual in layers:
        if a.only and a.only not in name:
            continue
        g = torch.Generator().manual_seed(1)
        x = torch.randn((b, h, w, cin), generator=g).cuda()
        wt = (torch.randn((cout, cin, 1, 1), generator=g) / cin ** 0.5).cuda()
        scale, shift = torch.ones(cout).cuda(), torch.zeros(cout).cuda()
        res = torch.randn((b, h, w, cout), generator=g).cuda() if residual else None
        y = torch.empty((b, h, w, cout), device="cuda")
        scratch = torch.empty(cout * cin, device="cuda")
        p = lambda t_: C.c_void_p(t_.data_ptr()) if t_ is not None else None
        ms = {}
        for mode in (0, 2):
            lib.quber_set_tuning(35, mode)
            call = lambda: _lib.check(lib.quber_op_conv2d(p(x), b, h, w, cin, p(wt), cout, 1, 1, 0, 1, p(scale), p(shift), p(res), 1, p(scratch), p(y), st))
            for _ in range(2):
                call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                call()
            e1.record()
            torch.cuda.synchronize()
            ms[mode] = e0.elapsed_time(e1) / a.iters
        gf = 2.0 * b * h * w * cin * cout / 1e9
        print(f"| {name} | {gf:.1f} | {ms[0]:.3f} | {gf / ms[0]:.1f} | {ms[2]:.3f} | {gf / ms[2]:.1f} | {ms[0] / ms[2]:.2f} |", flush=True)
    lib.quber_set_tuning(35, 1)
    lib.quber_set_tuning(12, 0)


if __name__ == "__main__":
    main()
