#!/usr/bin/env python3
"""Sweeps (tile shape, K partitions) for every distinct convolution shape of the refiner on the stand-alone conv op and
prints, per shape, the automatic choice next to the best forced one.  usage: conv_sweep.py [frames=16] [json out | -] [frame height=480] [frame width=640]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quber_amd import _lib  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
FH, FW = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (480, 640)      # the table below is written for 640 x 480 frames
# name, images per frame (streams / grouped heads), H_in, W_in, Cin, Cout, k, stride, dil, residual
SHAPES = [
    ("stem.conv1 3x3s2 8>32", 2, 480, 640, 8, 32, 3, 2, 1, 0),
    ("stem.conv2 3x3 32>32", 2, 240, 320, 32, 32, 3, 1, 1, 0),
    ("stem.conv3 3x3 32>64", 2, 240, 320, 32, 64, 3, 1, 1, 0),
    ("res2.conv1 1x1 64>64", 2, 120, 160, 64, 64, 1, 1, 1, 0),
    ("res2.conv1 1x1 256>64", 2, 120, 160, 256, 64, 1, 1, 1, 0),
    ("res2.conv2 3x3 64>64", 2, 120, 160, 64, 64, 3, 1, 1, 0),
    ("res2.conv3 1x1 64>256 +res", 2, 120, 160, 64, 256, 1, 1, 1, 1),
    ("res2.shortcut 1x1 64>256", 2, 120, 160, 64, 256, 1, 1, 1, 0),
    ("res3.conv1 1x1s2 256>128", 2, 120, 160, 256, 128, 1, 2, 1, 0),
    ("res3.conv1 1x1 512>128", 2, 60, 80, 512, 128, 1, 1, 1, 0),
    ("res3.conv2 3x3 128>128", 2, 60, 80, 128, 128, 3, 1, 1, 0),
    ("res3.conv3 1x1 128>512 +res", 2, 60, 80, 128, 512, 1, 1, 1, 1),
    ("res3.shortcut 1x1s2 256>512", 2, 120, 160, 256, 512, 1, 2, 1, 0),
    ("res4.conv1 1x1s2 512>256", 2, 60, 80, 512, 256, 1, 2, 1, 0),
    ("res4.conv1 1x1 1024>256", 2, 30, 40, 1024, 256, 1, 1, 1, 0),
    ("res4.conv2 3x3 256>256", 2, 30, 40, 256, 256, 3, 1, 1, 0),
    ("res4.conv3 1x1 256>1024 +res", 2, 30, 40, 256, 1024, 1, 1, 1, 1),
    ("res4.shortcut 1x1s2 512>1024", 2, 60, 80, 512, 1024, 1, 2, 1, 0),
    ("res5.conv1 1x1 1024>512", 2, 30, 40, 1024, 512, 1, 1, 1, 0),
    ("res5.conv1 1x1 2048>512", 2, 30, 40, 2048, 512, 1, 1, 1, 0),
    ("res5.conv2 3x3d4 512>512", 2, 30, 40, 512, 512, 3, 1, 4, 0),
    ("res5.conv3 1x1 512>2048 +res", 2, 30, 40, 512, 2048, 1, 1, 1, 1),
    ("res5.shortcut 1x1 1024>2048", 2, 30, 40, 1024, 2048, 1, 1, 1, 0),
    ("fusion_res2 1x1 512>256", 1, 120, 160, 512, 256, 1, 1, 1, 0),
    ("fusion_res2 3x3 256>256", 1, 120, 160, 256, 256, 3, 1, 1, 0),
    ("fusion_res3 1x1 1024>512", 1, 60, 80, 1024, 512, 1, 1, 1, 0),
    ("fusion_res3 3x3 512>512", 1, 60, 80, 512, 512, 3, 1, 1, 0),
    ("fusion_res5 1x1 4096>2048", 1, 30, 40, 4096, 2048, 1, 1, 1, 0),
    ("aspp 1x1 2048>256", 1, 30, 40, 2048, 256, 1, 1, 1, 0),
    ("aspp 3x3d12 2048>256", 1, 30, 40, 2048, 256, 3, 1, 12, 0),
    ("aspp.project 1x1 1280>256", 1, 30, 40, 1280, 256, 1, 1, 1, 0),
    ("dec.res3.project 1x1 512>64", 1, 60, 80, 512, 64, 1, 1, 1, 0),
    ("dec.res3.fuse0 3x3 320>128", 1, 60, 80, 320, 128, 3, 1, 1, 0),
    ("dec.res3.fuse1 3x3 128>128", 1, 60, 80, 128, 128, 3, 1, 1, 0),
    ("dec.res2.project 1x1 256>32", 1, 120, 160, 256, 32, 1, 1, 1, 0),
    ("dec.res2.fuse0 3x3 160>128", 1, 120, 160, 160, 128, 3, 1, 1, 0),
    ("head 3x3 128>128", 1, 120, 160, 128, 128, 3, 1, 1, 0),
    ("head 3x3 128>32", 1, 120, 160, 128, 32, 3, 1, 1, 0),
    ("head.fusion 1x1 164>128", 1, 120, 160, 164, 128, 1, 1, 1, 0),
    ("heads x3 3x3 128>128", 3, 120, 160, 128, 128, 3, 1, 1, 0),
    ("heads x3 3x3 128>32", 3, 120, 160, 128, 32, 3, 1, 1, 0),
]
ALL = []
TILES = {1: (64, 64, 7), 2: (128, 128, 3), 4: (256, 32, 3)}    # quber_set_tuning key 4 values -> (BM, BN, blocks per CU)


def main():
    lib = _lib.load()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    lib.quber_set_tuning(2, 1)
    print(f"| shape ({F} frames) | GFLOP | auto ms | auto TF/s | best (tile, S) | best ms | best TF/s | gain | runners-up |")
    print("|---|---|---|---|---|---|---|---|---|")
    tot_auto = tot_best = 0.0
    for (name, ipf, H, W, Cin, Cout, k, s, d, res) in SHAPES:
        H, W = H * FH // 480, W * FW // 640
        B = ipf * F
        pad = d * (k // 2)
        OH, OW = (H + 2 * pad - d * (k - 1) - 1) // s + 1, (W + 2 * pad - d * (k - 1) - 1) // s + 1
        x = torch.randn(B, H, W, Cin, device="cuda")
        w = torch.randn(Cout, Cin, k, k, device="cuda") / np.sqrt(Cin * k * k)
        sc, sh = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda")
        y = torch.empty(B, OH, OW, Cout, device="cuda")
        r = torch.randn(B, OH, OW, Cout, device="cuda") if res else None
        Kpad = (k * k * Cin + 31) // 32 * 32
        packed = torch.empty(Cout * Kpad, device="cuda")
        flops = 2.0 * B * OH * OW * Cin * k * k * Cout
        M, nk = B * OH * OW, Kpad // 32

        def timed(tile, S, rounds=4):
            lib.quber_set_tuning(4, tile)
            lib.quber_set_tuning(3, S)
            ts = []
            for rd in range(rounds + 1):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    _lib.check(lib.quber_op_conv2d(p(x), B, H, W, Cin, p(w), Cout, k, s, pad, d, p(sc), p(sh), p(r), 1,
                                                   p(packed), p(y), st))
                e1.record()
                torch.cuda.synchronize()
                if rd:
                    ts.append(e0.elapsed_time(e1) / 3)
            return float(np.median(ts))

        auto = timed(0, 0)                  # the launcher's own choice (workspace available)
        ref = y.clone()
        results = []
        for tile, (bm, bn, bpc) in TILES.items():
            if bn < 64 and Cout > 64:
                continue
            blocks = ((M + bm - 1) // bm) * ((Cout + bn - 1) // bn)
            for S in (1, 2, 3, 4, 6, 8, 12):
                if S > 1 and (nk // S < 4 or blocks * S > 256 * bpc * 6 or S * M * Cout > (256 << 20)):
                    continue
                t = timed(tile, S)
                assert torch.allclose(y, ref, rtol=1e-4, atol=1e-4), (name, tile, S)
                results.append((t, tile, S))
        results.sort()
        ALL.append({"name": name, "frames": F, "M": M, "Cout": Cout, "nk": nk, "res": res, "auto": auto,
                    "runs": [(TILES[t_][0], TILES[t_][1], S_, tt) for tt, t_, S_ in results]})
        bt, btile, bS = results[0]
        tot_auto += auto
        tot_best += min(bt, auto)
        ru = ", ".join("(%dx%d,%d) %.3f" % (TILES[t_][0], TILES[t_][1], S_, tt) for tt, t_, S_ in results[1:4])
        print("| %s | %.1f | %.3f | %.1f | (%dx%d, %d) | %.3f | %.1f | %+.0f %% | %s |" % (
            name, flops / 1e9, auto, flops / auto / 1e9, TILES[btile][0], TILES[btile][1], bS, bt, flops / bt / 1e9,
            (auto / bt - 1) * 100, ru), flush=True)
    print("| sum over distinct shapes | | %.3f | | | %.3f | | | |" % (tot_auto, tot_best))
    if len(sys.argv) > 2 and sys.argv[2] != "-":
        import json
        json.dump(ALL, open(sys.argv[2], "w"))
    lib.quber_set_tuning(3, 0)
    lib.quber_set_tuning(4, 0)
    lib.quber_set_tuning(2, 0)


if __name__ == "__main__":
    main()
