#!/usr/bin/env python3
"""Phase timing inside the single-kernel Winograd layer (diagnostic build in a scratch copy: QUBER_LIB=$(tools/diag_build.sh wfstamps WFX=-DWF_STAMPS); wino_fused.hip
WF_STAMP): per wave of the first 512 blocks, shader-clock stamps at entry, after the prologue, after the first / second round,
at the end of the K loop, after the accumulators are in LDS, after the output transform + stores, at exit; plus the chip-wide
100 MHz clock at entry / exit and the hardware id.   GPU box only.
usage: wf_stamps.py H W Cin Cout [frames=16] [groups=1] [norm=0]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quber_amd import _lib  # noqa: E402

H, W, Cin, Cout = (int(v) for v in sys.argv[1:5])
F = int(sys.argv[5]) if len(sys.argv) > 5 else 16
lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)
lib.quber_set_tuning(2, 1)
for kv in os.environ.get("QUBER_TUNE", "").split(","):
    if "=" in kv:
        lib.quber_set_tuning(*(int(v) for v in kv.split("=")))
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
B = F
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.randn(B, H, W, Cin, device="cuda", generator=g)
w = torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) / np.sqrt(Cin * 9)
sc, sh = torch.rand(Cout, device="cuda", generator=g) + 0.5, torch.randn(Cout, device="cuda", generator=g)
tiles = B * ((H + 3) // 4) * ((W + 3) // 4)
u = torch.empty(36 * Cout * Cin, device="cuda")
ws = torch.empty(max(36 * tiles * (Cin + Cout), 36 * Cout * Cin + 2 * B * Cin), device="cuda")
y = torch.empty(B, H, W, Cout, device="cuda")
run = lambda: _lib.check(lib.quber_op_conv3x3_winograd(p(x), B, H, W, Cin, p(w), Cout, 1, 4, p(sc), p(sh), 1, p(u), p(ws), ws.numel(), p(y), st))
run()
lib.quber_set_tuning(26, 1)
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
wide = Cout % 64 == 0 and (Cin // 32) % 2 == 0
ft, fc = (16, 64) if wide else (32, 32)
nblocks = -(-tiles // ft) * (Cout // fc)
print(f"layer {Cin}>{Cout} @{H}x{W} x{B}: {us:.1f} us, {nblocks} blocks of {ft} tiles x {fc} channels ({nblocks / 256:.2f} per CU), "
      f"{2 * 36 * tiles * Cin * Cout / us / 1e6:.1f} TFLOP/s executed")
NB, NS = 512, 16
buf = (C.c_ulonglong * (NB * 4 * NS))()
assert raw.quber_wf_read_stamps(buf, NB * 4 * NS) == 0
s = np.array(buf[:], dtype=np.int64).reshape(NB, 4, NS)[:min(NB, nblocks)].astype(np.float64)
names = [("prologue (patches of rounds 0, 1: load, transform, first image)", 0, 1), ("round 0", 1, 2), ("round 1", 2, 3),
         ("rounds 2 .. R-1", 3, 6), ("accumulators -> LDS + barrier", 6, 7), ("output transform + stores", 7, 8), ("GroupNorm sums / exit", 8, 9),
         ("whole block", 0, 9)]
R = Cin // (32 if wide else 16)
mf = 18 * (16 * 32 if wide else 4 * 64)             # MFMA cycles of a round
print(f"  R = {R} rounds of {mf} MFMA cycles; all figures: shader-clock cycles, median over {s.shape[0]} blocks x 4 waves [p10 .. p90]")
for nm, a, b in names:
    d = (s[:, :, b] - s[:, :, a]).ravel()
    d = d[(s[:, :, b].ravel() > 0) & (s[:, :, a].ravel() > 0)]
    if len(d):
        print(f"  {nm:70s} {np.median(d):9.0f}  [{np.percentile(d, 10):.0f} .. {np.percentile(d, 90):.0f}]")
tot = np.median((s[:, :, 9] - s[:, :, 0]).ravel())
print(f"  MFMA cycles per block {R * mf} = {R * mf / tot:.2f} of the block's time")
rt0, rt1 = s[:, 0, 15], s[:, 0, 14]
t0 = rt0.min()
print(f"  100 MHz clock: block entry p0/p50/p100 = {np.percentile(rt0 - t0, [0, 50, 100]) / 100.0} us, exit = {np.percentile(rt1 - t0, [0, 50, 100]) / 100.0} us, "
      f"duration p50 {np.median(rt1 - rt0) / 100.0:.1f} us -> effective clock {tot / (np.median(rt1 - rt0) / 100.0) / 1e3:.2f} GHz")
