#!/usr/bin/env python3
"""Phase timing inside the persistent convolution kernel (diagnostic build in a scratch copy: QUBER_LIB=$(tools/diag_build.sh pk STAMPS=1)).
For the first tiles of the first 48 blocks: K loop, staging of the next tile's first K-slice, epilogue, restart.
GPU box only.  usage: pk_stamps.py [K] [N] [B]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quber_amd import _lib  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
B = int(sys.argv[3]) if len(sys.argv) > 3 else 36
H, W = 120, 160
lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
lib.quber_set_tuning(2, 1)
lib.quber_set_tuning(15, 0)
lib.quber_set_tuning(13, 1)
lib.quber_set_tuning(4, 2)
for kv in (sys.argv[4].split(",") if len(sys.argv) > 4 else []):
    lib.quber_set_tuning(int(kv.split("=")[0]), int(kv.split("=")[1]))
x = torch.randn(B, H, W, K, device="cuda")
w = torch.randn(N, K, 1, 1, device="cuda") / np.sqrt(K)
y = torch.empty(B, H, W, N, device="cuda")
packed = torch.empty(N * K, device="cuda")
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(lib.quber_op_conv2d(p(x), B, H, W, K, p(w), N, 1, 1, 0, 1, p(None), p(None), p(None), 0, p(packed), p(y), st))
    e1.record()
    torch.cuda.synchronize()
print(f"last launch (weight packing + convolution), HIP events: {e0.elapsed_time(e1) * 1e3:.1f} us")
span = (C.c_ulonglong * 8192)()
assert raw.quber_pk_read_span(span, 8192) == 0
sp = np.array(span[:], dtype=np.int64).reshape(2048, 4)[:768].astype(np.float64)
hw = np.array(span[:], dtype=np.uint64).reshape(2048, 4)[:768, 3]
t0 = sp[:, 0].min()
pct = lambda v: "min %.1f  p50 %.1f  max %.1f" % tuple(np.percentile((v - t0) / 100.0, [0, 50, 100]))
cu = ((hw >> np.uint64(8)) & np.uint64(0xFF)).astype(np.int64) + 256 * ((hw >> np.uint64(32)) & np.uint64(7)).astype(np.int64)
per_cu = np.bincount(cu)
print(f"  768 blocks, us from the first entry: entry {pct(sp[:, 0])} | first K loop {pct(sp[:, 1])} | exit {pct(sp[:, 2])}")
print(f"  blocks per CU: {sorted(set(per_cu[per_cu > 0].tolist()))} on {int((per_cu > 0).sum())} CUs")
xcc = ((hw >> np.uint64(32)) & np.uint64(7)).astype(np.int64)
dur = (sp[:, 2] - sp[:, 1]) / 100.0           # first K loop -> exit, us
for x in range(8):
    d = dur[xcc == x]
    print(f"    XCC {x} (blockIdx % 8 in {sorted(set((np.arange(768)[xcc == x] % 8).tolist()))}): work time min {d.min():.0f} p50 {np.median(d):.0f} max {d.max():.0f} us")
# spread inside a CU against spread between CUs
cu_mean = np.array([dur[cu == c].mean() for c in np.unique(cu)])
cu_rng = np.array([dur[cu == c].max() - dur[cu == c].min() for c in np.unique(cu)])
print(f"    per-CU mean work time: min {cu_mean.min():.0f} p50 {np.median(cu_mean):.0f} max {cu_mean.max():.0f} us; range inside a CU: p50 {np.median(cu_rng):.0f} max {cu_rng.max():.0f} us")
se = ((hw >> np.uint64(13)) & np.uint64(7)).astype(np.int64)
for x in range(2):
    for e in sorted(set(se[xcc == x].tolist())):
        d = dur[(xcc == x) & (se == e)]
        print(f"    XCC {x} SE {e}: {len(d)} blocks, work time p50 {np.median(d):.0f} us")
NB, NT, NS = 48, 24, 12
buf = (C.c_ulonglong * (NB * NT * NS))()
assert raw.quber_pk_read_stamps(buf, NB * NT * NS) == 0
s = np.array(buf[:], dtype=np.int64).reshape(NB, NT, NS).astype(np.float64)
nk = K // 32
# stamps: 0 = first slice in LDS (K loop starts), 4 = second slice in LDS, 1 = last slice multiplied (barrier passed),
# 2 = next tile's first slice stored to LDS, 3 = epilogue stores issued; next tile's 0 = restart
tiles = ((B * H * W + 127) // 128) * ((N + 127) // 128)
per_block = min(NT, tiles // 768)          # whole tiles every block computes
t = s[:, 1:per_block - 1]          # skip the first tile (cold) and the last recorded
nxt0 = s[:, 2:per_block, 0]
ghz = 2.35
def us(c): return float(np.median(c)) / ghz / 1e3
print(f"K {K} (nk {nk}) N {N} M {B * H * W}; medians over {t.shape[0]} blocks x {t.shape[1]} tiles, us at {ghz} GHz")
print(f"  first K-slice (0 -> 4)                    {us(t[:, :, 4] - t[:, :, 0]):7.2f}")
print(f"  K loop, all slices (0 -> 1)               {us(t[:, :, 1] - t[:, :, 0]):7.2f}   = {us(t[:, :, 1] - t[:, :, 0]) / nk:.2f} per slice")
print(f"  wait for + store next tile's slice (1->2) {us(t[:, :, 2] - t[:, :, 1]):7.2f}")
print(f"  epilogue (2 -> 3)                         {us(t[:, :, 3] - t[:, :, 2]):7.2f}")
for a, b_, name in ((2, 5, "descriptors, addresses (2 -> 5)"), (5, 6, "issue affine-parameter loads (5 -> 6)"), (6, 7, "tile 0: wait, transposition, 4 stores (6 -> 7)"),
                    (7, 8, "tile 1 (7 -> 8)"), (8, 9, "tile 2 (8 -> 9)"), (9, 10, "tile 3 (9 -> 10)"), (10, 3, "end of the epilogue (10 -> 3)")):
    print(f"      {name:48s} {us(t[:, :, b_] - t[:, :, a]):7.2f}")
print(f"  zero accumulators + barrier (3 -> next 0) {us(nxt0 - t[:, :, 3]):7.2f}")
print(f"  whole tile (0 -> next 0)                  {us(nxt0 - t[:, :, 0]):7.2f}")
# are the blocks of a launch in lockstep?  spread of the K-loop start of tile 5 over the blocks

