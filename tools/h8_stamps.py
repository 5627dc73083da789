#!/usr/bin/env python3
"""Where a tile's time goes inside csrc/conv_h8.hip (diagnostic build in a scratch copy: QUBER_LIB=$(tools/diag_build.sh h8stamps H8X=-DH8_STAMPS), i.e. another
directory than the product build).  s_memtime of wave 0 of every block at: tile start, K loop start, K loop end, epilogue end,
start of K-tile 4, start of K-tile nk - 4.  usage (GPU box): h8_stamps.py [layer substring of tools/h8_bench.py]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quber_amd import _lib  # noqa: E402

layer = sys.argv[1] if len(sys.argv) > 1 else "fusion_res2.conv0"
lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)
sys.argv = [sys.argv[0], "--only", layer, "--iters", "2"]
import h8_bench  # noqa: E402
h8_bench.main()
N = 256 * 16 * 8
buf = (C.c_ulonglong * N)()
assert raw.quber_h8_read_stamps(buf, N) == 0
s = np.array(buf[:], dtype=np.int64).reshape(256, 16, 8).astype(np.float64)
ok = s[:, :, 3] > 0
print(f"tiles stamped per block: {ok.sum(1).min()} .. {ok.sum(1).max()}")
for ti in range(int(ok.sum(1).max())):
    v = s[:, ti][ok[:, ti]]
    if len(v) == 0:
        break
    pre, loop, epi = v[:, 1] - v[:, 0], v[:, 2] - v[:, 1], v[:, 3] - v[:, 2]
    mid = v[:, 5] - v[:, 4]
    line = f"tile {ti}: setup+stagger p50 {np.median(pre):7.0f}  K loop p50 {np.median(loop):8.0f}  epilogue p50 {np.median(epi):7.0f} max {epi.max():7.0f}  steady K-tiles [4, nk-4) p50 {np.median(mid):8.0f} cycles"
    if v[:, 6].any():     # the 256 x 128 / 256 x 64 kernels stamp the starts of K-tiles 1, 2, 3 instead
        line = (f"tile {ti}: setup+stagger p50 {np.median(pre):7.0f}  K loop p50 {np.median(loop):8.0f}  epilogue p50 {np.median(epi):7.0f} max {epi.max():7.0f}  "
                f"K-tile 0 p50 {np.median(v[:, 4] - v[:, 1]):6.0f}  K-tile 1 {np.median(v[:, 5] - v[:, 4]):6.0f}  K-tile 2 {np.median(v[:, 6] - v[:, 5]):6.0f}")
    if ti + 1 < 16 and ok[:, ti + 1].any():
        nxt = s[:, ti + 1][ok[:, ti + 1]]
        both = ok[:, ti] & ok[:, ti + 1]
        line += f"  tile period p50 {np.median(s[both, ti + 1, 0] - s[both, ti, 0]):8.0f}"
    print(line)

