#!/usr/bin/env python3
"""Where a K-slice's time goes inside csrc/conv_x8.hip (diagnostic build in a scratch copy: QUBER_LIB=$(tools/diag_build.sh x8stamps X8X=-DX8_STAMPS)).  s_memtime of waves 0 and 4
(SIMD partners: wave 4 runs half a phase behind) of every block at the phase boundaries of K-slice 8 of the block's last tile:
0 slice start | 1 reads + split of k-step 0 done | 2 DMA issued | 3 MFMAs of phase 0 + both barriers | 4 reads + split of k-step 1 | 5 DMA wait | 6 MFMAs of phase 1.
usage (GPU box): x8_stamps.py [layer substring of tools/x8_bench.py]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quber_amd import _lib  # noqa: E402

layer = sys.argv[1] if len(sys.argv) > 1 else "fusion_res5.conv"
lib = _lib.load()
raw = C.CDLL(_lib.LIB_PATH)
sys.argv = [sys.argv[0], "--only", layer, "--iters", "2"]
import x8_bench  # noqa: E402
x8_bench.main()
N = 256 * 2 * 16
buf = (C.c_ulonglong * N)()
assert raw.quber_x8_read_stamps(buf, N) == 0
s = np.array(buf[:], dtype=np.int64).reshape(256, 2, 16).astype(np.float64)
names = ["reads + split 0", "DMA issue", "barrier + 24 MFMAs + barrier", "reads + split 1", "DMA wait", "barrier + 24 MFMAs + barrier"]
for w in (0, 1):
    v = s[:, w]
    ok = v[:, 6] > v[:, 0]
    d = np.diff(v[ok][:, :7], axis=1)
    print(f"wave {4 * w}: K-slice p50 {np.median(v[ok][:, 6] - v[ok][:, 0]):6.0f} cycles (matrix pipe, both waves of the SIMD: 3 072) = " +
          ", ".join(f"{n} {np.median(d[:, i]):5.0f}" for i, n in enumerate(names)))
