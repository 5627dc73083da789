"""CPU: the work distribution of the persistent convolution launches (csrc/conv_persist.hip) through its host view.
Every K-slice of every tile is computed exactly once; a partial tile's pieces are exactly what the fix-up pass sums, in K
order; no two pieces share a workspace slot; the blocks of an XCD get the same number of K-slices to within one tile (or one shortest share)."""
import ctypes as C

import numpy as np
import pytest

from quber_amd import _lib

CASES = [
    # tiles, blocks, K-slices per tile, shortest share (0 = remainder tiles whole)
    (2400, 768, 128, 4),      # fusion_res5.conv at batch 16: 3 rounds + 96 tiles shared
    (10800, 768, 8, 0),       # Winograd GEMM: 14 rounds + 48 whole remainder tiles
    (300, 768, 32, 4),        # fewer tiles than blocks: K shared throughout
    (1200, 768, 16, 0),
    (1350, 768, 8, 4),
    (600, 768, 64, 4),
    (7, 8, 5, 4),             # fewer tiles than XCD runs
    (9, 16, 3, 1),
    (1, 8, 64, 4),
    (777, 512, 33, 4),        # bf16x3 occupancy (512 blocks), odd everything
    (5000, 1792, 9, 4),       # 64x64 tiles: 7 blocks per CU
    (96, 768, 1, 4),          # single-slice tiles
    (2400, 512, 128, 4),      # round 3: two resident blocks per CU (two accumulator sets): fusion_res5.conv = 4 rounds + 352 tiles shared
    (10800, 512, 8, 0),       # Winograd GEMM on 512 blocks: 21 rounds + 48 whole remainder tiles
    (1200, 1280, 16, 0),      # 64x64 tiles at 5 blocks per CU
]


def segments(lib, T, P, nk, m, bid):
    buf = (C.c_int32 * (4 * 4096))()
    n = lib.quber_debug_persistent_segments(T, P, nk, m, bid, buf, 4096)
    assert 0 <= n < 4096
    return [tuple(buf[4 * i:4 * i + 4]) for i in range(n)]


@pytest.mark.parametrize("T,P,nk,m", CASES)
def test_every_k_slice_once_and_fixup_matches(T, P, nk, m):
    lib = _lib.load()
    cover = np.zeros((T, nk), np.int32)
    piece_of_slot = {}
    per_block = []
    for bid in range(P):
        segs = segments(lib, T, P, nk, m, bid)
        work = 0
        for (tile, k0, k1, slot) in segs:
            assert 0 <= tile < T and 0 <= k0 < k1 <= nk
            cover[tile, k0:k1] += 1
            work += k1 - k0
            if slot >= 0:
                assert (k1 - k0) < nk and slot not in piece_of_slot and 0 <= slot < 2 * P
                piece_of_slot[slot] = (tile, k0, k1)
            else:
                assert (k0, k1) == (0, nk)
        per_block.append(work)
    assert (cover == 1).all()
    # balance inside every XCD run: whole rounds are equal, the shared remainder differs by at most one K-slice
    for xcd in range(8):
        w = per_block[xcd::8]
        if max(w) > 0:
            assert max(w) - min(w) <= max(nk, m)             # never more than a tile (or one shortest share) apart
    # the fix-up pass sums exactly the pieces of each partial tile, in K order
    partial = {}
    for slot, (tile, k0, k1) in piece_of_slot.items():
        partial.setdefault(tile, []).append((k0, k1, slot))
    seen = set()
    for xcd in range(8):
        for j in range(P // 8 + 1):
            tile, slots = C.c_int32(-1), (C.c_int32 * 64)()
            n = lib.quber_debug_persistent_fixup(T, P, nk, m, xcd, j, C.byref(tile), slots, 64)
            if n < 0:
                break
            if n == 0:
                assert tile.value not in partial
                continue
            want = sorted(partial[tile.value])
            assert [s for (_, _, s) in want] == list(slots[:n])
            assert want[0][0] == 0 and want[-1][1] == nk and all(a[1] == b[0] for a, b in zip(want, want[1:]))
            seen.add(tile.value)
    assert seen == set(partial)
