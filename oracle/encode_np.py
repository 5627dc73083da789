"""ORACLE (test infrastructure, never shipped): initial-mask -> (centre heat-map, y/x offsets) encoding.

Restates maskrefiner/predictor.py:246-251 (Gaussian template) and :304-357 (per-mask loop); the
identical arithmetic also lives in explicit_error_estimation/util.py:142-228
(``PerturbedInputOffsetGenerator``), which IS importable in the build container and pins this
file: tests/golden/encode_*.npz were produced by oracle/gen_golden.py from that class.

Arithmetic that matters for bit-exactness:
  * centroid = float64 mean of the integer pixel indices (np.mean of int64),
  * integer centre = Python round() (half to even) of that float64,
  * heat-map = max(existing f32, f32(float64 Gaussian)),
  * offsets = f32((centroid - coord) / extent): `np.float64 scalar - float32 array`.  Under numpy >= 2 (NEP 50; this
    image, and what the golden fixtures pin) that is float64 arithmetic rounded once; under the reference's own pinned
    numpy==1.23.1 (INSTALL.md:14) value-based casting keeps it in float32 throughout.  ``legacy_promotion=True`` restates
    the latter with explicit casts - **parity unpinned** for that mode (numpy 1.x cannot be run here); it differs from
    the pinned mode by 1 ulp on about a third of the mask pixels (max 6e-8),
  * later masks overwrite earlier ones in the offset planes; empty masks are skipped.
"""
import numpy as np


def gaussian_template(sigma=10):
    size = 6 * sigma + 3
    ax = np.arange(size, dtype=np.float64)
    c = 3 * sigma + 1
    return np.exp(-((ax[None, :] - c) ** 2 + (ax[:, None] - c) ** 2) / (2 * sigma ** 2))


def encode_initial_masks(masks, height=None, width=None, sigma=10, legacy_promotion=False):
    """masks: iterable of [H,W] arrays (non-zero = inside). Returns float32 [3,H,W]."""
    masks = list(masks)
    if height is None:
        height, width = masks[0].shape
    g = gaussian_template(sigma)
    r = 3 * sigma + 1
    heat = np.zeros((height, width), np.float32)
    off = np.zeros((2, height, width), np.float32)
    for m in masks:
        ys, xs = np.nonzero(m)
        if ys.size == 0:
            continue
        cy, cx = ys.mean(), xs.mean()                      # float64
        iy, ix = int(round(cy)), int(round(cx))            # banker's rounding
        x0, y0 = ix - r, iy - r                            # upper-left of the 63x63 window
        x1, y1 = ix + r + 1, iy + r + 1
        wx0, wx1 = max(x0, 0), min(x1, width)
        wy0, wy1 = max(y0, 0), min(y1, height)
        if wx1 > wx0 and wy1 > wy0:
            win = g[wy0 - y0:wy1 - y0, wx0 - x0:wx1 - x0]
            heat[wy0:wy1, wx0:wx1] = np.maximum(heat[wy0:wy1, wx0:wx1], win)
        if legacy_promotion:
            off[0, ys, xs] = (np.float32(cy) - ys.astype(np.float32)) / np.float32(height)
            off[1, ys, xs] = (np.float32(cx) - xs.astype(np.float32)) / np.float32(width)
        else:
            off[0, ys, xs] = (cy - ys.astype(np.float32)) / height
            off[1, ys, xs] = (cx - xs.astype(np.float32)) / width
    return np.stack([heat, off[0], off[1]]).astype(np.float32)
