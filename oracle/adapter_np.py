"""ORACLE (test infrastructure, never shipped): adapter pre-processing.

``cv2_resize_*`` restate OpenCV's 8-bit resize (cv2.resize as called at eval/refiner_model.py:229-232, 246, 254) from its
published algorithm - OpenCV is absent from the image, so these are **parity unpinned** (hand-derived cases in
tests/test_oracle_golden.py).

Restates eval/preprocess_utils.py:12-28 ``normalize_depth``.  Pinned: tests/golden/depthnorm_*.npz hold the outputs of
the imported reference function (cv2 is imported at that module's top but unused by it; an empty stand-in module was
registered for the import, see oracle/gen_golden.py)."""
import numpy as np


def normalize_depth(depth, min_val=250.0, max_val=1500.0):
    d = np.array(depth)                       # keeps the input dtype: uint16 stays integer until the subtraction
    d[d < min_val] = min_val
    d[d > max_val] = max_val
    d = (d - min_val) / (max_val - min_val) * 255
    if d.ndim == 2:
        d = d[..., None]
    return np.uint8(np.repeat(d, 3, -1))


def cv2_resize_nearest(img, dw, dh):
    """cv2.resize(img, (dw, dh), interpolation=cv2.INTER_NEAREST): sx = min(floor(dx / (dw / sw)), sw - 1)."""
    sh, sw = img.shape[:2]
    ifx, ify = 1.0 / (dw / sw), 1.0 / (dh / sh)
    xs = np.minimum(np.floor(np.arange(dw) * ifx).astype(np.int64), sw - 1)
    ys = np.minimum(np.floor(np.arange(dh) * ify).astype(np.int64), sh - 1)
    return np.ascontiguousarray(img[ys][:, xs])


def _lin_coef(dsize, ssize):
    scale = 1.0 / (dsize / ssize)
    f = ((np.arange(dsize) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    lo, hi = s < 0, s >= ssize - 1
    f[lo | hi] = 0
    s[lo] = 0
    s[hi] = ssize - 1
    a1 = np.rint(f * np.float32(2048)).astype(np.int64)          # cvRound: half to even
    a0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
    return s, np.minimum(s + 1, ssize - 1), a0, a1


def cv2_resize_linear_u8(img, dw, dh):
    """cv2.resize(img, (dw, dh)) (INTER_LINEAR) for uint8: 11-bit fixed-point weights, int32 horizontal pass, the vertical
    pass's shift sequence; exactly half scale in both directions runs the 2x2 area filter instead."""
    img = np.asarray(img, np.uint8)
    sh, sw = img.shape[:2]
    x = img.astype(np.int64)
    if sw == 2 * dw and sh == 2 * dh:
        return ((x[0::2, 0::2] + x[0::2, 1::2] + x[1::2, 0::2] + x[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    x0, x1, a0, a1 = _lin_coef(dw, sw)
    y0, y1, b0, b1 = _lin_coef(dh, sh)
    tail = (None,) * (img.ndim - 2)                       # broadcast the weights over the channel axis, if any
    hrz = x[:, x0] * a0[(None, slice(None)) + tail] + x[:, x1] * a1[(None, slice(None)) + tail]      # int32 in OpenCV
    b0, b1 = b0[(slice(None), None) + tail], b1[(slice(None), None) + tail]
    out = (((b0 * (hrz[y0] >> 4)) >> 16) + ((b1 * (hrz[y1] >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)
