"""ORACLE (test infrastructure, never shipped): adapter pre-processing.

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
