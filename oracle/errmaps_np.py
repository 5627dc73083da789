"""ORACLE (test infrastructure, never shipped): explicit quadruple error maps.

Restates
  * explicit_error_estimation/util.py:62-68   masks_to_fg_mask  (uint8 wrap-around sum, then > 0)
  * explicit_error_estimation/util.py:72-90   mask_to_boundary  (1-px zero pad, 3x3 erode x d, mask - eroded)
  * explicit_error_estimation/util.py:92-99   masks_to_boundary (uint8 wrap-around sum of bands, then > 0)
  * tools/ours/panoptic2eee.py:110-123        TP/TN/FP/FN = and/not of (gt, input) maps
Channel order TP,TN,FP,FN follows maskrefiner/modeling/mask_refiner/model.py:187-190.

``masks_to_fg_mask`` is pinned by tests/golden/fgunion_*.npz (generated from the imported
reference function).  ``mask_to_boundary`` needs cv2 in the reference (absent here) so its parity
is **unpinned**: the erosion is restated as a running 3x3 minimum with a zero border, which is what
cv2.erode does on the zero-padded mask (iterations=d), and checked on hand-derived rectangles.
"""
import numpy as np


def masks_to_fg_mask(masks):
    acc = np.zeros_like(masks[0])
    for m in masks:
        acc = acc + m                      # same dtype as the masks: uint8 wraps like the reference
    return (acc > 0).astype(np.uint8)


def boundary_width(h, w, ratio):
    d = int(round(ratio * np.sqrt(h ** 2 + w ** 2)))
    return max(d, 1)


def erode3x3(mask, iterations):
    """iterated 3x3 minimum filter; everything outside the image counts as 0."""
    h, w = mask.shape
    cur = mask.copy()
    for _ in range(iterations):
        p = np.zeros((h + 2, w + 2), mask.dtype)
        p[1:-1, 1:-1] = cur
        nxt = p[1:-1, 1:-1].copy()
        for dy in (0, 1, 2):
            for dx in (0, 1, 2):
                nxt = np.minimum(nxt, p[dy:dy + h, dx:dx + w])
        cur = nxt
    return cur


def mask_to_boundary(mask, dilation_ratio=0.02):
    h, w = mask.shape
    d = boundary_width(h, w, dilation_ratio)
    return mask - erode3x3(mask, d)


def masks_to_boundary(masks, dilation_ratio=0.01):
    acc = np.zeros_like(masks_to_fg_mask(masks))
    for m in masks:
        acc = acc + mask_to_boundary(m, dilation_ratio).astype(acc.dtype)
    return (acc > 0).astype(np.uint8)


def quadruple(gt, inp):
    gt, inp = gt.astype(bool), inp.astype(bool)
    return np.stack([gt & inp, ~gt & ~inp, ~gt & inp, gt & ~inp]).astype(np.uint8)


def explicit_error_maps(init_masks, gt_masks, dilation_ratio=0.01):
    """-> uint8 [2,4,H,W]: [region, boundary] x [TP,TN,FP,FN]."""
    region = quadruple(masks_to_fg_mask(gt_masks), masks_to_fg_mask(init_masks))
    bnd = quadruple(masks_to_boundary(gt_masks, dilation_ratio), masks_to_boundary(init_masks, dilation_ratio))
    return np.stack([region, bnd])
