"""Seeded synthetic RGB-D scenes with initial instance masks.

No dataset ships with the reference, so benchmarks and parity tests run on these: smooth random
BGR, a depth ramp with per-instance steps replicated to 3 channels (what ``normalize_depth``
produces, eval/preprocess_utils.py:24-27), and N non-empty initial masks laid out on a jittered
grid, each randomly grown/shrunk a few pixels to mimic the reference's mask perturbation
(tools/ours/perturbation_utils.py:22-71).  Some neighbours overlap so the "later mask overwrites"
rule of the encoder is exercised.
"""
import numpy as np


def _grid(n, h, w):
    cols = int(np.ceil(np.sqrt(n * w / h)))
    rows = int(np.ceil(n / cols))
    return rows, cols


def _shift_or(m, r):
    """binary dilation by a (2r+1)^2 square without scipy (bool [H,W])."""
    out = m.copy()
    for _ in range(r):
        p = np.pad(out, 1)
        out = p[1:-1, 1:-1] | p[:-2, 1:-1] | p[2:, 1:-1] | p[1:-1, :-2] | p[1:-1, 2:] \
            | p[:-2, :-2] | p[:-2, 2:] | p[2:, :-2] | p[2:, 2:]
    return out


def make_masks(rng, n, h, w, perturb=True):
    rows, cols = _grid(n, h, w)
    ch, cw = h / rows, w / cols
    yy, xx = np.mgrid[0:h, 0:w]
    gt = np.zeros((n, h, w), bool)
    init = np.zeros((n, h, w), bool)
    for i in range(n):
        r, c = divmod(i, cols)
        cy = (r + 0.5) * ch + rng.uniform(-0.12, 0.12) * ch
        cx = (c + 0.5) * cw + rng.uniform(-0.12, 0.12) * cw
        ry = rng.uniform(0.30, 0.46) * ch
        rx = rng.uniform(0.30, 0.46) * cw
        if rng.random() < 0.5:
            m = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
        else:
            m = (np.abs(yy - cy) <= ry) & (np.abs(xx - cx) <= rx)
        gt[i] = m
        if perturb:
            k = int(rng.integers(3, 11)) * max(1, min(h, w) // 480)
            if rng.random() < 0.5:
                pm = _shift_or(m, k)
            else:
                pm = ~_shift_or(~m, k)
                if pm.sum() < 64:
                    pm = m
            # small random shift
            pm = np.roll(pm, (int(rng.integers(-3, 4)), int(rng.integers(-3, 4))), axis=(0, 1))
            init[i] = pm
        else:
            init[i] = m
    return gt, init


def make_scene(seed, h=480, w=640, n=20, perturb=True):
    """-> dict(rgb u8[H,W,3] (BGR), depth u8[H,W,3], masks u8[N,H,W] in {0,255}, gt_masks u8[N,H,W])."""
    rng = np.random.default_rng(seed)
    low = rng.integers(0, 256, (h // 16 + 2, w // 16 + 2, 3)).astype(np.float32)
    rgb = np.kron(low, np.ones((16, 16, 1), np.float32))[:h, :w]
    rgb = np.clip(rgb + rng.normal(0, 6, rgb.shape), 0, 255).astype(np.uint8)
    gt, init = make_masks(rng, n, h, w, perturb)
    d = np.linspace(60, 200, h, dtype=np.float32)[:, None] * np.ones((1, w), np.float32)
    for i in range(n):
        d[gt[i]] -= rng.uniform(10, 40)
    d = np.clip(d, 0, 255).astype(np.uint8)
    depth = np.repeat(d[:, :, None], 3, axis=2)
    return {
        "rgb": np.ascontiguousarray(rgb),
        "depth": np.ascontiguousarray(depth),
        "masks": (init.astype(np.uint8) * 255),
        "gt_masks": (gt.astype(np.uint8) * 255),
    }


def make_batch(seed, b, h=480, w=640, n=20):
    scenes = [make_scene(seed * 1000 + i, h, w, n) for i in range(b)]
    return {k: np.stack([s[k] for s in scenes]) for k in scenes[0]}


def fake_head_outputs(enc, masks, rng=None, noise=0.0):
    """Plausible refined-head outputs derived from an encoding (for post-processing tests).

    enc f32 [3,H,W] (heat-map, off_y/H, off_x/W); masks u8 [N,H,W].
    -> fg logits [1,H,W], centre [1,H,W], offsets (pixels) [2,H,W], all f32.
    """
    h, w = enc.shape[1:]
    fg = (masks != 0).any(0)
    logit = np.where(fg, 4.0, -4.0).astype(np.float32)
    center = enc[0:1].copy()
    off = np.stack([enc[1] * h, enc[2] * w]).astype(np.float32)
    if rng is not None and noise > 0:
        logit = logit + rng.normal(0, noise, logit.shape).astype(np.float32)
        off = off + rng.normal(0, noise, off.shape).astype(np.float32)
        center = center + rng.normal(0, noise * 0.01, center.shape).astype(np.float32)
    return logit[None].astype(np.float32), center.astype(np.float32), off.astype(np.float32)
