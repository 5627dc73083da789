"""ORACLE (test infrastructure, never shipped): cv2.inpaint(img, mask, radius, cv2.INPAINT_TELEA) for one 8-bit channel, as
``inpaint_depth`` calls it (eval/preprocess_utils.py:44-64).

**Parity unpinned**: OpenCV is absent from the image and from /root/reference.  This restates Telea's fast-marching
in-painting (J. Graphics Tools 9(1), 2004) in the form OpenCV implements it, step for step like
quber_amd/csrc/inpaint.hip but written independently (dict-free heapq loop, numpy arrays), so the two check each other;
tests/test_oracle_golden.py holds it to hand-derivable properties (constant and linear-ramp images)."""
import heapq
import math

import numpy as np

KNOWN, BAND, INSIDE, CHANGE = 0, 1, 2, 3
NB4 = ((-1, 0), (0, -1), (1, 0), (0, 1))


def _solve(t, f, i1, j1, i2, j2):
    a11, a22 = float(t[i1, j1]), float(t[i2, j2])
    m12 = min(a11, a22)
    k1, k2 = f[i1, j1] != INSIDE, f[i2, j2] != INSIDE
    if k1:
        if k2:
            sol = 1 + m12 if abs(a11 - a22) >= 1.0 else (a11 + a22 + math.sqrt(2 - (a11 - a22) ** 2)) * 0.5
        else:
            sol = 1 + a11
    elif k2:
        sol = 1 + a22
    else:
        sol = 1 + m12
    return np.float32(sol)


def _dist(t, f, i, j):
    return min(_solve(t, f, i - 1, j, i, j - 1), _solve(t, f, i + 1, j, i, j - 1),
               _solve(t, f, i - 1, j, i, j + 1), _solve(t, f, i + 1, j, i, j + 1))


def inpaint_telea_u8(img, mask, radius=3):
    img = np.asarray(img, np.uint8)
    h, w = img.shape
    rng = max(1, min(100, int(radius)))
    R, C = h + 2, w + 2
    m = np.zeros((R, C), bool)
    m[1:-1, 1:-1] = np.asarray(mask) != 0
    f = np.full((R, C), KNOWN, np.uint8)
    t = np.full((R, C), 1.0e6, np.float32)
    f[m] = INSIDE
    nb = np.zeros_like(m)
    nb[1:-1, 1:-1] = m[:-2, 1:-1] | m[2:, 1:-1] | m[1:-1, :-2] | m[1:-1, 2:]
    band = nb & ~m
    f[band] = BAND
    t[band] = 0
    heap, seq = [], 0
    for i, j in zip(*np.nonzero(band)):
        heap.append((np.float32(0), seq, int(i), int(j)))
        seq += 1
    heap_out = list(heap)
    heapq.heapify(heap)
    heapq.heapify(heap_out)
    # outside pass over the known pixels within the (2 rng + 1)^2 neighbourhood of the hole
    near = np.zeros_like(m)
    ys, xs = np.nonzero(m)
    for y, x in zip(ys, xs):
        near[max(1, y - rng):min(R - 2, y + rng) + 1, max(1, x - rng):min(C - 2, x + rng) + 1] = True
    ring = np.full((R, C), KNOWN, np.uint8)
    ring[near & ~m & ~band] = INSIDE
    ring[0, :] = ring[-1, :] = ring[:, 0] = ring[:, -1] = KNOWN
    while heap_out:
        _, _, ii, jj = heapq.heappop(heap_out)
        ring[ii, jj] = CHANGE
        for di, dj in NB4:
            i, j = ii + di, jj + dj
            if i <= 0 or j <= 0 or i >= R - 1 or j >= C - 1 or ring[i, j] != INSIDE:
                continue
            d = _dist(t, ring, i, j)
            t[i, j] = d
            ring[i, j] = BAND
            heapq.heappush(heap_out, (d, seq, i, j))
            seq += 1
    neg = (ring == CHANGE) & (f != BAND)
    t[neg] = -t[neg]

    out = img.copy()
    I = lambda y, x: np.float32(out[y, x])
    f32 = np.float32
    while heap:
        _, _, ii, jj = heapq.heappop(heap)
        f[ii, jj] = KNOWN
        for di, dj in NB4:
            i, j = ii + di, jj + dj
            if i <= 0 or j <= 0 or i >= R - 1 or j >= C - 1 or f[i, j] != INSIDE:
                continue
            dist = _dist(t, f, i, j)
            t[i, j] = dist
            if f[i, j + 1] != INSIDE:
                gtx = (t[i, j + 1] - t[i, j - 1]) * f32(0.5) if f[i, j - 1] != INSIDE else t[i, j + 1] - t[i, j]
            else:
                gtx = t[i, j] - t[i, j - 1] if f[i, j - 1] != INSIDE else f32(0)
            if f[i + 1, j] != INSIDE:
                gty = (t[i + 1, j] - t[i - 1, j]) * f32(0.5) if f[i - 1, j] != INSIDE else t[i + 1, j] - t[i, j]
            else:
                gty = t[i, j] - t[i - 1, j] if f[i - 1, j] != INSIDE else f32(0)
            Ia, Jx, Jy, s = f32(0), f32(0), f32(0), f32(1.0e-20)
            for k in range(i - rng, i + rng + 1):
                km, kp = k - 1 + (k == 1), k - 1 - (k == R - 2)
                for l in range(j - rng, j + rng + 1):
                    lm, lp = l - 1 + (l == 1), l - 1 - (l == C - 2)
                    if not (0 < k < R - 1 and 0 < l < C - 1):
                        continue
                    if f[k, l] == INSIDE or (l - j) ** 2 + (k - i) ** 2 > rng * rng:
                        continue
                    ry, rx = f32(i - k), f32(j - l)
                    len2 = rx * rx + ry * ry
                    dst = f32(1.0 / (float(len2) * math.sqrt(float(len2))))
                    lev = f32(1.0 / (1 + abs(float(t[k, l]) - float(t[i, j]))))
                    dr = rx * gtx + ry * gty
                    if abs(dr) <= f32(0.01):
                        dr = f32(0.000001)
                    wgt = f32(abs(dst * lev * dr))
                    if f[k, l + 1] != INSIDE:
                        gix = (I(km, lp + 1) - I(km, lm - 1)) * f32(2) if f[k, l - 1] != INSIDE else I(km, lp + 1) - I(km, lm)
                    else:
                        gix = I(km, lp) - I(km, lm - 1) if f[k, l - 1] != INSIDE else f32(0)
                    if f[k + 1, l] != INSIDE:
                        giy = (I(kp + 1, lm) - I(km - 1, lm)) * f32(2) if f[k - 1, l] != INSIDE else I(kp + 1, lm) - I(km, lm)
                    else:
                        giy = I(kp, lm) - I(km - 1, lm) if f[k - 1, l] != INSIDE else f32(0)
                    Ia += wgt * I(km, lm)
                    Jx -= wgt * gix * rx
                    Jy -= wgt * giy * ry
                    s += wgt
            sat = Ia / s + (Jx + Jy) / (f32(math.sqrt(float(Jx * Jx + Jy * Jy))) + f32(1.0e-20)) + f32(0.5)
            out[i - 1, j - 1] = np.uint8(max(0.0, min(255.0, math.floor(float(sat)))))
            f[i, j] = BAND
            heapq.heappush(heap, (dist, seq, i, j))
            seq += 1
    return out


def inpaint_depth(depth3, kernel_size=3):
    """eval/preprocess_utils.py:44-64 with factor = 1: mask = all channels zero, dilated by a kernel_size^2 square; TELEA with
    radius kernel_size; only the zero pixels are replaced."""
    from scipy import ndimage
    depth3 = np.asarray(depth3, np.uint8)
    mask = np.all(depth3 == 0, axis=2)
    mask = ndimage.binary_dilation(mask, structure=np.ones((kernel_size, kernel_size), bool))
    filled = np.stack([inpaint_telea_u8(depth3[..., c], mask, kernel_size) for c in range(depth3.shape[2])], -1)
    return np.where(depth3 == 0, filled, depth3)
