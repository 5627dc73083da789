// Adapter-side depth in-painting: the reference fills the holes of the normalised depth image with
// cv2.inpaint(..., 3, cv2.INPAINT_TELEA) (eval/preprocess_utils.py:44-64) before the frame reaches the network.
//
// HOST code, like the reference's (OpenCV runs it on the CPU, once per frame, outside the refiner's timed region): Telea's
// fast-marching method visits the unknown pixels strictly in order of their distance to the hole's border through one
// priority queue - a sequential algorithm with nothing for a GPU to do on holes of a few thousand pixels.
//
// OpenCV is not in the build image and its source is not in the reference, so this restates the published algorithm
// (A. Telea, "An image inpainting technique based on the fast marching method", J. Graphics Tools 9(1), 2004) in the form
// OpenCV implements it, as recalled: a one-pixel frame around the image; T = 0 on the known pixels 4-adjacent to the hole,
// T computed inward by the fast-marching eikonal update and, negated, outward over a (2r+1)^2 neighbourhood of the hole;
// a pixel is filled, when it joins the narrow band, from the known pixels within radius r with the weights
// |dir * dst * lev| (alignment with grad T, inverse cubed distance, closeness of the level sets) plus the normalised
// gradient-correction term.  Parity is UNPINNED and the stage carries its own tolerance (tests/test_oracle_golden.py:
// constant and linear-ramp images are reproduced to +-1 / +-2 grey levels; oracle/inpaint_np.py restates the same steps
// independently in numpy).
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <queue>
#include <vector>

#include "common.h"

namespace quber {
namespace {

enum : uint8_t { KNOWN = 0, BAND = 1, INSIDE = 2, CHANGE = 3 };

struct Node {
    float t;
    uint32_t seq;     // first in, first out among equal T
    int i, j;
    bool operator<(const Node& o) const { return t != o.t ? t > o.t : seq > o.seq; }
};

struct Grid {
    int rows, cols;   // padded
    std::vector<uint8_t> f;
    std::vector<float> t;
    uint8_t& F(int i, int j) { return f[(size_t)i * cols + j]; }
    float& T(int i, int j) { return t[(size_t)i * cols + j]; }
};

float fm_solve(Grid& g, std::vector<uint8_t>& f, int i1, int j1, int i2, int j2) {
    const double a11 = g.T(i1, j1), a22 = g.T(i2, j2), m12 = std::min(a11, a22);
    const bool k1 = f[(size_t)i1 * g.cols + j1] != INSIDE, k2 = f[(size_t)i2 * g.cols + j2] != INSIDE;
    double sol;
    if (k1) {
        if (k2) sol = fabs(a11 - a22) >= 1.0 ? 1 + m12 : (a11 + a22 + sqrt(2 - (a11 - a22) * (a11 - a22))) * 0.5;
        else sol = 1 + a11;
    } else if (k2) {
        sol = 1 + a22;
    } else {
        sol = 1 + m12;
    }
    return (float)sol;
}

float fm_dist(Grid& g, std::vector<uint8_t>& f, int i, int j) {
    return std::min(std::min(fm_solve(g, f, i - 1, j, i, j - 1), fm_solve(g, f, i + 1, j, i, j - 1)),
                    std::min(fm_solve(g, f, i - 1, j, i, j + 1), fm_solve(g, f, i + 1, j, i, j + 1)));
}

}  // namespace

int inpaint_telea_u8_host(const uint8_t* img, const uint8_t* mask, int H, int W, int radius, uint8_t* out) {
    if (!img || !mask || !out || H < 1 || W < 1) return fail("inpaint: bad argument");
    const int range = std::max(1, std::min(100, radius));
    Grid g;
    g.rows = H + 2;
    g.cols = W + 2;
    const int R = g.rows, C = g.cols;
    g.f.assign((size_t)R * C, KNOWN);
    g.t.assign((size_t)R * C, 1.0e6f);
    std::vector<uint8_t> m((size_t)R * C, 0), ring((size_t)R * C, KNOWN);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            out[(size_t)y * W + x] = img[(size_t)y * W + x];
            if (mask[(size_t)y * W + x]) m[(size_t)(y + 1) * C + x + 1] = 1;
        }
    auto M = [&](int i, int j) -> uint8_t { return m[(size_t)i * C + j]; };
    // band: known pixels with a 4-neighbour in the hole (3x3 cross dilation minus the hole), frame excluded
    std::priority_queue<Node> heap, heap_out;
    uint32_t seq = 0;
    for (int i = 1; i < R - 1; ++i)
        for (int j = 1; j < C - 1; ++j) {
            if (M(i, j)) {
                g.F(i, j) = INSIDE;
            } else if (M(i - 1, j) || M(i + 1, j) || M(i, j - 1) || M(i, j + 1)) {
                g.F(i, j) = BAND;
                g.T(i, j) = 0.f;
                heap.push({0.f, seq, i, j});
                heap_out.push({0.f, seq, i, j});
                ++seq;
            }
        }
    // outside pass: T over the known pixels within the (2 range + 1)^2 neighbourhood of the hole, negated afterwards
    for (int i = 1; i < R - 1; ++i)
        for (int j = 1; j < C - 1; ++j) {
            if (M(i, j) || g.F(i, j) == BAND) continue;
            bool near = false;
            for (int k = std::max(1, i - range); k <= std::min(R - 2, i + range) && !near; ++k)
                for (int l = std::max(1, j - range); l <= std::min(C - 2, j + range); ++l)
                    if (M(k, l)) { near = true; break; }
            if (near) ring[(size_t)i * C + j] = INSIDE;
        }
    while (!heap_out.empty()) {
        const Node n = heap_out.top();
        heap_out.pop();
        ring[(size_t)n.i * C + n.j] = CHANGE;
        const int di[4] = {-1, 0, 1, 0}, dj[4] = {0, -1, 0, 1};
        for (int q = 0; q < 4; ++q) {
            const int i = n.i + di[q], j = n.j + dj[q];
            if (i <= 0 || j <= 0 || i > R - 1 || j > C - 1) continue;
            if (i >= R - 1 || j >= C - 1) continue;                 // the frame itself is never marched
            if (ring[(size_t)i * C + j] != INSIDE) continue;
            const float d = fm_dist(g, ring, i, j);
            g.T(i, j) = d;
            ring[(size_t)i * C + j] = BAND;
            heap_out.push({d, seq++, i, j});
        }
    }
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < C; ++j)
            if (ring[(size_t)i * C + j] == CHANGE && g.F(i, j) != BAND) g.T(i, j) = -g.T(i, j);

    auto I = [&](int y, int x) -> float { return (float)out[(size_t)y * W + x]; };    // unpadded image coordinates
    while (!heap.empty()) {
        const Node n = heap.top();
        heap.pop();
        g.F(n.i, n.j) = KNOWN;
        const int di[4] = {-1, 0, 1, 0}, dj[4] = {0, -1, 0, 1};
        for (int q = 0; q < 4; ++q) {
            const int i = n.i + di[q], j = n.j + dj[q];
            if (i <= 0 || j <= 0 || i >= R - 1 || j >= C - 1) continue;
            if (g.F(i, j) != INSIDE) continue;
            const float dist = fm_dist(g, g.f, i, j);
            g.T(i, j) = dist;
            float gtx, gty;
            if (g.F(i, j + 1) != INSIDE) gtx = g.F(i, j - 1) != INSIDE ? (g.T(i, j + 1) - g.T(i, j - 1)) * 0.5f : g.T(i, j + 1) - g.T(i, j);
            else gtx = g.F(i, j - 1) != INSIDE ? g.T(i, j) - g.T(i, j - 1) : 0.f;
            if (g.F(i + 1, j) != INSIDE) gty = g.F(i - 1, j) != INSIDE ? (g.T(i + 1, j) - g.T(i - 1, j)) * 0.5f : g.T(i + 1, j) - g.T(i, j);
            else gty = g.F(i - 1, j) != INSIDE ? g.T(i, j) - g.T(i - 1, j) : 0.f;
            float Ia = 0.f, Jx = 0.f, Jy = 0.f, s = 1.0e-20f;
            for (int k = i - range; k <= i + range; ++k) {
                const int km = k - 1 + (k == 1), kp = k - 1 - (k == R - 2);
                for (int l = j - range; l <= j + range; ++l) {
                    const int lm = l - 1 + (l == 1), lp = l - 1 - (l == C - 2);
                    if (!(k > 0 && l > 0 && k < R - 1 && l < C - 1)) continue;
                    if (g.F(k, l) == INSIDE || (l - j) * (l - j) + (k - i) * (k - i) > range * range) continue;
                    const float ry = (float)(i - k), rx = (float)(j - l);
                    const float len2 = rx * rx + ry * ry;
                    const float dst = (float)(1. / (len2 * sqrt((double)len2)));
                    const float lev = (float)(1. / (1 + fabs(g.T(k, l) - g.T(i, j))));
                    float dir = rx * gtx + ry * gty;
                    if (fabsf(dir) <= 0.01f) dir = 0.000001f;
                    const float w = fabsf(dst * lev * dir);
                    float gix, giy;
                    if (g.F(k, l + 1) != INSIDE) gix = g.F(k, l - 1) != INSIDE ? (I(km, lp + 1) - I(km, lm - 1)) * 2.0f : I(km, lp + 1) - I(km, lm);
                    else gix = g.F(k, l - 1) != INSIDE ? I(km, lp) - I(km, lm - 1) : 0.f;
                    if (g.F(k + 1, l) != INSIDE) giy = g.F(k - 1, l) != INSIDE ? (I(kp + 1, lm) - I(km - 1, lm)) * 2.0f : I(kp + 1, lm) - I(km, lm);
                    else giy = g.F(k - 1, l) != INSIDE ? I(kp, lm) - I(km - 1, lm) : 0.f;
                    Ia += w * I(km, lm);
                    Jx -= w * gix * rx;
                    Jy -= w * giy * ry;
                    s += w;
                }
            }
            const float sat = Ia / s + (Jx + Jy) / (sqrtf(Jx * Jx + Jy * Jy) + 1.0e-20f) + 0.5f;
            out[(size_t)(i - 1) * W + (j - 1)] = (uint8_t)std::max(0.f, std::min(255.f, floorf(sat)));
            g.F(i, j) = BAND;
            heap.push({dist, seq++, i, j});
        }
    }
    return 0;
}

// inpaint_depth of eval/preprocess_utils.py:44-64 (factor 1) on one 3-channel 8-bit depth image, whole: mask = pixels whose three
// channels are all 0, dilated by a kernel x kernel square; cv2.inpaint(..., kernel, INPAINT_TELEA) per channel; only the pixels that
// were 0 take the filled value.  One call = one GIL-free stretch for the Python adapter's worker threads (the numpy form of the
// mask preparation held the interpreter lock for ~2 ms per frame: tools/stream_probe.py).
int inpaint_depth_u8_host(const uint8_t* depth3, int H, int W, int kernel, uint8_t* out3) {
    if (!depth3 || !out3 || H < 1 || W < 1 || kernel < 1 || !(kernel & 1)) return fail("inpaint_depth: bad argument");
    const size_t n = (size_t)H * W;
    std::vector<uint8_t> zero(n), mask(n, 0), src(n), dst(n);
    bool any = false, same = true;
    for (size_t i = 0; i < n; ++i) {
        const uint8_t a = depth3[3 * i], b = depth3[3 * i + 1], c = depth3[3 * i + 2];
        zero[i] = (a | b | c) == 0;
        any |= zero[i] != 0;
        same &= a == b && a == c;
        out3[3 * i] = a; out3[3 * i + 1] = b; out3[3 * i + 2] = c;
    }
    if (!any) return 0;
    const int r = kernel / 2;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            if (!zero[(size_t)y * W + x]) continue;
            for (int dy = -r; dy <= r; ++dy)
                for (int dx = -r; dx <= r; ++dx) {
                    const int yy = y + dy, xx = x + dx;
                    if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) mask[(size_t)yy * W + xx] = 1;
                }
        }
    for (int c = 0; c < (same ? 1 : 3); ++c) {           // normalize_depth replicates one channel: in-paint it once
        for (size_t i = 0; i < n; ++i) src[i] = depth3[3 * i + c];
        if (int rc = inpaint_telea_u8_host(src.data(), mask.data(), H, W, kernel, dst.data())) return rc;
        for (size_t i = 0; i < n; ++i) {
            // np.where(depth == 0, filled, depth), element-wise per channel
            if (same) {
                if (depth3[3 * i] == 0) out3[3 * i] = out3[3 * i + 1] = out3[3 * i + 2] = dst[i];
            } else if (depth3[3 * i + c] == 0) {
                out3[3 * i + c] = dst[i];
            }
        }
    }
    return 0;
}

}  // namespace quber
