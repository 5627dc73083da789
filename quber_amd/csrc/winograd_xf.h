// Transform arithmetic shared by the Winograd kernels (winograd.hip: the three-kernel pipeline; wino_fused.hip: the
// single-kernel F(4x4,3x3) layer): B^T / A^T in one dimension, the tile index map.
#pragma once
#include "common.h"

namespace quber {
namespace wxf {

// V consecutive channels of one pixel (V = 4: 16-byte accesses; V = 2 for the 8x8 transforms of m = 6, whose 64
// intermediate values per channel would not fit the register file four channels at a time)
template <int V>
struct Vec {
    float v[V];
    __device__ inline Vec operator+(const Vec& o) const { Vec r; for (int e = 0; e < V; ++e) r.v[e] = v[e] + o.v[e]; return r; }
    __device__ inline Vec operator-(const Vec& o) const { Vec r; for (int e = 0; e < V; ++e) r.v[e] = v[e] - o.v[e]; return r; }
};
template <int V>
__device__ inline Vec<V> operator*(float s, const Vec<V>& a) { Vec<V> r; for (int e = 0; e < V; ++e) r.v[e] = s * a.v[e]; return r; }
template <int V>
__device__ inline Vec<V> vzero() { Vec<V> r; for (int e = 0; e < V; ++e) r.v[e] = 0.f; return r; }
template <int V>
__device__ inline Vec<V> vload(const float* p) {
    Vec<V> r;
    if constexpr (V == 4) { const float4 t = *reinterpret_cast<const float4*>(p); r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w; }
    else { const float2 t = *reinterpret_cast<const float2*>(p); r.v[0] = t.x; r.v[1] = t.y; }
    return r;
}
template <int V>
__device__ inline void vstore(float* p, const Vec<V>& a) {
    if constexpr (V == 4) *reinterpret_cast<float4*>(p) = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]);
    else *reinterpret_cast<float2*>(p) = make_float2(a.v[0], a.v[1]);
}

// y = B^T x  (one dimension)
template <int O, class T>
__device__ inline void bt(const T* x, T* y) {
    if constexpr (O == 2) {
        y[0] = x[0] - x[2];
        y[1] = x[1] + x[2];
        y[2] = x[2] - x[1];
        y[3] = x[1] - x[3];
    } else if constexpr (O == 4) {
        // points 0, +-3/4, +-3/2, inf (all constants dyadic, hence exact in fp32)
        const T e1 = x[4] - 2.25f * x[2], o1 = 0.75f * x[3] - 1.6875f * x[1];
        const T e2 = x[4] - 0.5625f * x[2], o2 = 1.5f * x[3] - 0.84375f * x[1];
        y[0] = (1.265625f * x[0] - 2.8125f * x[2]) + x[4];
        y[1] = e1 + o1;
        y[2] = e1 - o1;
        y[3] = e2 + o2;
        y[4] = e2 - o2;
        y[5] = (1.265625f * x[1] - 2.8125f * x[3]) + x[5];
    } else {
        // points 0, +-1, +-2, +-1/2, inf (Lavin & Gray / wincnn)
        const T a = (x[2] + x[6]) - 4.25f * x[4], b = (x[1] + x[5]) - 4.25f * x[3];
        const T c = (0.25f * x[2] - 1.25f * x[4]) + x[6], d = (0.5f * x[1] - 2.5f * x[3]) + 2.f * x[5];
        const T e = (4.f * x[2] - 5.f * x[4]) + x[6], f = (2.f * x[1] - 2.5f * x[3]) + 0.5f * x[5];
        y[0] = (x[0] - x[6]) + 5.25f * (x[4] - x[2]);
        y[1] = a + b;
        y[2] = a - b;
        y[3] = c + d;
        y[4] = c - d;
        y[5] = e + f;
        y[6] = e - f;
        y[7] = (x[7] - x[1]) + 5.25f * (x[3] - x[5]);
    }
}

// y = A^T x  (one dimension)
template <int O, class T>
__device__ inline void at(const T* x, T* y) {
    if constexpr (O == 2) {
        y[0] = (x[0] + x[1]) + x[2];
        y[1] = (x[1] - x[2]) - x[3];
    } else if constexpr (O == 4) {
        const T a = x[1] + x[2], b = x[1] - x[2], c = x[3] + x[4], d = x[3] - x[4];
        y[0] = (x[0] + a) + c;
        y[1] = 0.75f * b + 1.5f * d;
        y[2] = 0.5625f * a + 2.25f * c;
        y[3] = (0.421875f * b + 3.375f * d) + x[5];
    } else {
        const T s1 = x[1] + x[2], d1 = x[1] - x[2], s2 = x[3] + x[4], d2 = x[3] - x[4], s3 = x[5] + x[6], d3 = x[5] - x[6];
        y[0] = ((x[0] + s1) + s2) + s3;
        y[1] = (d1 + 2.f * d2) + 0.5f * d3;
        y[2] = (s1 + 4.f * s2) + 0.25f * s3;
        y[3] = (d1 + 8.f * d2) + 0.125f * d3;
        y[4] = (s1 + 16.f * s2) + 0.0625f * s3;
        y[5] = ((d1 + 32.f * d2) + 0.03125f * d3) + x[7];
    }
}

// tile index -> image, phase and tile position
struct TileAt { int b, py, px, ty, tx; };
__device__ inline TileAt locate(long tile, int TH, int TW, int d) {
    TileAt t;
    t.tx = tile % TW;
    long r = tile / TW;
    t.ty = r % TH;
    r /= TH;
    t.px = r % d;
    r /= d;
    t.py = r % d;
    t.b = (int)(r / d);
    return t;
}


// the same on 32-bit indices (the single-kernel form checks tiles < 2^31: 64-bit divisions cost its prologue ~100 instructions)
__device__ inline TileAt locate32(unsigned tile, unsigned TH, unsigned TW, unsigned d) {
    TileAt t;
    unsigned r = tile / TW;
    t.tx = (int)(tile - r * TW);
    unsigned r2 = r / TH;
    t.ty = (int)(r - r2 * TH);
    unsigned r3 = r2 / d;
    t.px = (int)(r2 - r3 * d);
    const unsigned r4 = r3 / d;
    t.py = (int)(r3 - r4 * d);
    t.b = (int)r4;
    return t;
}

// tiles per image: d*d phases of ceil(ceil(H/d)/m) x ceil(ceil(W/d)/m) tiles
inline int tiles_1d(int n, int d, int m) { return ((n + d - 1) / d + m - 1) / m; }
inline long wino_tiles(int H, int W, int d, int m) { return (long)d * d * tiles_1d(H, d, m) * tiles_1d(W, d, m); }

}  // namespace wxf
}  // namespace quber
