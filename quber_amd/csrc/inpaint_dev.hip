// inpaint_depth (eval/preprocess_utils.py:44-64: zero mask -> 3x3 dilation -> cv2.inpaint(..., 3, INPAINT_TELEA) -> only the zero
// pixels replaced) ON THE DEVICE, bit-equal to the host restatement in inpaint.hip.
//
// Telea's fast-marching method is sequential in the order of the priority queue - but only among pixels that can influence each
// other.  A pixel that is filled reads flags, T and image values within radius + 1 of itself, the outward pass marches T over the
// known pixels within `radius` of the hole, and nothing else couples two parts of the image.  So:
//   1. the (dilated) hole mask is dilated once more by D = radius + 2 and its connected components are labelled (union-find with
//      atomicMin, one pass): two holes whose gap is <= 2 D end up in ONE component, holes further apart cannot interact;
//   2. every component is marched by ONE WAVE with the host's algorithm and the host's queue order restricted to the component
//      (the queue is ordered by (T, push number); restricting a run to an independent subset keeps the relative order of its
//      pushes, so each component sees exactly the sequence of events it sees in the global run);
//   3. inside a component the march stays sequential (scalar, wave-uniform code: flags, T, both binary heaps live in global memory,
//      L2-resident), but the fill of a pixel - weights and gradient terms of its (2 r + 1)^2 neighbours - is evaluated one neighbour
//      per lane and summed by every lane in the host's raster order (products rounded separately, -ffp-contract=off on both sides),
//      which is what makes the result bit-equal rather than merely close.
// Frames of a batch are stacked (each keeps its one-pixel frame of padding, which never belongs to a mask), so components of
// different frames - and of different holes - run side by side on the chip.
#include <math.h>
#include <stdint.h>

#include "common.h"

namespace quber {
namespace {

enum : uint8_t { KNOWN = 0, BAND = 1, INSIDE = 2, CHANGE = 3 };
constexpr int MAXR = 3;                      // radius of the fill neighbourhood: (2 r + 1)^2 <= 49 lanes
constexpr int MAXCOMP_BLOCKS = 256;          // blocks of the march launch (each loops over components; a block without work exits at once)

struct DNode { float t; uint32_t seq; int i, j; };

struct Tel {
    int B, H, W, R, C, range;                // R = H + 2, C = W + 2 (per frame, padded)
    const uint8_t* depth3;                   // [B][H][W][3]
    uint8_t* out3;                           // [B][H][W][3]
    uint8_t* zero;                           // [B][H][W]   all three channels 0
    uint8_t* m;                              // [B][R][C]   the mask to fill (zero, dilated by the kernel)
    int* parent;                             // [B][R][C]   union-find over the mask dilated by D; -1 outside
    uint8_t* f;                              // [B][R][C]   flags of the inward march
    uint8_t* ring;                           // [B][R][C]   flags of the outward march
    float* T;                                // [B][R][C]
    uint8_t* img;                            // [B][R][C]   the channel being filled, padded coordinates (updated in place as pixels are filled)
    DNode* heap_in;                          // one slot per padded pixel; a component owns a contiguous range
    DNode* heap_out;
    int* comp_root;                          // [ncomp] root pixel (global padded index) of every component
    int* comp_off;                           // [ncomp] start of its heap range
    int* cnt;                                // [B][R][C] pixels per root (at the root's index)
    int* bbox;                               // [B][R][C][4] per root: min i, max i, min j, max j (frame-local)
    int* counters;                           // [0] number of components, [1] heap slots handed out, [2] any hole at all, [3] channels differ
};

__device__ __forceinline__ bool heap_less(const DNode& a, const DNode& b) { return a.t != b.t ? a.t < b.t : a.seq < b.seq; }

// Binary min-heap on (t, seq): the keys are unique, so the order of pops is the total order - whatever the internal layout.
// The march is a chain of dependent accesses (a pop walks ~log2 n levels), and one wave has nothing to hide their latency behind:
// the first HCAP entries - the top levels, which every operation touches - live in LDS (64-cycle accesses), deeper ones in the
// component's range of global memory (an L2 round trip each).  The narrow band of a hole holds about its perimeter: thousands of
// pixels fit the LDS part entirely (16 KB: the march must not crowd the refiner's kernels off the CUs it shares with them).
constexpr int HCAP = 1024;
struct Heap {
    DNode* lds;
    DNode* glob;
    int n;
    __device__ __forceinline__ DNode get(int k) const { return k < HCAP ? lds[k] : glob[k]; }
    __device__ __forceinline__ void set(int k, const DNode& v) { if (k < HCAP) lds[k] = v; else glob[k] = v; }
    __device__ inline void push(DNode v) {
        int k = n++;
        while (k > 0) {
            const int p = (k - 1) >> 1;
            const DNode pv = get(p);
            if (!heap_less(v, pv)) break;
            set(k, pv);
            k = p;
        }
        set(k, v);
    }
    __device__ inline DNode pop() {
        const DNode top = get(0);
        const DNode v = get(--n);
        int k = 0;
        for (;;) {
            int c = 2 * k + 1;
            if (c >= n) break;
            DNode cv = get(c);
            if (c + 1 < n) {
                const DNode rv = get(c + 1);
                if (heap_less(rv, cv)) { cv = rv; ++c; }
            }
            if (!heap_less(cv, v)) break;
            set(k, cv);
            k = c;
        }
        if (n > 0) set(k, v);
        return top;
    }
};

__device__ inline float fm_solve(const float* T, const uint8_t* f, long base, int C, int i1, int j1, int i2, int j2) {
    const double a11 = T[base + (long)i1 * C + j1], a22 = T[base + (long)i2 * C + j2], m12 = a11 < a22 ? a11 : a22;
    const bool k1 = f[base + (long)i1 * C + j1] != INSIDE, k2 = f[base + (long)i2 * C + j2] != INSIDE;
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
__device__ inline float fm_dist(const float* T, const uint8_t* f, long base, int C, int i, int j) {
    return fminf(fminf(fm_solve(T, f, base, C, i - 1, j, i, j - 1), fm_solve(T, f, base, C, i + 1, j, i, j - 1)),
                 fminf(fm_solve(T, f, base, C, i - 1, j, i, j + 1), fm_solve(T, f, base, C, i + 1, j, i, j + 1)));
}

// ---- pass 1: zero pixels, whether the channels differ, the output starts as a copy ----
__global__ void tel_zero_kernel(const Tel a) {
    const long n = (long)a.B * a.H * a.W;
    bool any = false, differ = false;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < n; p += (long)gridDim.x * blockDim.x) {
        const uint8_t x = a.depth3[3 * p], y = a.depth3[3 * p + 1], z = a.depth3[3 * p + 2];
        const bool zr = (x | y | z) == 0;
        a.zero[p] = zr;
        a.out3[3 * p] = x; a.out3[3 * p + 1] = y; a.out3[3 * p + 2] = z;
        any |= zr;
        differ |= x != y || x != z;
    }
    if (__any(any) && (threadIdx.x & 63) == 0) a.counters[2] = 1;
    if (__any(differ) && (threadIdx.x & 63) == 0) a.counters[3] = 1;
}

// ---- pass 2: per padded pixel: the mask (zero dilated by the kernel), the union-find seed over the mask dilated by D, march state ----
__global__ void tel_mask_kernel(const Tel a, int kr, int D) {
    const long n = (long)a.B * a.R * a.C;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < n; p += (long)gridDim.x * blockDim.x) {
        const int j = (int)(p % a.C);
        const long q = p / a.C;
        const int i = (int)(q % a.R), b = (int)(q / a.R);
        const int y = i - 1, x = j - 1;
        bool msk = false, wide = false;
        if (y >= 0 && x >= 0 && y < a.H && x < a.W) {
            const uint8_t* z = a.zero + (long)b * a.H * a.W;
            const int reach = kr + D;
            for (int dy = -reach; dy <= reach; ++dy) {
                const int yy = y + dy;
                if (yy < 0 || yy >= a.H) continue;
                for (int dx = -reach; dx <= reach; ++dx) {
                    const int xx = x + dx;
                    if (xx < 0 || xx >= a.W || !z[(long)yy * a.W + xx]) continue;
                    wide = true;
                    if (dy >= -kr && dy <= kr && dx >= -kr && dx <= kr) msk = true;
                }
            }
        }
        a.m[p] = msk;
        a.parent[p] = wide ? (int)p : -1;
        a.f[p] = KNOWN;
        a.ring[p] = KNOWN;
        a.T[p] = 1.0e6f;
        a.cnt[p] = 0;
        a.bbox[4 * p] = 0x7fffffff; a.bbox[4 * p + 1] = -1; a.bbox[4 * p + 2] = 0x7fffffff; a.bbox[4 * p + 3] = -1;
    }
}

__device__ inline int uf_find(int* parent, int x) {
    int r = x;
    while (true) {
        const int p = parent[r];
        if (p == r) return r;
        r = p;
    }
}
__device__ inline void uf_union(int* parent, int x, int y) {
    for (;;) {
        x = uf_find(parent, x);
        y = uf_find(parent, y);
        if (x == y) return;
        if (x < y) { const int t = x; x = y; y = t; }        // x > y: link x under y
        const int old = atomicMin(&parent[x], y);
        if (old == x) return;
        x = old;                                             // somebody else linked x meanwhile: continue from there
    }
}
// ---- pass 3: union with the four neighbours already visited in raster order (8-connectivity) ----
__global__ void tel_union_kernel(const Tel a) {
    const long n = (long)a.B * a.R * a.C;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < n; p += (long)gridDim.x * blockDim.x) {
        if (a.parent[p] < 0) continue;
        const int j = (int)(p % a.C), i = (int)((p / a.C) % a.R);
        if (j > 0 && a.parent[p - 1] >= 0) uf_union(a.parent, (int)p, (int)p - 1);
        if (i > 0) {
            if (a.parent[p - a.C] >= 0) uf_union(a.parent, (int)p, (int)(p - a.C));
            if (j > 0 && a.parent[p - a.C - 1] >= 0) uf_union(a.parent, (int)p, (int)(p - a.C - 1));
            if (j + 1 < a.C && a.parent[p - a.C + 1] >= 0) uf_union(a.parent, (int)p, (int)(p - a.C + 1));
        }
    }
}
// ---- pass 4: flatten, component sizes and bounding boxes at the roots ----
__global__ void tel_flatten_kernel(const Tel a) {
    const long n = (long)a.B * a.R * a.C;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < n; p += (long)gridDim.x * blockDim.x) {
        if (a.parent[p] < 0) continue;
        const int r = uf_find(a.parent, (int)p);
        a.parent[p] = r;                                     // (roots never change any more: every union is done)
        const int j = (int)(p % a.C), i = (int)((p / a.C) % a.R);
        atomicAdd(&a.cnt[r], 1);
        atomicMin(&a.bbox[4 * (long)r], i); atomicMax(&a.bbox[4 * (long)r + 1], i);
        atomicMin(&a.bbox[4 * (long)r + 2], j); atomicMax(&a.bbox[4 * (long)r + 3], j);
    }
}
// ---- pass 5: the list of components that contain something to fill, each with a range of heap slots ----
__global__ void tel_list_kernel(const Tel a) {
    const long n = (long)a.B * a.R * a.C;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < n; p += (long)gridDim.x * blockDim.x) {
        if (a.parent[p] != (int)p) continue;
        const int k = atomicAdd(&a.counters[0], 1);
        a.comp_root[k] = (int)p;
        a.comp_off[k] = atomicAdd(&a.counters[1], a.cnt[p]);
    }
}

// ---- pass 6: one wave per component: outward march, inward march with the fills ----
// The march is a chain of dependent accesses with one wave to run it: every global access is an L2 round trip (~0.6 us).  A component
// whose bounding box (+ 1: the image frame) holds at most WCAP pixels therefore keeps its T / flag / image window in LDS (other
// components' pixels inside that rectangle are never read: anything within reach of this component's hole would be part of it);
// larger ones march in global memory.
constexpr int WCAP = 9216;                       // window pixels (T 4 B + two flag bytes + image byte = 7 B each: 63 KB)
constexpr int WIN_BYTES = WCAP * 7;
__global__ __launch_bounds__(64) void tel_march_kernel(const Tel a, int ch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    __shared__ float4 sv[64];                    // per neighbour: (w I, w gix rx, w giy ry, w); zeros for the neighbours that do not count
    __shared__ DNode hl[HCAP];                   // the top of the active heap (16 KB)
    const int lane = threadIdx.x;
    if (ch > 0 && !a.counters[3]) return;        // the three channels are alike (what normalize_depth produces): channel 0's fill serves all
    const int R = a.R, C = a.C, range = a.range;
    for (int comp = blockIdx.x; comp < a.counters[0]; comp += gridDim.x) {
        const int root = a.comp_root[comp];
        const int b = root / (R * C);
        const long base = (long)b * R * C;                   // this frame's padded grid
        const int i0 = a.bbox[4 * (long)root], i1 = a.bbox[4 * (long)root + 1], j0 = a.bbox[4 * (long)root + 2], j1 = a.bbox[4 * (long)root + 3];
        DNode* const hin = a.heap_in + a.comp_off[comp];
        DNode* const hout = a.heap_out + a.comp_off[comp];
        const uint8_t* const M = a.m + base;
        const int* const lab = a.parent + base;
        uint8_t* const gimg = a.img + base;                  // padded coordinates, like everything else here
        auto mine = [&](int i, int j) { return lab[(long)i * C + j] == root; };
        // window: the bounding box and one pixel around it (the frame of the image where the box touches it)
        const int wi0 = max(i0 - 1, 0), wi1 = min(i1 + 1, R - 1), wj0 = max(j0 - 1, 0), wj1 = min(j1 + 1, C - 1);
        const int wh = wi1 - wi0 + 1, ww = wj1 - wj0 + 1;
        const bool win = wh * ww <= WCAP;
        float* T; uint8_t *F, *RG, *img;
        int ld, oi, oj;
        __syncthreads();
        if (win) {
            T = reinterpret_cast<float*>(dyn); F = dyn + 4 * WCAP; RG = F + WCAP; img = RG + WCAP;
            ld = ww; oi = wi0; oj = wj0;
            for (int p = lane; p < wh * ww; p += 64) {
                T[p] = 1.0e6f; F[p] = KNOWN; RG[p] = KNOWN;
                img[p] = gimg[(long)(wi0 + p / ww) * C + wj0 + p % ww];
            }
        } else {
            T = a.T + base; F = a.f + base; RG = a.ring + base; img = gimg;
            ld = C; oi = 0; oj = 0;
        }
        __syncthreads();
        auto X = [&](int i, int j) -> long { return (long)(i - oi) * ld + (j - oj); };
        // ---- init over the bounding box, in raster order: INSIDE / BAND, both heaps start as the band in push order ----
        int nin = 0;
        uint32_t seq = 0;
        for (int i = i0; i <= i1; ++i)
            for (int jb = j0; jb <= j1; jb += 64) {
                const int j = jb + lane;
                bool band = false;
                if (j <= j1 && i >= 1 && j >= 1 && i < R - 1 && j < C - 1 && mine(i, j)) {
                    if (M[(long)i * C + j]) {
                        F[X(i, j)] = INSIDE;
                    } else if (M[(long)(i - 1) * C + j] || M[(long)(i + 1) * C + j] || M[(long)i * C + j - 1] || M[(long)i * C + j + 1]) {
                        band = true;
                        F[X(i, j)] = BAND;
                        T[X(i, j)] = 0.f;
                    }
                }
                const unsigned long long bal = __ballot(band);
                if (band) {
                    const int k = nin + __popcll(bal & ((1ull << lane) - 1ull));
                    const DNode nd{0.f, seq + (uint32_t)(k - nin), i, j};
                    hin[k] = nd;                              // equal keys t = 0, increasing push number: appended in order, already a heap
                    hout[k] = nd;
                }
                const int c = __popcll(bal);
                nin += c;
                seq += (uint32_t)c;
            }
        const int nband = nin;
        __threadfence_block();
        __syncthreads();
        // ---- the ring: known pixels within the (2 range + 1)^2 box of a mask pixel, not BAND ----
        for (int i = i0; i <= i1; ++i)
            for (int jb = j0; jb <= j1; jb += 64) {
                const int j = jb + lane;
                if (j > j1 || i < 1 || j < 1 || i >= R - 1 || j >= C - 1 || !mine(i, j)) continue;
                if (M[(long)i * C + j] || F[X(i, j)] == BAND) continue;
                bool near = false;
                for (int k = max(1, i - range); k <= min(R - 2, i + range) && !near; ++k)
                    for (int l = max(1, j - range); l <= min(C - 2, j + range); ++l)
                        if (M[(long)k * C + l]) { near = true; break; }
                if (near) RG[X(i, j)] = INSIDE;
            }
        __threadfence_block();
        __syncthreads();
        // ---- outward march (wave-uniform scalar code: every lane computes the same) ----
        for (int k = lane; k < min(nband, HCAP); k += 64) hl[k] = hout[k];
        __syncthreads();
        Heap ho{hl, hout, nband};
        while (ho.n > 0) {
            const DNode nd = ho.pop();
            RG[X(nd.i, nd.j)] = CHANGE;
            // the four neighbours' flags first (independent loads: one round trip instead of four)
            uint8_t nf[4];
            for (int q = 0; q < 4; ++q) {
                const int i = nd.i + (q == 0 ? -1 : q == 2 ? 1 : 0), j = nd.j + (q == 1 ? -1 : q == 3 ? 1 : 0);
                nf[q] = (i <= 0 || j <= 0 || i >= R - 1 || j >= C - 1) ? (uint8_t)KNOWN : RG[X(i, j)];     // the frame itself is never marched
            }
            for (int q = 0; q < 4; ++q) {
                if (nf[q] != INSIDE) continue;
                const int i = nd.i + (q == 0 ? -1 : q == 2 ? 1 : 0), j = nd.j + (q == 1 ? -1 : q == 3 ? 1 : 0);
                const float d = fm_dist(T, RG, -((long)oi * ld + oj), ld, i, j);
                T[X(i, j)] = d;
                RG[X(i, j)] = BAND;
                ho.push(DNode{d, seq++, i, j});
            }
        }
        __threadfence_block();
        __syncthreads();
        for (int i = i0; i <= i1; ++i)
            for (int jb = j0; jb <= j1; jb += 64) {
                const int j = jb + lane;
                if (j <= j1 && mine(i, j) && RG[X(i, j)] == CHANGE && F[X(i, j)] != BAND) T[X(i, j)] = -T[X(i, j)];
            }
        __threadfence_block();
        __syncthreads();
        // ---- inward march with the fills ----
        auto I = [&](int y, int x) -> float { return (float)img[X(y + 1, x + 1)]; };      // unpadded image coordinates, as the host code
        for (int k = lane; k < min(nband, HCAP); k += 64) hl[k] = hin[k];
        __syncthreads();
        Heap hi{hl, hin, nband};
        const long xo = -((long)oi * ld + oj);
        while (hi.n > 0) {
            const DNode nd = hi.pop();
            F[X(nd.i, nd.j)] = KNOWN;
            for (int q = 0; q < 4; ++q) {
                // (the flag is read per neighbour, after the previous neighbour's fill: a fill turns its pixel BAND and may be this one)
                const int i = nd.i + (q == 0 ? -1 : q == 2 ? 1 : 0), j = nd.j + (q == 1 ? -1 : q == 3 ? 1 : 0);
                if (i <= 0 || j <= 0 || i >= R - 1 || j >= C - 1) continue;
                if (F[X(i, j)] != INSIDE) continue;
                const float dist = fm_dist(T, F, xo, ld, i, j);
                T[X(i, j)] = dist;
                __threadfence_block();
                float gtx, gty;
                if (F[X(i, j + 1)] != INSIDE) gtx = F[X(i, j - 1)] != INSIDE ? (T[X(i, j + 1)] - T[X(i, j - 1)]) * 0.5f : T[X(i, j + 1)] - T[X(i, j)];
                else gtx = F[X(i, j - 1)] != INSIDE ? T[X(i, j)] - T[X(i, j - 1)] : 0.f;
                if (F[X(i + 1, j)] != INSIDE) gty = F[X(i - 1, j)] != INSIDE ? (T[X(i + 1, j)] - T[X(i - 1, j)]) * 0.5f : T[X(i + 1, j)] - T[X(i, j)];
                else gty = F[X(i - 1, j)] != INSIDE ? T[X(i, j)] - T[X(i - 1, j)] : 0.f;
                // one neighbour per lane: lane n is (k, l) = (i - range + n / side, j - range + n % side)
                const int side = 2 * range + 1;
                float4 v = {0.f, 0.f, 0.f, 0.f};
                if (lane < side * side) {
                    const int k = i - range + lane / side, l = j - range + lane % side;
                    if (k > 0 && l > 0 && k < R - 1 && l < C - 1 && F[X(k, l)] != INSIDE && (l - j) * (l - j) + (k - i) * (k - i) <= range * range) {
                        const int km = k - 1 + (k == 1), kp = k - 1 - (k == R - 2);
                        const int lm = l - 1 + (l == 1), lp = l - 1 - (l == C - 2);
                        const float ry = (float)(i - k), rx = (float)(j - l);
                        const float len2 = rx * rx + ry * ry;
                        const float dst = (float)(1. / (len2 * sqrt((double)len2)));
                        const float lev = (float)(1. / (1 + fabs(T[X(k, l)] - T[X(i, j)])));
                        float dir = rx * gtx + ry * gty;
                        if (fabsf(dir) <= 0.01f) dir = 0.000001f;
                        const float w = fabsf(dst * lev * dir);
                        float gix, giy;
                        if (F[X(k, l + 1)] != INSIDE) gix = F[X(k, l - 1)] != INSIDE ? (I(km, lp + 1) - I(km, lm - 1)) * 2.0f : I(km, lp + 1) - I(km, lm);
                        else gix = F[X(k, l - 1)] != INSIDE ? I(km, lp) - I(km, lm - 1) : 0.f;
                        if (F[X(k + 1, l)] != INSIDE) giy = F[X(k - 1, l)] != INSIDE ? (I(kp + 1, lm) - I(km - 1, lm)) * 2.0f : I(kp + 1, lm) - I(km, lm);
                        else giy = F[X(k - 1, l)] != INSIDE ? I(kp, lm) - I(km - 1, lm) : 0.f;
                        v = float4{w * I(km, lm), w * gix * rx, w * giy * ry, w};
                    }
                }
                sv[lane] = v;
                __syncthreads();                             // (one wave per block)
                // ... summed by every lane in the host's order (k outer, l inner).  The neighbours that do not count contribute exact
                // zeros: x + 0 and x - 0 return x bit for bit (the sums never are -0: Ia, s >= +0; Jx, Jy start at +0 and 0 - 0 = +0)
                float Ia = 0.f, Jx = 0.f, Jy = 0.f, sm = 1.0e-20f;
#pragma unroll 7
                for (int n = 0; n < side * side; ++n) {
                    const float4 e = sv[n];
                    Ia += e.x;
                    Jx -= e.y;
                    Jy -= e.z;
                    sm += e.w;
                }
                __syncthreads();
                const float sat = Ia / sm + (Jx + Jy) / (sqrtf(Jx * Jx + Jy * Jy) + 1.0e-20f) + 0.5f;
                img[X(i, j)] = (uint8_t)fmaxf(0.f, fminf(255.f, floorf(sat)));
                F[X(i, j)] = BAND;
                hi.push(DNode{dist, seq++, i, j});
                __threadfence_block();
            }
        }
        __threadfence_block();
        __syncthreads();
        if (win) {                                            // the filled pixels back to the channel image
            for (int i = i0; i <= i1; ++i)
                for (int jb = j0; jb <= j1; jb += 64) {
                    const int j = jb + lane;
                    if (j <= j1 && i >= 1 && j >= 1 && i < R - 1 && j < C - 1 && mine(i, j) && M[(long)i * C + j]) gimg[(long)i * C + j] = img[X(i, j)];
                }
        }
        __threadfence_block();
    }
}

// channel `ch` of depth3 -> img (all frames; padded coordinates, the frame itself is never read)
__global__ void tel_take_channel_kernel(const Tel a, int ch) {
    if (ch > 0 && !a.counters[3]) return;
    const long n = (long)a.B * a.H * a.W;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < n; p += (long)gridDim.x * blockDim.x) {
        const int x = (int)(p % a.W);
        const long q = p / a.W;
        const int y = (int)(q % a.H), b = (int)(q / a.H);
        a.img[((long)b * a.R + y + 1) * a.C + x + 1] = a.depth3[3 * p + ch];
    }
}
// np.where(depth == 0, filled, depth), element-wise: when the three channels are alike they share channel 0's fill
__global__ void tel_merge_kernel(const Tel a, int ch) {
    const long n = (long)a.B * a.H * a.W;
    const bool all = !a.counters[3];
    if (all && ch > 0) return;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < n; p += (long)gridDim.x * blockDim.x) {
        const int x = (int)(p % a.W);
        const long q = p / a.W;
        const int y = (int)(q % a.H), b = (int)(q / a.H);
        const uint8_t v = a.img[((long)b * a.R + y + 1) * a.C + x + 1];
        if (all) {
            if (a.depth3[3 * p] == 0) a.out3[3 * p] = a.out3[3 * p + 1] = a.out3[3 * p + 2] = v;
        } else if (a.depth3[3 * p + ch] == 0) {
            a.out3[3 * p + ch] = v;
        }
    }
}
// march-state reset between channels (different images, same mask)
__global__ void tel_reset_kernel(const Tel a) {
    if (!a.counters[3]) return;
    const long n = (long)a.B * a.R * a.C;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < n; p += (long)gridDim.x * blockDim.x) {
        a.f[p] = KNOWN; a.ring[p] = KNOWN; a.T[p] = 1.0e6f;
    }
}

inline size_t al(size_t n) { return (n + 255) & ~(size_t)255; }

}  // namespace

size_t inpaint_depth_ws_bytes(int B, int H, int W) {
    const size_t px = (size_t)B * H * W, pp = (size_t)B * (H + 2) * (W + 2);
    return al(px) + al(pp) * 4 + al(pp * 4) * 3 + al(pp * 16) + al(pp * sizeof(DNode)) * 2 + al(pp * 4) * 2 + al(64);
}

// depth3 / out3: device u8 [B][H][W][3]
int launch_inpaint_depth(const uint8_t* depth3, int B, int H, int W, int kernel, void* ws, size_t ws_bytes, uint8_t* out3, hipStream_t st) {
    if (!depth3 || !out3 || !ws || B < 1 || H < 1 || W < 1 || kernel < 1 || !(kernel & 1)) return fail("inpaint_depth (device): bad argument");
    if (kernel > MAXR) return fail("inpaint_depth (device): kernel size up to 3 (the reference's call: eval/refiner_model.py:255)");
    if (ws_bytes < inpaint_depth_ws_bytes(B, H, W)) return fail("inpaint_depth (device): workspace too small");
    if ((long)B * (H + 2) * (W + 2) >= (1L << 31)) return fail("inpaint_depth (device): batch too large for 32-bit pixel indices");
    Tel a{};
    a.B = B; a.H = H; a.W = W; a.R = H + 2; a.C = W + 2; a.range = std::max(1, std::min(100, kernel));
    a.depth3 = depth3; a.out3 = out3;
    const size_t px = (size_t)B * H * W, pp = (size_t)B * a.R * a.C;
    char* w = reinterpret_cast<char*>(ws);
    a.zero = reinterpret_cast<uint8_t*>(w); w += al(px);
    a.img = reinterpret_cast<uint8_t*>(w); w += al(pp);
    a.m = reinterpret_cast<uint8_t*>(w); w += al(pp);
    a.f = reinterpret_cast<uint8_t*>(w); w += al(pp);
    a.ring = reinterpret_cast<uint8_t*>(w); w += al(pp);
    a.parent = reinterpret_cast<int*>(w); w += al(pp * 4);
    a.T = reinterpret_cast<float*>(w); w += al(pp * 4);
    a.cnt = reinterpret_cast<int*>(w); w += al(pp * 4);
    a.bbox = reinterpret_cast<int*>(w); w += al(pp * 16);
    a.heap_in = reinterpret_cast<DNode*>(w); w += al(pp * sizeof(DNode));
    a.heap_out = reinterpret_cast<DNode*>(w); w += al(pp * sizeof(DNode));
    a.comp_root = reinterpret_cast<int*>(w); w += al(pp * 4);
    a.comp_off = reinterpret_cast<int*>(w); w += al(pp * 4);
    a.counters = reinterpret_cast<int*>(w);
    static const hipError_t lds_ok = hipFuncSetAttribute((const void*)tel_march_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WIN_BYTES);
    QB_CHECK(lds_ok);
    if (int rc = launch_zero(a.counters, 64, st)) return rc;
    const int kr = kernel / 2, D = a.range + 2;
    const unsigned gp = (unsigned)std::min<size_t>((px + 255) / 256, 4096), gq = (unsigned)std::min<size_t>((pp + 255) / 256, 4096);
    hipLaunchKernelGGL(tel_zero_kernel, dim3(gp), dim3(256), 0, st, a);
    hipLaunchKernelGGL(tel_mask_kernel, dim3(gq), dim3(256), 0, st, a, kr, D);
    hipLaunchKernelGGL(tel_union_kernel, dim3(gq), dim3(256), 0, st, a);
    hipLaunchKernelGGL(tel_flatten_kernel, dim3(gq), dim3(256), 0, st, a);
    hipLaunchKernelGGL(tel_list_kernel, dim3(gq), dim3(256), 0, st, a);
    // channel 0 always; channels 1, 2 only do work when the channels differ (device-side flag; otherwise the merge of channel 0
    // writes all three).  No host synchronisation anywhere: the launches are sized for the worst case.
    for (int ch = 0; ch < 3; ++ch) {
        if (ch > 0) hipLaunchKernelGGL(tel_reset_kernel, dim3(gq), dim3(256), 0, st, a);
        hipLaunchKernelGGL(tel_take_channel_kernel, dim3(gp), dim3(256), 0, st, a, ch);
        hipLaunchKernelGGL(tel_march_kernel, dim3(MAXCOMP_BLOCKS), dim3(64), WIN_BYTES, st, a, ch);
        hipLaunchKernelGGL(tel_merge_kernel, dim3(gp), dim3(256), 0, st, a, ch);
    }
    QB_CHECK(hipGetLastError());
    return 0;
}

}  // namespace quber
