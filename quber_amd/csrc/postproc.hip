// Post-processing: head logits -> panoptic label map + per-instance score / box / mask, without
// host synchronisation (fixed capacity `cap` = TOP_K_INSTANCE = 200 slots per frame).
// Replaces
//   maskrefiner/modeling/mask_refiner/post_processing.py:9-41    find_instance_center
//   maskrefiner/modeling/mask_refiner/post_processing.py:44-76   group_pixels
//   maskrefiner/modeling/mask_refiner/post_processing.py:110-162 merge_semantic_and_instance
//   maskrefiner/modeling/mask_refiner/model.py:291-356           sigmoid().round(), instance extraction
// Bit-exactness notes (all probed against torch CPU, see DECISIONS.md section 2):
//   * sigmoid(x).round() == 1  <=>  x > 1.5 * 2^-24 in correctly rounded fp32 arithmetic,
//   * torch.norm over the (dy,dx) pair evaluates sqrt(fma(dx, dx, dy*dy)); argmin keeps the first minimum,
//   * top-k keeps values STRICTLY greater than max(k-th largest, 0): at most k-1 centres, raster order,
//   * instances smaller than 512 px are dropped before the running relabel 1001, 1002, ...
#include "common.h"

namespace quber {

constexpr int NT = 32;  // NMS tile

// ---- P1: threshold + (2r+1)^2 max-pool NMS -> candidate map (value or -1) ----
// The window maximum is separable (rows, then columns: 2 (2r + 1) LDS reads per pixel instead of (2r + 1)^2; a maximum does not depend
// on the order it is taken in).  Surviving candidates are also appended - value and pixel index - to a short per-frame list
// (LCAND slots): a frame has a few dozen of them, and P2 then never walks the 300 000-pixel map; a frame with more candidates than
// slots (plateaus: every pixel of a constant region is its window's maximum) keeps the map path.
constexpr int LCAND = 2048;
struct CandList { unsigned n; unsigned pad; float val[LCAND]; int idx[LCAND]; };

__global__ __launch_bounds__(256) void nms_kernel(const float* __restrict__ logits, int nch, int H, int W, float thr,
                                                  int r, float* __restrict__ cand, CandList* __restrict__ lists) {
    extern __shared__ float sv[];  // (NT+2r)^2 thresholded values, then (NT+2r) x NT row maxima
    const int b = blockIdx.z;
    const int P = NT + 2 * r;
    float* const rowmax = sv + P * P;
    const float* c = logits + ((long)b * nch + 1) * H * W;
    const int ty0 = blockIdx.y * NT, tx0 = blockIdx.x * NT;
    for (int i = threadIdx.x; i < P * P; i += 256) {
        const int ly = i / P, lx = i - ly * P;
        const int gy = ty0 + ly - r, gx = tx0 + lx - r;
        float v = -INFINITY;
        if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
            v = c[(long)gy * W + gx];
            v = v > thr ? v : -1.f;
        }
        sv[i] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < P * NT; i += 256) {
        const int ly = i / NT, lx = i - ly * NT;
        float m = -INFINITY;
        for (int dx = 0; dx <= 2 * r; ++dx) m = fmaxf(m, sv[ly * P + lx + dx]);
        rowmax[i] = m;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NT * NT; i += 256) {
        const int ly = i / NT, lx = i - ly * NT;
        const int gy = ty0 + ly, gx = tx0 + lx;
        if (gy >= H || gx >= W) continue;
        const float v = sv[(ly + r) * P + lx + r];
        float m = -INFINITY;
        for (int dy = 0; dy <= 2 * r; ++dy) m = fmaxf(m, rowmax[(ly + dy) * NT + lx]);
        const bool keep = v == m;
        cand[(long)b * H * W + (long)gy * W + gx] = keep ? v : -1.f;
        if (keep && v > 0.f) {                     // (P2 counts strictly positive candidates only)
            const unsigned slot = atomicAdd(&lists[b].n, 1u);
            if (slot < (unsigned)LCAND) { lists[b].val[slot] = v; lists[b].idx[slot] = gy * W + gx; }
        }
    }
}

// ---- P2: exact k-th largest (radix select on the float bits) + raster-order compaction; one block per frame ----
__global__ __launch_bounds__(1024) void select_kernel(const float* __restrict__ cand, int HW, int W, int top_k, int cap,
                                                      int* __restrict__ centers, int* __restrict__ ncenters,
                                                      const CandList* __restrict__ lists) {
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_mask, s_remaining, s_npos;
    __shared__ unsigned scan[1024];
    const int b = blockIdx.x, t = threadIdx.x;
    const float* c = cand + (long)b * HW;
    // ---- the short list (P1): the same selection on a few dozen (value, pixel) pairs instead of the whole map ----
    const unsigned nl = lists[b].n;
    if (nl <= (unsigned)LCAND) {
        __shared__ float lv[LCAND];
        __shared__ int li[LCAND];
        for (unsigned i = t; i < nl; i += 1024) { lv[i] = lists[b].val[i]; li[i] = lists[b].idx[i]; }
        __syncthreads();
        float cut = 0.f;                           // max(k-th largest, 0); fewer than k positives: 0
        if ((int)nl >= top_k) {
            if (t == 0) { s_prefix = 0; s_mask = 0; s_remaining = (unsigned)top_k; }
            __syncthreads();
            for (int shift = 24; shift >= 0; shift -= 8) {
                if (t < 256) hist[t] = 0;
                __syncthreads();
                const unsigned prefix = s_prefix, mask = s_mask;
                for (unsigned i = t; i < nl; i += 1024) {
                    const unsigned u = __float_as_uint(lv[i]);
                    if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u);
                }
                __syncthreads();
                if (t == 0) {
                    unsigned rem = s_remaining;
                    int d = 255;
                    for (; d > 0; --d) {
                        if (hist[d] >= rem) break;
                        rem -= hist[d];
                    }
                    s_prefix = prefix | ((unsigned)d << shift);
                    s_mask = mask | (255u << shift);
                    s_remaining = rem;
                }
                __syncthreads();
            }
            cut = __uint_as_float(s_prefix);
        }
        // survivors (strictly above the cut: at most top_k - 1, or all nl < top_k) in raster order: a survivor's slot is the number
        // of survivors with a smaller pixel index
        if (t == 0) s_npos = 0;
        __syncthreads();
        for (unsigned i = t; i < nl; i += 1024) {
            if (!(lv[i] > cut)) continue;
            const int me = li[i];
            unsigned at = 0;
            for (unsigned j = 0; j < nl; ++j) at += (lv[j] > cut && li[j] < me) ? 1u : 0u;
            atomicAdd(&s_npos, 1u);
            if ((int)at < cap) {
                centers[((long)b * cap + at) * 2] = me / W;
                centers[((long)b * cap + at) * 2 + 1] = me % W;
            }
        }
        __syncthreads();
        if (t == 0) ncenters[b] = min((int)s_npos, cap);
        return;
    }
    // count strictly positive candidates
    unsigned local = 0;
    if ((HW & 3) == 0) {
        for (int i = 4 * t; i < HW; i += 4096) {
            const float4 v = *reinterpret_cast<const float4*>(c + i);
            local += (v.x > 0.f) + (v.y > 0.f) + (v.z > 0.f) + (v.w > 0.f);
        }
    } else {
        for (int i = t; i < HW; i += 1024) local += c[i] > 0.f;
    }
    if (t == 0) s_npos = 0;
    __syncthreads();
    atomicAdd(&s_npos, local);
    __syncthreads();
    float cut = 0.f;  // max(k-th largest, 0); with fewer than k positives the k-th value is -1 -> 0
    if ((int)s_npos >= top_k) {
        if (t == 0) { s_prefix = 0; s_mask = 0; s_remaining = (unsigned)top_k; }
        __syncthreads();
        for (int shift = 24; shift >= 0; shift -= 8) {
            if (t < 256) hist[t] = 0;
            __syncthreads();
            const unsigned prefix = s_prefix, mask = s_mask;
            for (int i = t; i < HW; i += 1024) {
                const float v = c[i];
                if (v > 0.f) {
                    const unsigned u = __float_as_uint(v);
                    if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u);
                }
            }
            __syncthreads();
            if (t == 0) {
                unsigned rem = s_remaining;
                int d = 255;
                for (; d > 0; --d) {
                    if (hist[d] >= rem) break;
                    rem -= hist[d];
                }
                s_prefix = prefix | ((unsigned)d << shift);
                s_mask = mask | (255u << shift);
                s_remaining = rem;
            }
            __syncthreads();
        }
        cut = __uint_as_float(s_prefix);
    }
    // ordered compaction: wave v owns the contiguous pixel run [v*seg, (v+1)*seg), walked 64 x 4 pixels at a time (lane l
    // holds 4 consecutive pixels, so lane order is raster order); survivors are at most top_k - 1, so the second walk
    // skips almost every step on its ballot
    const int wave = t >> 6, lane = t & 63;
    const int seg = ((HW + 15) / 16 + 255) / 256 * 256;          // multiple of the 256-pixel step
    const int p0 = wave * seg, p1 = min(HW, p0 + seg);
    auto flags = [&](int base) {                                    // bit e: pixel base + 4*lane + e survives
        const int i = base + 4 * lane;
        unsigned f = 0;
        if (i + 3 < p1 && (HW & 3) == 0) {
            const float4 v = *reinterpret_cast<const float4*>(c + i);
            f = (v.x > cut ? 1u : 0u) | (v.y > cut ? 2u : 0u) | (v.z > cut ? 4u : 0u) | (v.w > cut ? 8u : 0u);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (i + e < p1 && c[i + e] > cut) f |= 1u << e;
        }
        return f;
    };
    unsigned n = 0;
    for (int base = p0; base < p1; base += 256) n += __popc(flags(base));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
    if (lane == 0) scan[wave] = n;
    __syncthreads();
    unsigned pos = 0, total = 0;
    for (int v = 0; v < 16; ++v) {
        if (v < wave) pos += scan[v];
        total += scan[v];
    }
    if (n) {
        for (int base = p0; base < p1; base += 256) {
            const unsigned f = flags(base);
            if (__ballot(f != 0) == 0) continue;
            const unsigned cnt = __popc(f);
            unsigned incl = cnt;                                    // inclusive scan of the per-lane counts
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned up = __shfl_up(incl, o);
                if (lane >= o) incl += up;
            }
            unsigned at = pos + incl - cnt;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (f & (1u << e)) {
                    const int i = base + 4 * lane + e;
                    if ((int)at < cap) {
                        centers[((long)b * cap + at) * 2] = i / W;
                        centers[((long)b * cap + at) * 2 + 1] = i % W;
                    }
                    ++at;
                }
            pos += __shfl(incl, 63);
        }
    }
    if (t == 0) ncenters[b] = min((int)total, cap);
}

// ---- P3: nearest-centre grouping + instance areas ----
// idmap: 0 = background, 1..K = instance id on foreground, 255 = foreground with no centre at all
// A block owns GP_PIX consecutive pixels (8 per thread, their three planes requested up front): a block per 256 pixels spent its
// life on the prologue - K centres into LDS, a barrier, 256 area counters to flush - 32 768 times per 8-frame 1024x1024 step.
constexpr int GP_PIX = 2048;

__global__ __launch_bounds__(256) void group_kernel(const float* __restrict__ logits, int nch, int H, int W, int cap,
                                                    const int* __restrict__ centers, const int* __restrict__ ncenters,
                                                    uint8_t* __restrict__ idmap, unsigned* __restrict__ area) {
    __shared__ float cy[256], cx[256];
    __shared__ unsigned harea[256];
    const int b = blockIdx.y;
    const int K = ncenters[b];
    harea[threadIdx.x] = 0;
    if (threadIdx.x < K) {
        cy[threadIdx.x] = (float)centers[((long)b * cap + threadIdx.x) * 2];
        cx[threadIdx.x] = (float)centers[((long)b * cap + threadIdx.x) * 2 + 1];
    }
    const long HW = (long)H * W;
    const float* base = logits + (long)b * nch * HW;
    const long p0 = (long)blockIdx.x * GP_PIX + threadIdx.x;
    constexpr int NP = GP_PIX / 256;
    float l0[NP], oy[NP], ox[NP];
#pragma unroll
    for (int it = 0; it < NP; ++it) {
        const long p = p0 + it * 256;
        const bool in = p < HW;
        l0[it] = in ? base[p] : -1.f;
        oy[it] = in ? base[2 * HW + p] : 0.f;
        ox[it] = in ? base[3 * HW + p] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NP; ++it) {
        const long p = p0 + it * 256;
        unsigned id = 0;
        const bool fg = p < HW && l0[it] > 0x1.8p-24f;  // sigmoid(x).round() == 1
        if (fg) {
            if (K == 0) {
                id = 255;
            } else {
                const int y = (int)(p / W), x = (int)(p - (long)y * W);
                const float ly = __fadd_rn((float)y, oy[it]);
                const float lx = __fadd_rn((float)x, ox[it]);
                // argmin over k of sqrt(dx^2 + dy^2), first index on ties (torch.norm + argmin).  The square root is monotone, so a
                // smaller d^2 can only tie - never lose - after it; the root (a quarter-rate instruction) is taken just for the
                // pairs whose squares are within a few ulp of each other, where two different squares may round to one root
                float best2 = INFINITY;
                for (int k = 0; k < K; ++k) {
                    const float dy = __fsub_rn(cy[k], ly), dx = __fsub_rn(cx[k], lx);
                    const float d2 = __fmaf_rn(dx, dx, __fmul_rn(dy, dy));
                    if (d2 < best2) {
                        // (the root through fp64, correctly rounded to fp32: HIP's __fsqrt_rn is the NATIVE square-root instruction - 1 ulp -
                        // unless OCML_BASIC_ROUNDED_OPERATIONS is defined; with it the grouping of near-equidistant pixels left the
                        // reference's, tests/test_gpu_parity.py::test_group_near_ties_follow_the_rounded_norm)
                        if (d2 > best2 * 0.999999f && !((float)sqrt((double)d2) < (float)sqrt((double)best2))) continue;    // the roots tie: the earlier centre stays
                        best2 = d2;
                        id = k + 1;
                    }
                }
                if (id == 0) id = 1;  // all-NaN distances: argmin returns index 0
            }
        }
        if (p < HW) idmap[(long)b * HW + p] = (uint8_t)id;
        // areas: one LDS atomic per wave when its foreground lanes agree on the instance (inside an object: nearly always)
        const unsigned long long m = __ballot(fg);
        if (m) {
            const unsigned first = __shfl(id, __ffsll((long long)m) - 1);
            if (__all(!fg || id == first)) {
                if ((threadIdx.x & 63) == 0) atomicAdd(&harea[first], (unsigned)__popcll(m));
            } else if (fg) {
                atomicAdd(&harea[id], 1u);
            }
        }
    }
    __syncthreads();
    if (harea[threadIdx.x]) atomicAdd(&area[(long)b * 256 + threadIdx.x], harea[threadIdx.x]);
}

struct InstStat {
    double prob;
    unsigned long long sy, sx;
    unsigned cnt;
    int xmin, ymin, xmax, ymax;
    int pad;
};

// ---- P4: 512-px filter + running relabel; one block per frame ----
__global__ void relabel_kernel(const unsigned* __restrict__ area, const int* __restrict__ ncenters, int min_area,
                               int stuff_area, int label_divisor, int cap, float* __restrict__ lut,
                               int* __restrict__ count, float* __restrict__ labels, InstStat* __restrict__ stats) {
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < 256; i += blockDim.x) lut[(long)b * 256 + i] = -1.f;
    for (int i = threadIdx.x; i < cap; i += blockDim.x) {
        labels[(long)b * cap + i] = -1.f;
        InstStat s;
        s.prob = 0.0; s.sy = 0; s.sx = 0; s.cnt = 0;
        s.xmin = 1 << 30; s.ymin = 1 << 30; s.xmax = -1; s.ymax = -1; s.pad = 0;
        stats[(long)b * cap + i] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int K = ncenters[b];
        int n = 0;
        if (K == 0) {
            // class 1 is 'stuff' for thing_ids = [0]: a centre-less foreground blob becomes label 1*divisor
            if ((int)area[(long)b * 256 + 255] >= stuff_area) {
                lut[(long)b * 256 + 255] = (float)label_divisor;
                labels[(long)b * cap] = (float)label_divisor;
                n = 1;
            }
        } else {
            for (int k = 1; k <= K; ++k) {
                if ((int)area[(long)b * 256 + k] < min_area) continue;
                ++n;
                lut[(long)b * 256 + k] = (float)(label_divisor + n);
                labels[(long)b * cap + n - 1] = (float)(label_divisor + n);
            }
        }
        count[b] = n;
    }
}

// ---- P5: write the panoptic map and accumulate per-instance statistics ----
// Each block owns PS_PIX consecutive pixels.  Statistics are first combined per wave (shuffles, when the 64
// lanes agree on the instance - the common case inside an object), then per block in LDS, and only then
// flushed with one global atomic per (block, instance, field): a frame costs a few thousand global atomics
// instead of eight per foreground pixel.
constexpr int PS_PIX = 4096;

__global__ __launch_bounds__(256) void paint_stats_kernel(const float* __restrict__ logits, int nch, int H, int W,
                                                          int cap, int label_divisor, const uint8_t* __restrict__ idmap,
                                                          const float* __restrict__ lut, float* __restrict__ pan,
                                                          InstStat* __restrict__ stats) {
    __shared__ float slut[256];
    __shared__ double s_prob[256];
    __shared__ unsigned s_cnt[256], s_sy[256], s_sx[256];
    __shared__ int s_x0[256], s_y0[256], s_x1[256], s_y1[256];
    const int b = blockIdx.y, t = threadIdx.x;
    slut[t] = lut[(long)b * 256 + t];
    s_prob[t] = 0.0; s_cnt[t] = 0; s_sy[t] = 0; s_sx[t] = 0;
    s_x0[t] = 1 << 30; s_y0[t] = 1 << 30; s_x1[t] = -1; s_y1[t] = -1;
    __syncthreads();
    const long HW = (long)H * W;
    const long base = (long)blockIdx.x * PS_PIX;
    // the block's instance ids and foreground logits are requested up front (16 + 16 loads in flight per thread): 2048 blocks
    // walking 16 dependent load pairs each left the launch latency-bound at 6 % of the HBM rate
    uint8_t ids[PS_PIX / 256];
    float lgs[PS_PIX / 256];
#pragma unroll
    for (int it = 0; it < PS_PIX / 256; ++it) {
        const long p = base + it * 256 + t;
        ids[it] = p < HW ? idmap[(long)b * HW + p] : 0;
        lgs[it] = p < HW ? logits[(long)b * nch * HW + p] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < PS_PIX / 256; ++it) {
        const long p = base + it * 256 + t;
        int slot = -1, y = 0, x = 0;
        double pr = 0.0;
        if (p < HW) {
            const float lab = slut[ids[it]];
            pan[(long)b * HW + p] = lab;
            if (lab >= 0.f) {
                slot = (int)lab - label_divisor;     // 1000 -> 0 (centre-less blob), 1001.. -> 0..
                if (slot > 0) slot -= 1;
                y = (int)(p / W);
                x = (int)(p - (long)y * W);
                pr = (double)(1.f / (1.f + expf(-lgs[it])));
            }
        }
        // one wave reduction per DISTINCT instance among the 64 lanes (one inside an object, two or three on a boundary): per-lane LDS
        // atomics on a boundary wave were 512 operations on two or three addresses
        unsigned long long remaining = __ballot(slot >= 0);
        while (remaining) {
            const int s = __shfl(slot, __ffsll((long long)remaining) - 1);
            const bool mine = slot == s;
            unsigned cnt = mine ? 1u : 0u, sy = mine ? (unsigned)y : 0u, sx = mine ? (unsigned)x : 0u;
            double pw = mine ? pr : 0.0;
            int x0 = mine ? x : 1 << 30, y0 = mine ? y : 1 << 30, x1 = mine ? x : -1, y1 = mine ? y : -1;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                cnt += __shfl_down(cnt, o);
                sy += __shfl_down(sy, o);
                sx += __shfl_down(sx, o);
                pw += __shfl_down(pw, o);
                x0 = min(x0, __shfl_down(x0, o)); y0 = min(y0, __shfl_down(y0, o));
                x1 = max(x1, __shfl_down(x1, o)); y1 = max(y1, __shfl_down(y1, o));
            }
            if ((t & 63) == 0) {
                atomicAdd(&s_prob[s], pw); atomicAdd(&s_cnt[s], cnt);
                atomicAdd(&s_sy[s], sy); atomicAdd(&s_sx[s], sx);
                atomicMin(&s_x0[s], x0); atomicMin(&s_y0[s], y0);
                atomicMax(&s_x1[s], x1); atomicMax(&s_y1[s], y1);
            }
            remaining &= ~__ballot(mine);
        }
    }
    __syncthreads();
    if (t < cap && s_cnt[t]) {
        InstStat* s = stats + (long)b * cap + t;
        atomicAdd(&s->prob, s_prob[t]);
        atomicAdd(&s->sy, (unsigned long long)s_sy[t]);
        atomicAdd(&s->sx, (unsigned long long)s_sx[t]);
        atomicAdd(&s->cnt, s_cnt[t]);
        atomicMin(&s->xmin, s_x0[t]);
        atomicMin(&s->ymin, s_y0[t]);
        atomicMax(&s->xmax, s_x1[t]);
        atomicMax(&s->ymax, s_y1[t]);
    }
}

// ---- P6: scores and boxes ----
__global__ void finalize_kernel(const float* __restrict__ logits, int nch, int H, int W, int cap,
                                const int* __restrict__ count, const InstStat* __restrict__ stats,
                                float* __restrict__ scores, float* __restrict__ boxes) {
    const int b = blockIdx.x;
    const long HW = (long)H * W;
    for (int i = threadIdx.x; i < cap; i += blockDim.x) {
        float sc = 0.f, bx[4] = {0.f, 0.f, 0.f, 0.f};
        if (i < count[b]) {
            const InstStat s = stats[(long)b * cap + i];
            if (s.cnt) {
                const float sem = (float)(s.prob / (double)s.cnt);
                const float my = (float)((double)s.sy / (double)s.cnt);
                const float mx = (float)((double)s.sx / (double)s.cnt);
                const float cs = logits[((long)b * nch + 1) * HW + (long)((int)my) * W + (int)mx];
                sc = sem * cs;
                bx[0] = (float)s.xmin; bx[1] = (float)s.ymin; bx[2] = (float)(s.xmax + 1); bx[3] = (float)(s.ymax + 1);
            }
        }
        scores[(long)b * cap + i] = sc;
        for (int k = 0; k < 4; ++k) boxes[((long)b * cap + i) * 4 + k] = bx[k];
    }
}

// ---- P7: per-instance binary masks ----
__global__ void extract_masks_kernel(const float* __restrict__ pan, const float* __restrict__ labels, int HW, int cap,
                                     int max_inst, uint8_t* __restrict__ out) {
    const int b = blockIdx.z, i = blockIdx.y;
    const float lab = labels[(long)b * cap + i];
    const long p = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (p >= HW) return;
    const float4* src = reinterpret_cast<const float4*>(pan + (long)b * HW + p);
    unsigned w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float4 v = src[k];
        w[k] = (lab >= 0.f && v.x == lab ? 1u : 0u) | (lab >= 0.f && v.y == lab ? 1u << 8 : 0u) |
               (lab >= 0.f && v.z == lab ? 1u << 16 : 0u) | (lab >= 0.f && v.w == lab ? 1u << 24 : 0u);
    }
    *reinterpret_cast<uint4*>(out + ((long)b * max_inst + i) * HW + p) = make_uint4(w[0], w[1], w[2], w[3]);
}

__global__ void extract_masks_generic_kernel(const float* __restrict__ pan, const float* __restrict__ labels, int HW,
                                             int cap, int max_inst, uint8_t* __restrict__ out) {
    const int b = blockIdx.z, i = blockIdx.y;
    const float lab = labels[(long)b * cap + i];
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    out[((long)b * max_inst + i) * HW + p] = (lab >= 0.f && pan[(long)b * HW + p] == lab) ? 1 : 0;
}

// workspace layout (bytes): cand f32 B*HW | idmap u8 B*HW | area u32 B*256 | lut f32 B*256 | stats B*cap
static size_t al(size_t x) { return (x + 255) & ~(size_t)255; }
size_t postprocess_ws_bytes(int B, int H, int W, int cap) {
    const size_t HW = (size_t)H * W;
    return al(B * HW * 4) + al(B * HW) + al((size_t)B * 256 * 4) * 2 + al((size_t)B * cap * sizeof(InstStat)) + al((size_t)B * sizeof(CandList));
}

int launch_postprocess(const float* logits, int nch, int B, int H, int W, const PostCfg& c, void* ws, float* pan,
                       int* count, float* labels, float* scores, float* boxes, int* centers, int* ncenters,
                       hipStream_t st) {
    if (c.cap > 254 || c.top_k > c.cap) return fail("postprocess: top_k must be <= capacity <= 254");
    if (nch < 4) return fail("postprocess: logits need >= 4 channels (fg, centre, off_y, off_x)");
    if ((c.nms_kernel & 1) == 0 || c.nms_kernel > 15) return fail("postprocess: NMS kernel must be odd and <= 15");
    if ((long)H * W < c.top_k) return fail("postprocess: frame smaller than top_k");
    const size_t HW = (size_t)H * W;
    char* w = reinterpret_cast<char*>(ws);
    float* cand = reinterpret_cast<float*>(w); w += al(B * HW * 4);
    uint8_t* idmap = reinterpret_cast<uint8_t*>(w); w += al(B * HW);
    unsigned* area = reinterpret_cast<unsigned*>(w); w += al((size_t)B * 256 * 4);
    float* lut = reinterpret_cast<float*>(w); w += al((size_t)B * 256 * 4);
    InstStat* stats = reinterpret_cast<InstStat*>(w); w += al((size_t)B * c.cap * sizeof(InstStat));
    CandList* lists = reinterpret_cast<CandList*>(w);

    const int r = (c.nms_kernel - 1) / 2;
    const int P = NT + 2 * r;
    const double px = (double)B * HW;
    if (int rc = launch_zero(lists, (size_t)B * sizeof(CandList), st)) return rc;      // (the counts; the slots behind them ride along)
    {   // a8: centre plane in, candidate map out
        ProfScope prof("post_nms", 8.0 * px, 0.0, st);
        hipLaunchKernelGGL(nms_kernel, dim3((W + NT - 1) / NT, (H + NT - 1) / NT, B), dim3(256), sizeof(float) * (P * P + P * NT), st,
                           logits, nch, H, W, c.threshold, r, cand, lists);
    }
    {   // a8: k-th largest candidate + ordered compaction (one block per frame; reads the candidate map)
        ProfScope prof("post_select", 4.0 * px, 0.0, st);
        hipLaunchKernelGGL(select_kernel, dim3(B), dim3(1024), 0, st, cand, (int)HW, W, c.top_k, c.cap, centers, ncenters, lists);
    }
    if (int rc = launch_zero(area, (size_t)B * 256 * 4, st)) return rc;
    {   // a9: fg + 2 offset planes in, id map out
        ProfScope prof("post_group", 13.0 * px, 0.0, st);
        hipLaunchKernelGGL(group_kernel, dim3((int)((HW + GP_PIX - 1) / GP_PIX), B), dim3(256), 0, st, logits, nch, H, W, c.cap, centers, ncenters,
                           idmap, area);
    }
    {
        ProfScope prof("post_relabel", 2048.0 * B, 0.0, st);
        hipLaunchKernelGGL(relabel_kernel, dim3(B), dim3(256), 0, st, area, ncenters, c.min_area, c.stuff_area,
                           c.label_divisor, c.cap, lut, count, labels, stats);
    }
    {   // a10 / a11: id map + fg plane in, label map out, per-instance sums
        ProfScope prof("post_paint_stats", 9.0 * px, 0.0, st);
        hipLaunchKernelGGL(paint_stats_kernel, dim3((int)((HW + PS_PIX - 1) / PS_PIX), B), dim3(256), 0, st, logits, nch, H, W, c.cap,
                           c.label_divisor, idmap, lut, pan, stats);
    }
    {
        ProfScope prof("post_finalize", 64.0 * B * c.cap, 0.0, st);
        hipLaunchKernelGGL(finalize_kernel, dim3(B), dim3(256), 0, st, logits, nch, H, W, c.cap, count, stats, scores,
                           boxes);
    }
    QB_CHECK(hipGetLastError());
    return 0;
}

// test harness (quber_op_group_pixels): a9 alone on a caller-supplied centre list
int launch_group_pixels(const float* logits, int nch, int B, int H, int W, int cap, const int* centers, const int* ncenters,
                        uint8_t* idmap, unsigned* area, hipStream_t st) {
    if (cap < 1 || cap > 254) return fail("group_pixels: capacity must be in 1..254");
    if (nch < 4) return fail("group_pixels: logits need >= 4 channels (fg, centre, off_y, off_x)");
    const long HW = (long)H * W;
    if (int rc = launch_zero(area, (size_t)B * 256 * 4, st)) return rc;
    hipLaunchKernelGGL(group_kernel, dim3((int)((HW + GP_PIX - 1) / GP_PIX), B), dim3(256), 0, st, logits, nch, H, W, cap, centers, ncenters,
                       idmap, area);
    QB_CHECK(hipGetLastError());
    return 0;
}

int launch_extract_masks(const float* pan, const float* labels, int B, int H, int W, int cap, int max_inst,
                         uint8_t* out, hipStream_t st) {
    const long HW = (long)H * W;
    if (max_inst < 1 || max_inst > cap) return fail("extract_masks: max_inst out of range");
    ProfScope prof("extract_masks", (double)B * HW * (4.0 + max_inst), 0.0, st);   // the label map once (L2 serves the re-reads), every mask byte once
    if (HW % 16 == 0 && (((uintptr_t)pan | (uintptr_t)out) & 15) == 0)
        hipLaunchKernelGGL(extract_masks_kernel, dim3((int)((HW / 16 + 255) / 256), max_inst, B), dim3(256), 0, st, pan,
                           labels, (int)HW, cap, max_inst, out);
    else
        hipLaunchKernelGGL(extract_masks_generic_kernel, dim3((int)((HW + 255) / 256), max_inst, B), dim3(256), 0, st,
                           pan, labels, (int)HW, cap, max_inst, out);
    QB_CHECK(hipGetLastError());
    return 0;
}

}  // namespace quber
