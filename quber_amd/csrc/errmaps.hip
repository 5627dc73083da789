// Explicit quadruple error maps (region + boundary TP/TN/FP/FN) of a set of initial masks against a
// set of ground-truth masks.  Replaces
//   explicit_error_estimation/util.py:62-68   masks_to_fg_mask   (uint8 wrap-around sum > 0)
//   explicit_error_estimation/util.py:72-90   mask_to_boundary   (3x3 erosion x d on the zero-padded mask)
//   explicit_error_estimation/util.py:92-99   masks_to_boundary
//   tools/ours/panoptic2eee.py:110-123        TP/TN/FP/FN
//
// Bit-plane pipeline (binary masks, one non-zero value v per mask set - what `mask.astype(np.uint8)` / `* 255` give):
//   1. pack      every mask byte is read ONCE, 16 bytes per lane, and leaves as one bit: 64-pixel words assembled with
//                two wave shuffles (or one __ballot for widths that are not a multiple of 16); per set, the OR and the
//                complement-OR of the non-zero bytes tell whether the set is binary with a single value.
//   2. erode     d iterations of a 3x3 erosion with a zero border = a (2d+1)^2 minimum = AND of the bits in that window:
//                horizontally shifted ANDs across word boundaries, then ANDs of the rows above / below, on the bit
//                planes (1/8 of the mask bytes).  boundary = mask & ~eroded  (uint8 `mask - eroded`, util.py:90).
//   3. quadruple per pixel the reference sums N bytes in uint8 (wrap-around) and tests > 0: with one value v the sum is
//                count * v mod 256, non-zero iff the low 8 - ctz(v) bits of count are not all zero.  The counts are kept
//                bit-sliced (8 planes, 16 pixels per lane per plane, ripple-carry increment), then the four one-hot
//                classes of (gt, input) are written, 16 bytes per lane per output plane.
// Masks with several distinct non-zero values are grey-level images to cv2.erode (a minimum filter, not a logical one):
// the byte-wise kernels below (LDS tiles, separable minimum) handle them; each path returns at once when the flags
// computed by pass 1 say the call belongs to the other one, so the launch sequence is fixed and graph-capturable.
#include "common.h"

namespace quber {

using u64 = unsigned long long;

struct SetStat { unsigned or_all, nand_all; };   // OR of the bytes; OR of ~byte over the NON-ZERO bytes  (per mask set)

__device__ inline bool set_uniform(const SetStat& s) { return (s.or_all & s.nand_all & 0xffu) == 0; }

// OR the bits a wave saw into the set's flags.  Every wave of a call sees the same one or two byte values, so after the
// first arrivals nothing is new: a relaxed read first keeps ~10^5 same-address atomics per call (418 us measured) down to
// a handful.  A stale read only costs a redundant atomic.
__device__ inline void stat_merge(SetStat* stat, unsigned any_or, unsigned any_nand) {
    const unsigned cur_or = __hip_atomic_load(&stat->or_all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned cur_nand = __hip_atomic_load(&stat->nand_all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (any_or & ~cur_or) atomicOr(&stat->or_all, any_or);
    if (any_nand & ~cur_nand) atomicOr(&stat->nand_all, any_nand);
}

// ---- 1. pack ---------------------------------------------------------------------------------------------------------
// grid (ceil(H * wpr * 4 / (256 * PACK_IT)), n masks, B); a thread packs PACK_IT 16-pixel groups of the row-padded group grid
// (wpr * 4 per row), 256 groups apart, their 16-byte loads requested together (one load per thread and block left the launch at a
// quarter of the HBM rate: 48 000 blocks per 16-frame step)
constexpr int PACK_IT = 4;

__global__ __launch_bounds__(256) void errmaps_pack_kernel(const uint8_t* __restrict__ masks, int N, int H, int W, int wpr,
                                                           u64* __restrict__ mbits, int n0, int Ntot,
                                                           SetStat* __restrict__ stat) {
    const int n = blockIdx.y, b = blockIdx.z;
    const long HW = (long)H * W;
    const int gpr = wpr * 4;
    uint4 v[PACK_IT];
    int ys[PACK_IT], gxs[PACK_IT];
#pragma unroll
    for (int it = 0; it < PACK_IT; ++it) {
        const long gid = ((long)blockIdx.x * PACK_IT + it) * 256 + threadIdx.x;
        const int y = (int)(gid / gpr), gx = (int)(gid - (long)y * gpr);
        ys[it] = y; gxs[it] = gx;
        v[it] = make_uint4(0u, 0u, 0u, 0u);
        if (y < H && gx * 16 < W) v[it] = *reinterpret_cast<const uint4*>(masks + ((long)b * N + n) * HW + (long)y * W + gx * 16);
    }
    unsigned any_or = 0, any_nand = 0;
#pragma unroll
    for (int it = 0; it < PACK_IT; ++it) {
        const int y = ys[it], gx = gxs[it];
        unsigned piece = 0;
        const unsigned wds[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned wd = wds[j];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned byte = (wd >> (8 * k)) & 0xffu;
                if (byte) {
                    piece |= 1u << (4 * j + k);
                    any_or |= byte;
                    any_nand |= ~byte & 0xffu;
                }
            }
        }
        // four consecutive lanes hold the four 16-bit pieces of one word
        unsigned pair = piece | (__shfl_down(piece, 1) << 16);
        const unsigned hi = __shfl_down(pair, 2);
        if ((threadIdx.x & 3) == 0 && y < H)
            mbits[(((long)b * Ntot + n0 + n) * H + y) * wpr + (gx >> 2)] = (u64)pair | ((u64)hi << 32);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        any_or |= __shfl_down(any_or, o);
        any_nand |= __shfl_down(any_nand, o);
    }
    if ((threadIdx.x & 63) == 0 && any_or) stat_merge(stat, any_or, any_nand);
}

// widths that are not a multiple of 16: one pixel per lane, one 64-pixel word per wave via __ballot
__global__ __launch_bounds__(256) void errmaps_pack_ballot_kernel(const uint8_t* __restrict__ masks, int N, int H, int W,
                                                                  int wpr, u64* __restrict__ mbits, int n0, int Ntot,
                                                                  SetStat* __restrict__ stat) {
    const int n = blockIdx.y, b = blockIdx.z;
    const long HW = (long)H * W;
    const long word = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int y = (int)(word / wpr), wx = (int)(word - (long)y * wpr);
    const int x = wx * 64 + (threadIdx.x & 63);
    unsigned byte = 0;
    if (y < H && x < W) byte = masks[((long)b * N + n) * HW + (long)y * W + x];
    const u64 bits = __ballot(byte != 0);
    unsigned any_or = byte, any_nand = byte ? (~byte & 0xffu) : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        any_or |= __shfl_down(any_or, o);
        any_nand |= __shfl_down(any_nand, o);
    }
    if ((threadIdx.x & 63) == 0) {
        if (y < H) mbits[(((long)b * Ntot + n0 + n) * H + y) * wpr + wx] = bits;
        if (any_or) stat_merge(stat, any_or, any_nand);
    }
}

// ---- 2. erode --------------------------------------------------------------------------------------------------------
constexpr int ER_ROWS = 32;      // output rows per block
constexpr int ER_MAX_WPR = 32;   // widths up to 2048 pixels

// grid (ceil(H / ER_ROWS), Ntot, B): boundary plane = mask & ~(mask eroded d times by a 3x3 square, zero border)
__global__ __launch_bounds__(256) void errmaps_erode_kernel(const u64* __restrict__ mbits, int H, int wpr, int d,
                                                            u64* __restrict__ bbits, const SetStat* __restrict__ stat) {
    if (!set_uniform(stat[0]) || !set_uniform(stat[1])) return;          // grey-level masks: the byte-wise path runs
    extern __shared__ u64 er_lds[];              // m[(ER_ROWS + 2d)][wpr], h[(ER_ROWS + 2d)][wpr]
    const int rows = ER_ROWS + 2 * d;
    u64* m = er_lds;
    u64* h = er_lds + rows * wpr;
    const long plane = ((long)blockIdx.z * gridDim.y + blockIdx.y) * H;
    const int r0 = blockIdx.x * ER_ROWS;
    for (int i = threadIdx.x; i < rows * wpr; i += 256) {
        const int ly = i / wpr, wx = i - ly * wpr;
        const int y = r0 + ly - d;
        m[i] = (unsigned)y < (unsigned)H ? mbits[(plane + y) * wpr + wx] : 0ull;
    }
    __syncthreads();
    // horizontal: AND of the 2d+1 shifted copies; bit x of word w is pixel 64 w + x, so "pixel x - s" is a left shift
    for (int i = threadIdx.x; i < rows * wpr; i += 256) {
        const int ly = i / wpr, wx = i - ly * wpr;
        const u64 c = m[i];
        const u64 l = wx > 0 ? m[i - 1] : 0ull, r = wx + 1 < wpr ? m[i + 1] : 0ull;
        u64 acc = c;
        for (int s = 1; s <= d && acc; ++s) {
            if (s < 64) {
                acc &= (c << s) | (l >> (64 - s));
                acc &= (c >> s) | (r << (64 - s));
            } else {
                acc = 0;         // d <= 32 (launcher), kept for completeness
            }
        }
        h[i] = acc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ER_ROWS * wpr; i += 256) {
        const int ly = i / wpr, wx = i - ly * wpr;
        const int y = r0 + ly;
        if (y >= H) continue;
        u64 acc = h[(ly + d) * wpr + wx];
        for (int s = 1; s <= d && acc; ++s) acc &= h[(ly + d - s) * wpr + wx] & h[(ly + d + s) * wpr + wx];
        bbits[(plane + y) * wpr + wx] = m[(ly + d) * wpr + wx] & ~acc;
    }
}

// ---- 3. quadruple ----------------------------------------------------------------------------------------------------
__device__ inline void count_inc(unsigned (&p)[8], unsigned bits) {      // p += bits (bit-sliced, mod 256)
    unsigned c = bits;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const unsigned t = p[k] & c;
        p[k] ^= c;
        c = t;
    }
}
__device__ inline unsigned count_nonzero_low(const unsigned (&p)[8], int k) {   // (count mod 2^k) != 0, per pixel
    unsigned r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (i < k) r |= p[i];
    return r;
}
// sum of n copies of v in uint8 is non-zero  <=>  the low (8 - ctz(v)) bits of n are not all zero  (v != 0)
__device__ inline int low_bits_for(unsigned v) { return v ? 8 - (__ffs((int)v) - 1) : 0; }

__device__ inline uint4 expand16(unsigned bits) {       // 16 bits -> 16 bytes in {0,1}
    unsigned w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned nib = (bits >> (4 * j)) & 0xfu;
        w[j] = (nib & 1u) | ((nib & 2u) << 7) | ((nib & 4u) << 14) | ((nib & 8u) << 21);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// grid (ceil(H * wpr * 4 / 256), B); thread = 16 pixels; out[b][2][4][H][W]
__global__ __launch_bounds__(256) void errmaps_quadruple_bits_kernel(const u64* __restrict__ mbits, const u64* __restrict__ bbits,
                                                                     int N, int Ng, int H, int W, int wpr,
                                                                     const SetStat* __restrict__ stat, uint8_t* __restrict__ out) {
    const SetStat s_in = stat[0], s_gt = stat[1];
    if (!set_uniform(s_in) || !set_uniform(s_gt)) return;
    const int b = blockIdx.y, Ntot = N + Ng;
    const int gpr = wpr * 4;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const int y = (int)(gid / gpr), gx = (int)(gid - (long)y * gpr);
    if (y >= H || gx * 16 >= W) return;
    const unsigned short* m16 = reinterpret_cast<const unsigned short*>(mbits);
    const unsigned short* b16 = reinterpret_cast<const unsigned short*>(bbits);
    unsigned fg_in[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bd_in[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned fg_gt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bd_gt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long row = ((long)b * Ntot * H + y) * gpr + gx;       // in 16-bit units; plane stride H * gpr
    const long ps = (long)H * gpr;
    for (int n = 0; n < N; ++n) {
        count_inc(fg_in, m16[row + n * ps]);
        count_inc(bd_in, b16[row + n * ps]);
    }
    for (int n = N; n < Ntot; ++n) {
        count_inc(fg_gt, m16[row + n * ps]);
        count_inc(bd_gt, b16[row + n * ps]);
    }
    const int k_in = low_bits_for(s_in.or_all & 0xffu), k_gt = low_bits_for(s_gt.or_all & 0xffu);
    const unsigned q[2][2] = {{count_nonzero_low(fg_gt, k_gt), count_nonzero_low(fg_in, k_in)},
                              {count_nonzero_low(bd_gt, k_gt), count_nonzero_low(bd_in, k_in)}};
    const long HW = (long)H * W;
    const bool vec = (W & 15) == 0 && (HW & 15) == 0;
    const int npx = min(16, W - gx * 16);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const unsigned g = q[k][0] & 0xffffu, in = q[k][1] & 0xffffu, all = 0xffffu;
        const unsigned cls[4] = {g & in, ~g & ~in & all, ~g & in & all, g & ~in & all};   // TP, TN, FP, FN
        uint8_t* o = out + (((long)b * 2 + k) * 4) * HW + (long)y * W + gx * 16;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (vec) {
                *reinterpret_cast<uint4*>(o + c * HW) = expand16(cls[c]);
            } else {
                for (int i = 0; i < npx; ++i) o[c * HW + i] = (cls[c] >> i) & 1u;
            }
        }
    }
}

// ---- byte-wise path: grey-level masks (several distinct non-zero values in a set) ------------------------------------
// d iterations of a 3x3 erosion with a zero border equal one (2d+1)^2 minimum filter, evaluated separably (row minimum,
// then column minimum) on an LDS tile with a d-pixel halo.
constexpr int ET_H = 32, ET_W = 64;

// grid (tiles_x, tiles_y, B); writes fg[b][p] and bnd[b][p] in {0,1}
__global__ __launch_bounds__(256) void fg_boundary_kernel(const uint8_t* __restrict__ masks, int N, int H, int W, int d,
                                                          uint8_t* __restrict__ fg, uint8_t* __restrict__ bnd,
                                                          const SetStat* __restrict__ stat) {
    if (set_uniform(stat[0]) && set_uniform(stat[1])) return;            // binary masks: the bit-plane path runs
    extern __shared__ uint8_t tile[];  // s0[(ET_H+2d)][(ET_W+2d)], s1[(ET_H+2d)][ET_W]
    const int PW = ET_W + 2 * d, PH = ET_H + 2 * d;
    uint8_t* s0 = tile;
    uint8_t* s1 = tile + PH * PW;
    const int b = blockIdx.z;
    const int ty0 = blockIdx.y * ET_H, tx0 = blockIdx.x * ET_W;
    const long HW = (long)H * W;
    constexpr int PPT = ET_H * ET_W / 256;  // pixels per thread
    unsigned accfg[PPT], accb[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) accfg[i] = accb[i] = 0;

    for (int n = 0; n < N; ++n) {
        const uint8_t* m = masks + ((long)b * N + n) * HW;
        for (int i = threadIdx.x; i < PH * PW; i += 256) {
            const int ly = i / PW, lx = i - ly * PW;
            const int gy = ty0 + ly - d, gx = tx0 + lx - d;
            s0[i] = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? m[(long)gy * W + gx] : 0;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < PH * ET_W; i += 256) {
            const int ly = i / ET_W, lx = i - ly * ET_W;
            const uint8_t* r = s0 + ly * PW + lx;
            unsigned v = 255;
            for (int k = 0; k <= 2 * d; ++k) v = min(v, (unsigned)r[k]);
            s1[i] = (uint8_t)v;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int idx = threadIdx.x + 256 * i;
            const int ly = idx / ET_W, lx = idx - ly * ET_W;
            unsigned v = 255;
            for (int k = 0; k <= 2 * d; ++k) v = min(v, (unsigned)s1[(ly + k) * ET_W + lx]);
            const unsigned mv = s0[(ly + d) * PW + lx + d];
            accfg[i] += mv;
            accb[i] += (mv - v) & 0xffu;   // uint8 `mask - eroded`
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int idx = threadIdx.x + 256 * i;
        const int ly = idx / ET_W, lx = idx - ly * ET_W;
        const int gy = ty0 + ly, gx = tx0 + lx;
        if (gy < H && gx < W) {
            const long o = (long)b * HW + (long)gy * W + gx;
            fg[o] = (accfg[i] & 0xffu) ? 1 : 0;    // the reference accumulates in uint8
            bnd[o] = (accb[i] & 0xffu) ? 1 : 0;
        }
    }
}

// ws = [gt_fg, in_fg, gt_bnd, in_bnd] each B*H*W ; out[b][2][4][H][W]
__global__ void quadruple_kernel(const uint8_t* __restrict__ ws, long BHW, long HW, uint8_t* __restrict__ out,
                                 const SetStat* __restrict__ stat) {
    if (set_uniform(stat[0]) && set_uniform(stat[1])) return;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < BHW; i += (long)gridDim.x * blockDim.x) {
        const long b = i / HW, p = i - b * HW;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const bool g = ws[(2 * k) * BHW + i] != 0, in = ws[(2 * k + 1) * BHW + i] != 0;
            uint8_t* o = out + ((b * 2 + k) * 4) * HW + p;
            o[0] = g && in;
            o[HW] = !g && !in;
            o[2 * HW] = !g && in;
            o[3 * HW] = g && !in;
        }
    }
}

static inline int words_per_row(int W) { return (W + 63) / 64; }
static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

// [SetStat x 2 (256 B)] [byte-wise path: 4 * B*H*W] [mask bit planes] [boundary bit planes]   (Nmax masks per set)
size_t errmaps_ws_bytes(int B, int Nmax, int H, int W) {
    const size_t planes = (size_t)B * 2 * Nmax * H * words_per_row(W) * sizeof(u64);
    return 256 + al256((size_t)4 * B * H * W) + 2 * al256(planes);
}

int launch_errmaps(const uint8_t* init, int N, const uint8_t* gt, int Ng, int B, int Nmax, int H, int W, int d, uint8_t* ws,
                   uint8_t* out, hipStream_t st) {
    if (d < 1 || d > 32) return fail("error maps: boundary width out of range (1..32)");
    if (N < 0 || Ng < 0 || (long)N + Ng > 2L * Nmax) return fail("error maps: more masks (init + gt) than 2 x max_instances");
    if (words_per_row(W) > ER_MAX_WPR) return fail("error maps: frames wider than 2048 pixels are not supported");
    const long HW = (long)H * W, BHW = (long)B * HW;
    const int wpr = words_per_row(W), Ntot = N + Ng;
    SetStat* stat = reinterpret_cast<SetStat*>(ws);
    uint8_t* bytews = ws + 256;
    const size_t planes = al256((size_t)B * 2 * Nmax * H * wpr * sizeof(u64));
    u64* mbits = reinterpret_cast<u64*>(bytews + al256((size_t)4 * BHW));
    u64* bbits = reinterpret_cast<u64*>(reinterpret_cast<uint8_t*>(mbits) + planes);
    if (int rc = launch_zero(stat, 2 * sizeof(SetStat), st)) return rc;
    {   // every mask byte once in, one bit out
        ProfScope prof("errmaps_pack", (double)BHW * Ntot * (1.0 + 1.0 / 8), 0.0, st);
        const bool fast = (W & 15) == 0 && (((uintptr_t)init | (uintptr_t)gt) & 15) == 0;
        for (int s = 0; s < 2; ++s) {
            const uint8_t* src = s ? gt : init;
            const int n = s ? Ng : N, n0 = s ? N : 0;
            if (n == 0) continue;
            if (fast)
                hipLaunchKernelGGL(errmaps_pack_kernel, dim3((unsigned)(((long)H * wpr * 4 + 256 * PACK_IT - 1) / (256 * PACK_IT)), n, B), dim3(256), 0, st,
                                   src, n, H, W, wpr, mbits, n0, Ntot, stat + s);
            else
                hipLaunchKernelGGL(errmaps_pack_ballot_kernel, dim3((unsigned)(((long)H * wpr + 3) / 4), n, B), dim3(256), 0, st,
                                   src, n, H, W, wpr, mbits, n0, Ntot, stat + s);
        }
    }
    QB_CHECK(hipGetLastError());
    if (Ntot > 0) {   // bit planes in, boundary bit planes out
        ProfScope prof("errmaps_erode", (double)B * Ntot * H * wpr * 16.0, 0.0, st);
        const size_t sm = (size_t)2 * (ER_ROWS + 2 * d) * wpr * sizeof(u64);
        hipLaunchKernelGGL(errmaps_erode_kernel, dim3((H + ER_ROWS - 1) / ER_ROWS, Ntot, B), dim3(256), sm, st, mbits, H, wpr, d,
                           bbits, stat);
    }
    QB_CHECK(hipGetLastError());
    {   // both bit-plane sets in, eight one-hot byte planes out
        ProfScope prof("errmaps_quadruple", (double)BHW * (8.0 + Ntot * 2.0 / 8), 0.0, st);
        hipLaunchKernelGGL(errmaps_quadruple_bits_kernel, dim3((unsigned)(((long)H * wpr * 4 + 255) / 256), B), dim3(256), 0, st,
                           mbits, bbits, N, Ng, H, W, wpr, stat, out);
    }
    QB_CHECK(hipGetLastError());
    // grey-level masks only (each kernel returns at once otherwise)
    const size_t sm = (size_t)(ET_H + 2 * d) * (ET_W + 2 * d) + (size_t)(ET_H + 2 * d) * ET_W;
    dim3 grid((W + ET_W - 1) / ET_W, (H + ET_H - 1) / ET_H, B);
    hipLaunchKernelGGL(fg_boundary_kernel, grid, dim3(256), sm, st, gt, Ng, H, W, d, bytews, bytews + 2 * BHW, stat);
    hipLaunchKernelGGL(fg_boundary_kernel, grid, dim3(256), sm, st, init, N, H, W, d, bytews + BHW, bytews + 3 * BHW, stat);
    int blocks = (int)((BHW + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(quadruple_kernel, dim3(blocks), dim3(256), 0, st, bytews, BHW, HW, out, stat);
    QB_CHECK(hipGetLastError());
    return 0;
}

}  // namespace quber
