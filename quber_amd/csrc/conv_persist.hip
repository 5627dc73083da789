// Persistent form of the implicit-GEMM convolution (conv_igemm.hip): the same tiles, LDS image and MFMA mapping, but the
// grid is one block per resident slot of the chip and every block walks a fixed list of K-ranges of tiles ("segments")
// instead of owning one tile:
//   * whole tiles, round-robin inside the block's XCD - the blocks of an XCD are on consecutive tiles at the same K phase,
//     as in the one-tile-per-block launch, so they share their A rows / weight rows in that XCD's L2;
//   * then, where K is long enough to pay for it (launch_conv_persistent), an equal share of the K-slices of the tiles
//     left over when the tile count is not a multiple of the slots (stream-K on the remainder only): a block finishes
//     someone's tile, computes tiles in between whole, and starts one more.  Partial tiles go to two workspace slots per
//     block and pk_fixup_kernel sums them in block order (deterministic) under the fused epilogue.
// What this buys over one tile per block: the first K-slice of the next segment is fetched while the last slice of the
// current one is multiplied and its epilogue needs neither LDS nor barriers, so a tile boundary costs the matrix pipe
// little - that is what short-K launches (the Winograd GEMMs, K = 128-512; the bottleneck 1x1s) lose most of their time
// to - and long-K launches end without a ragged last round.  Global memory is addressed through buffer descriptors
// (hardware range check instead of predicates).  Epilogue variants: affine + ReLU, + residual, + GroupNorm sums, and a
// 1x1 convolution over TWO inputs (a bottleneck's conv3 and its projection shortcut as one GEMM).
// Layers with padded filter rows skipped (MODE 3 / 4 of conv_igemm.hip) keep the one-tile-per-block kernel: their K range
// depends on the tile; so do the 16-bit operand modes and launches of a few dozen tiles (DECISIONS.md section 4).
#include "common.h"

namespace quber {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int BK = 32;
constexpr int PITCH = 36;
template <int DT> struct Half16 { using T = __bf16; };
template <> struct Half16<2> { using T = _Float16; };
template <> struct Half16<4> { using T = _Float16; };     // DT 4: the fp16 data path (fp16 activations and weights in HBM)
constexpr int PITCH_H = 40;
constexpr int NPL(int dt) { return dt == 3 ? 3 : 1; }

// Work of one XCD (blocks with blockIdx.x % 8 == xcd; PX of them): the contiguous run [c0, c0 + cn) of the T tiles
// (all groups, group-major, n-tile fastest).  R whole rounds, then `rem` tiles whose U = rem * nk K-slices are dealt
// out evenly to the first PXs blocks (every share at least min_slices long).
struct PkPlan { int c0, cn, PX, R, rem, U, PXs; };

__host__ __device__ inline PkPlan pk_plan(int T, int P, int xcd, int nk, int min_slices) {
    PkPlan s;
    const int q = T >> 3, r = T & 7;
    s.c0 = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    s.cn = q + (xcd < r ? 1 : 0);
    s.PX = P >> 3;
    s.R = s.cn / s.PX;
    s.rem = s.cn - s.R * s.PX;
    s.U = s.rem * nk;
    if (min_slices <= 0) {             // no sharing: the remainder tiles go whole to the first `rem` blocks
        s.PXs = s.rem > 0 ? s.rem : 1;
        return s;
    }
    int n = s.U / min_slices;
    if (n < 1) n = 1;
    s.PXs = n < s.PX ? n : s.PX;
    return s;
}
__host__ __device__ inline int pk_u0(const PkPlan& s, int l) { return (int)((long)l * s.U / s.PXs); }
// the block whose share holds K-slice u of the remainder
__host__ __device__ inline int pk_owner(const PkPlan& s, int u) {
    int b = (int)(((long)(u + 1) * s.PXs + s.U - 1) / s.U) - 1;
    if (b < 0) b = 0;
    if (b > s.PXs - 1) b = s.PXs - 1;
    while (b + 1 < s.PXs && pk_u0(s, b + 1) <= u) ++b;
    while (b > 0 && pk_u0(s, b) > u) --b;
    return b;
}

#ifdef PK_STAMPS
// diagnostic build (make STAMPS=1, tools/pk_stamps.py): s_memtime at the phase boundaries of the first tiles of the first blocks
constexpr int STAMP_BLOCKS = 48, STAMP_TILES = 24, STAMP_N = 12;
__device__ unsigned long long g_pk_stamps[STAMP_BLOCKS * STAMP_TILES * STAMP_N];
__device__ unsigned long long g_pk_span[2048 * 4];      // per block: kernel entry, first K loop, exit (s_memrealtime: 100 MHz, one counter for the chip), HW_ID
#define PK_STAMP(i) do { if (stamp_on && stamp_tile < STAMP_TILES) g_pk_stamps[((int)blockIdx.x * STAMP_TILES + stamp_tile) * STAMP_N + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PK_STAMP(i) do {} while (0)
#endif
// resident blocks per CU the kernel is built for (= waves per SIMD: a block is one wave on each SIMD)
// (two accumulator sets since round 3 - the MFMA chain and the chunk sums, see `fold` below: 64 + 64 registers for a 128x128 tile)
constexpr int pk_occupancy(int BM, int DT) { return DT == 4 ? 3 : BM == 64 ? 5 : 2; }

// All global accesses go through buffer descriptors (base + 32-bit byte offset, hardware range check):
//   * an out-of-image tap, or a lane whose channel is past Cout, adds OOB to its offset - loads return 0 and stores are
//     dropped, so neither the loader nor the epilogue carries predicates or zero-selects;
//   * the output descriptor of a tile ends at the end of the tensor, which drops the rows past M of the last m-tile.
// Every view therefore has to stay below 2 GiB (host check; larger launches keep the one-tile-per-block kernel).
constexpr unsigned OOB = 0x80000000u;
constexpr int RSRC_FLAGS = 0x00020000;
// the descriptor words pinned to SGPRs: a block-uniform value that the compiler happened to compute on the vector ALU
// (scalar registers are scarce in this kernel) would otherwise get a readfirstlane "waterfall" loop around EVERY access
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* ptr, int bytes) {
    const unsigned long a = (unsigned long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane(bytes), RSRC_FLAGS);
}
// A block-uniform pointer into GLOBAL memory: address in SGPRs, and typed as address space 1 so that loads through it
// are global_load (vmcnt only) - a generic pointer gives flat_load, whose wait (lgkmcnt) also covers the LDS stores
// just issued for the next tile's first K-slice.
using gf32x4_ptr = const f32x4 __attribute__((address_space(1)))*;
__device__ __forceinline__ gf32x4_ptr uniform_gptr(const float* ptr) {
    const unsigned long a = (unsigned long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return (gf32x4_ptr)(((unsigned long)hi << 32) | lo);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
    return __builtin_bit_cast(f32x4, v);
}
// 4x4 transposition inside every quad of lanes: in, a_k at quad lane q = M[k][q]; out, a_k at quad lane q = M[q][k].
// Two exchange steps (lane bit 0 with register bit 0, then bit 1 with bit 1), one DPP quad_perm move + select per register each.
__device__ __forceinline__ float dpp_quad_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float dpp_quad_xor2(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
}
__device__ __forceinline__ void quad_transpose(float& a0, float& a1, float& a2, float& a3, bool q0, bool q1) {
    // the exchanges are evaluated for every lane BEFORE the selects: inside a branch a DPP move would read disabled lanes
    const float x0 = dpp_quad_xor1(a0), x1 = dpp_quad_xor1(a1), x2 = dpp_quad_xor1(a2), x3 = dpp_quad_xor1(a3);
    const float b0 = q0 ? x1 : a0;
    const float b1 = q0 ? a1 : x0;
    const float b2 = q0 ? x3 : a2;
    const float b3 = q0 ? a3 : x2;
    const float y0 = dpp_quad_xor2(b0), y1 = dpp_quad_xor2(b1), y2 = dpp_quad_xor2(b2), y3 = dpp_quad_xor2(b3);
    a0 = q1 ? y2 : b0;
    a1 = q1 ? y3 : b1;
    a2 = q1 ? b2 : y0;
    a3 = q1 ? b3 : y1;
}
// 4 halfs (8 bytes) of the fp16 data path
using h16x4_t = __attribute__((ext_vector_type(4))) _Float16;
__device__ __forceinline__ f32x4 buf_load4h(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0);
    const h16x4_t hv = __builtin_bit_cast(h16x4_t, v);
    return f32x4{(float)hv.x, (float)hv.y, (float)hv.z, (float)hv.w};
}
__device__ __forceinline__ h16x4_t buf_store4h(f32x4 v, __amdgpu_buffer_rsrc_t r, unsigned voff) {
    using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
    const h16x4_t hv = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hv), r, voff, 0, 0);
    return hv;
}
__device__ __forceinline__ void buf_store4(f32x4 v, __amdgpu_buffer_rsrc_t r, unsigned voff) {
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, 0, 0);
}

// EPI: what the fused epilogue carries besides the affine and the ReLU - 0 nothing, 1 the residual add, 2 the GroupNorm sums
// of the stored values (fp64 sum and sum of squares per (image, norm group), as conv_igemm_f32)
template <int BM, int BN, int WM, int WN, int DT, int EPI>
__global__ __launch_bounds__(WM * WN * 64) __attribute__((amdgpu_waves_per_eu(pk_occupancy(BM, DT))))
void conv_igemm_pk(const ConvP p) {
    constexpr int NTH = WM * WN * 64;
    constexpr int RPP = NTH / 8;
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int AL = BM / RPP;
    constexpr int BL = BN / RPP;
    static_assert(BM % RPP == 0 && BN % RPP == 0, "loader geometry");
    static_assert(DT == 0 || DT == 3 || DT == 4, "the fp32-activation 16-bit operand modes keep the one-tile-per-block kernel");
    // DT 4 (fp16 data path): operands travel as raw 4-byte units (two halfs), exactly like conv_igemm_f32<.., 4>: the loader,
    // the LDS image and its pitch are the fp32 kernel's, a K-slice carries 64 halfs, outputs / residuals are fp16
    constexpr bool H16IO = DT == 4;
    constexpr bool TWO = DT != 4;          // two-level accumulation (the 16-bit mode keeps one chain: its tolerance is 100x wider)
    constexpr bool RES = EPI == 1, GN = EPI == 2;
    // EPI 3: a 1x1 convolution whose K runs over TWO inputs - p.K1 channels of `in` (stride 1), then the channels of `in2`
    // sampled with stride p.stride2: a bottleneck's conv3 and its projection shortcut as one GEMM (BN scales folded into
    // the packed weights, shifts added), which keeps the shortcut's output out of HBM altogether
    constexpr bool DUAL = EPI == 3;
    __shared__ double gacc[GN ? 2 * 32 * 2 : 1];       // [image b0 / b0 + 1][norm group][sum, sum of squares] of the tile being stored
    // Two K-slice images for the exact fp32 kernel on 128x128 tiles (2 blocks per CU: 2 x 72 KiB of the 160 KiB): the next
    // slice is stored into the image nobody reads while this one is multiplied, and ONE barrier per slice publishes it -
    // with two resident blocks instead of three there is less to hide a second barrier and the store phase behind.
    constexpr int NBUF = (DT == 0 && BM == 128 && BN == 128) ? 2 : 1;
    constexpr int IMG_FLOATS = DT == 3 ? NPL(DT) * (BM + BN) * PITCH_H / 2 : (BM + BN) * PITCH;
    constexpr int SMEM_FLOATS = NBUF * IMG_FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
    int buf = 0;                        // image being multiplied (NBUF == 2)
    float* const As = smem;
    float* const Bs = smem + BM * PITCH;
    using H16 = typename Half16<DT>::T;
    using h16x8 = __attribute__((ext_vector_type(8))) H16;
    using h16x4 = __attribute__((ext_vector_type(4))) H16;
    H16* const Ah = reinterpret_cast<H16*>(smem);
    H16* const Bh = Ah + NPL(DT) * BM * PITCH_H;

    const int t = threadIdx.x;
    if constexpr (GN) {
        if (t < 128) gacc[t] = 0.0;       // published by the barriers of the first K-slice
    }
#ifdef PK_STAMPS
    if (t == 0 && blockIdx.x < 2048) {
        g_pk_span[blockIdx.x * 4] = __builtin_amdgcn_s_memrealtime();
        g_pk_span[blockIdx.x * 4 + 3] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492);
    }
#endif
    const int nkt = p.Kpad / BK;
    const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
    const PkPlan pl = pk_plan(p.pk_T, gridDim.x, xcd, nkt, p.pk_min);

    // ---- segment cursor (block-uniform) ----
    int round = 0, u = 0, u_end = 0;
    if (l < pl.PXs) { u = pk_u0(pl, l); u_end = pk_u0(pl, l + 1); }
    const int u_first = u;
    int s_gt = 0, s_k0 = 0, s_k1 = 0, s_slot = 0;      // the segment handed out last: tile, K-slices [k0, k1), workspace slot if partial
    auto next_seg = [&]() -> bool {
        if (round < pl.R) {
            s_gt = pl.c0 + round * pl.PX + l; s_k0 = 0; s_k1 = nkt; s_slot = 0;
            ++round;
            return true;
        }
        if (u < u_end) {
            const int j = u / nkt;
            s_k0 = u - j * nkt;
            s_k1 = min(nkt, s_k0 + (u_end - u));
            s_gt = pl.c0 + pl.R * pl.PX + j;
            s_slot = 2 * (int)blockIdx.x + (u == u_first ? 0 : 1);
            u += s_k1 - s_k0;
            return true;
        }
        return false;
    };

    // ---- loader state of the segment being fetched ----
    const int kq = (t & 7) * 4;
    const int lrow = t >> 3;
    int iy0[AL], ix0[AL];
    unsigned rowoff[AL];                // byte offset of the row's window origin (may be "negative": wraps, fixed by + off)
    unsigned wrow[BL];                  // byte offset of the weight row at this thread's k column
    __amdgpu_buffer_rsrc_t rs_in = make_rsrc(p.in, 0);
    __amdgpu_buffer_rsrc_t rs_w = rs_in, rs_in2 = rs_in;
    int kc = 0, kx = 0, ky = 0, ks = 0;
    int n_k0 = 0;                      // first K-slice of that segment
    int n_m0 = 0, n_n0 = 0, n_g = 0;   // tile origin / group of that segment
    auto setup = [&]() __attribute__((always_inline)) {
        const int g = s_gt / p.pk_tpg;
        const int tile = s_gt - g * p.pk_tpg;
        const int mt = tile / p.ntiles;
        const int nt = tile - mt * p.ntiles;
        n_m0 = mt * BM; n_n0 = nt * BN; n_g = g;
        rs_in = make_rsrc(p.in + (long)g * p.in_gs, p.pk_in_bytes);
        rs_w = make_rsrc(p.w + (long)g * p.w_gs + (long)s_k0 * BK, (p.Cout * p.Kpad - s_k0 * BK) * 4);
        n_k0 = s_k0;
        const int ohw = p.OH * p.OW;
        if constexpr (DUAL) rs_in2 = make_rsrc(p.in2 + (long)g * p.in2_gs, p.pk_in2_bytes);
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            const int m = n_m0 + lrow + RPP * i;
            if constexpr (DUAL) {        // rowoff: pixel m of `in`; iy0 (reused): the strided pixel of `in2`; OOB for the rows past M
                if (m < p.M) {
                    const int b = m / ohw;
                    const int rem = m - b * ohw;
                    const int oy = rem / p.OW;
                    const int ox = rem - oy * p.OW;
                    rowoff[i] = (unsigned)(m * p.in_cs) * 4u;
                    iy0[i] = (int)((unsigned)(((b * p.H2 + oy * p.stride2) * p.W2 + ox * p.stride2) * p.in2_cs) * 4u);
                } else {
                    rowoff[i] = OOB;
                    iy0[i] = (int)OOB;
                }
                ix0[i] = 0;
                continue;
            }
            if (m < p.M) {
                const int b = m / ohw;
                const int rem = m - b * ohw;
                const int oy = rem / p.OW;
                const int ox = rem - oy * p.OW;
                iy0[i] = oy * p.stride - p.pad;
                ix0[i] = ox * p.stride - p.pad;
                rowoff[i] = (unsigned)(((b * p.H + iy0[i]) * p.W + ix0[i]) * p.in_cs) * 4u;
            } else {
                iy0[i] = -(1 << 28);
                ix0[i] = 0;
                rowoff[i] = OOB;
            }
        }
#pragma unroll
        for (int i = 0; i < BL; ++i) {
            const int n = n_n0 + lrow + RPP * i;
            wrow[i] = (unsigned)((n < p.Cout ? n : 0) * p.Kpad + kq) * 4u;
        }
        ks = 0;
        if (p.kmode) {
            const int taps = p.kh * p.kw;
            const int cb = s_k0 / taps, tap = s_k0 - cb * taps;
            kc = cb * BK + kq;
            ky = tap / p.kw;
            kx = tap - ky * p.kw;
        } else {
            const int k = s_k0 * BK + kq;
            const int tap = k / p.Cin;
            kc = k - tap * p.Cin;
            ky = tap / p.kw;
            kx = tap - ky * p.kw;
        }
    };

    const bool one_by_one = p.kh == 1 && p.kw == 1 && p.pad == 0 && p.K == p.Kpad && !p.kmode;
    f32x4 ra[AL], rb[BL];
    auto gload = [&]() __attribute__((always_inline)) {
        if constexpr (DUAL) {
            const int kg = n_k0 + ks, nk1 = p.K1 / BK;
            const bool second = kg >= nk1;                                    // block-uniform
            const unsigned off = (unsigned)((second ? kg - nk1 : kg) * BK + kq) * 4u;
#pragma unroll
            for (int i = 0; i < AL; ++i)
                ra[i] = second ? buf_load4(rs_in2, (unsigned)iy0[i] + off) : buf_load4(rs_in, rowoff[i] + off);
            const unsigned woff = (unsigned)(ks * BK) * 4u;
#pragma unroll
            for (int i = 0; i < BL; ++i) rb[i] = buf_load4(rs_w, wrow[i] + woff);
            ++ks;
            return;
        }
        if (one_by_one) {      // 1x1, no padding, K a multiple of 32: every tap is inside the image, only the rows past M are not
            const unsigned off = (unsigned)kc * 4u;
#pragma unroll
            for (int i = 0; i < AL; ++i) ra[i] = buf_load4(rs_in, rowoff[i] + off);
            const unsigned woff = (unsigned)(ks * BK) * 4u;
#pragma unroll
            for (int i = 0; i < BL; ++i) rb[i] = buf_load4(rs_w, wrow[i] + woff);
            ++ks;
            kc += BK;
            return;
        }
        const bool kok = p.kmode || ky < p.kh;
        const int dy = ky * p.dil, dx = kx * p.dil;
        const unsigned off = (unsigned)((dy * p.W + dx) * p.in_cs + kc) * 4u;
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            const bool ok = kok && (unsigned)(iy0[i] + dy) < (unsigned)p.H && (unsigned)(ix0[i] + dx) < (unsigned)p.W;
            ra[i] = buf_load4(rs_in, ok ? rowoff[i] + off : OOB);
        }
        const unsigned woff = (unsigned)(ks * BK) * 4u;
#pragma unroll
        for (int i = 0; i < BL; ++i) rb[i] = buf_load4(rs_w, wrow[i] + woff);
        ++ks;
        if (p.kmode) {
            if (++kx == p.kw) {
                kx = 0;
                if (++ky == p.kh) { ky = 0; kc += BK; }
            }
        } else {
            kc += BK;
#pragma unroll
            for (int it = 0; it < BK / 8; ++it) {
                if (kc >= p.Cin) {
                    kc -= p.Cin;
                    if (++kx == p.kw) { kx = 0; ++ky; }
                }
            }
        }
    };
    auto lstore = [&]() __attribute__((always_inline)) {
        if constexpr (DT == 3) {
            auto split = [&](const f32x4 v, H16* dst, int plane_stride) __attribute__((always_inline)) {
                h16x4 p1, p2, p3;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = v[e];
                    const H16 x1 = (H16)x;
                    const float r1 = x - (float)x1;
                    const H16 x2 = (H16)r1;
                    const float r2 = r1 - (float)x2;
                    p1[e] = x1; p2[e] = x2; p3[e] = (H16)r2;
                }
                *reinterpret_cast<h16x4*>(dst) = p1;
                *reinterpret_cast<h16x4*>(dst + plane_stride) = p2;
                *reinterpret_cast<h16x4*>(dst + 2 * plane_stride) = p3;
            };
#pragma unroll
            for (int i = 0; i < AL; ++i) split(ra[i], &Ah[(lrow + RPP * i) * PITCH_H + kq], BM * PITCH_H);
#pragma unroll
            for (int i = 0; i < BL; ++i) split(rb[i], &Bh[(lrow + RPP * i) * PITCH_H + kq], BN * PITCH_H);
        } else {
            const int o = NBUF == 2 ? (buf ^ 1) * IMG_FLOATS : 0;
#pragma unroll
            for (int i = 0; i < AL; ++i) *reinterpret_cast<f32x4*>(&As[o + (lrow + RPP * i) * PITCH + kq]) = ra[i];
#pragma unroll
            for (int i = 0; i < BL; ++i) *reinterpret_cast<f32x4*>(&Bs[o + (lrow + RPP * i) * PITCH + kq]) = rb[i];
        }
    };

    const int wave = t >> 6, lane = t & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;
    f32x16 acc[TM][TN];
    // Two-level accumulation: `acc` is the MFMA chain of at most p.acc_chunk K-slices; `top` sums the chunks.  The rounding
    // error of an fp32 sum grows with the length of its chain: one chain over K = 128 ... 4608 left the network 2-4x
    // further from a float64 evaluation than the CPU reference's blocked reduction is (tests/fp64_anchor.py); with chunks
    // of 64 k the two are level, Winograd layers included (profiles/r03f_anchor_chunk.md).
    f32x16 top[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    auto fold = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                top[i][j] += acc[i][j];
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
            }
    };
    int fold_in = 0;               // K-slices until the next fold

    auto mma_slice = [&]() __attribute__((always_inline)) {
        if constexpr (DT == 3) {
#pragma unroll
            for (int k16 = 0; k16 < BK / 16; ++k16) {
                const H16* ap = &Ah[(wm * TM * 32 + r) * PITCH_H + 8 * h + k16 * 16];
                const H16* bp = &Bh[(wn * 32 + r) * PITCH_H + 8 * h + k16 * 16];
                h16x8 a[3][TM], b[3][TN];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[q][i] = *reinterpret_cast<const h16x8*>(ap + q * BM * PITCH_H + i * 32 * PITCH_H);
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[q][j] = *reinterpret_cast<const h16x8*>(bp + q * BN * PITCH_H + j * WN * 32 * PITCH_H);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        f32x16 c = acc[i][j];
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2][j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], c, 0, 0, 0);
                        acc[i][j] = c;
                    }
            }
        } else if constexpr (DT == 4) {
#pragma unroll
            for (int ks = 0; ks < BK / 8; ++ks) {           // 64 halfs per slice: four k16 steps
                const H16* ap = reinterpret_cast<const H16*>(&As[(wm * TM * 32 + r) * PITCH]) + 8 * h + ks * 16;
                const H16* bp = reinterpret_cast<const H16*>(&Bs[(wn * 32 + r) * PITCH]) + 8 * h + ks * 16;
                h16x8 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const h16x8*>(ap + i * 32 * 2 * PITCH);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const h16x8*>(bp + j * WN * 32 * 2 * PITCH);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        } else {
            // exact fp32: the chain of a K-slice starts from zero (SrcC = 0 of its first MFMA) and is added to `top` when the
            // slice is done - 16 MFMA steps (32 k) per chain, K / 32 additions above them
            const int o = NBUF == 2 ? buf * IMG_FLOATS : 0;
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k8 = 0; k8 < BK / 8; ++k8) {
                const float* ap = &As[o + (wm * TM * 32 + r) * PITCH + 4 * h];
                const float* bp = &Bs[o + (wn * 32 + r) * PITCH + 4 * h];
                f32x4 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(ap + i * 32 * PITCH + k8 * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(bp + j * WN * 32 * PITCH + k8 * 8);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, k8 == 0 ? zero : acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                    }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) top[i][j] += acc[i][j];
        }
    };

    // ---- epilogue of the segment (c_m0, c_n0, c_g): y = acc*scale + shift (+ residual) (ReLU), or the raw tile into
    // workspace slot c_slot.  In the MFMA result a lane holds ONE channel (column r) of 16 pixels; a 4x4 transposition
    // inside every lane quad (two DPP quad_perm exchanges per register) leaves it with 4 consecutive channels of the
    // pixel (r & 3) + 4h + 8g instead: 16-byte stores, 8 lanes per 128-byte line, 1 KB per wave instruction, no LDS and
    // no barrier - the K-slice image already holds the next segment's first slice.  (Storing 4 bytes per lane straight
    // from the MFMA layout, or staging through LDS, costs 5-7 us per tile: make STAMPS=1, tools/pk_stamps.py.)
    int c_m0 = 0, c_n0 = 0, c_g = 0, c_slot = 0;
    bool c_raw = false;
    int fin_b0 = 0, fin_g = 0;     // GroupNorm sums of the tile stored last: its first image, its group ...
    bool fin_gn = false;           // ... and whether there are any (not for a partial tile)
#ifdef PK_STAMPS
    const bool stamp_on = t == 0 && blockIdx.x < STAMP_BLOCKS;
    int stamp_tile = -1;
#endif
    auto epilogue = [&]() __attribute__((always_inline)) {
        const long tile_org = (long)c_m0 * p.out_cs + c_n0;
        // element size of the tensors this tile touches: the split-K workspace is fp32 always, out / res are fp16 in the fp16 data path
        const unsigned oes = (H16IO && !c_raw) ? 2u : 4u;
        const char* const out_base = reinterpret_cast<const char*>(p.out) + ((long)c_g * p.out_gs + tile_org) * (H16IO ? 2 : 4);
        const __amdgpu_buffer_rsrc_t rs_out =
            make_rsrc(c_raw ? reinterpret_cast<const void*>(p.ws + (long)c_slot * (BM * BN)) : reinterpret_cast<const void*>(out_base),
                      p.pk_debug == 1 ? 0 : c_raw ? BM * BN * 4 : (int)(((long)p.M * p.out_cs - tile_org) * oes));
        const unsigned out_cs4 = (c_raw ? BN : p.out_cs) * oes;
        const long res_org = (long)c_m0 * p.res_cs + c_n0;
        constexpr unsigned RES_ES = H16IO ? 2u : 4u;
        __amdgpu_buffer_rsrc_t rs_res = rs_out;
        if constexpr (RES)
            rs_res = make_rsrc(reinterpret_cast<const char*>(p.res) + ((long)c_g * p.res_gs + res_org) * RES_ES,
                               c_raw ? 0 : (int)(((long)p.M * p.res_cs - res_org) * RES_ES));
        const unsigned res_cs4 = p.res_cs * RES_ES;
        const bool affine = p.scale != nullptr && !c_raw;
        const float lo = (p.relu && !c_raw) ? 0.f : -__builtin_inff();     // ReLU as max(y, lo)
        const gf32x4_ptr scale = uniform_gptr(p.scale + c_g * p.ss_gs + c_n0);      // 16-byte aligned: c_n0 % 64 == 0, host checks the base
        const gf32x4_ptr shift = uniform_gptr(p.shift + c_g * p.ss_gs + c_n0);
        const bool q0 = lane & 1, q1 = lane & 2;
        const unsigned lrow = (r & 3) + 4 * h, lcol4 = (r >> 2) * 4;
        const bool gn = GN && !c_raw;
        int rows0 = BM, rows = BM;             // tile rows of image b0 (the rest belong to b0 + 1); valid rows
        if constexpr (GN) {
            fin_b0 = c_m0 / p.ohw;
            fin_g = c_g;
            fin_gn = gn;
            rows0 = (fin_b0 + 1) * p.ohw - c_m0;
            rows = p.M - c_m0;
        }
        PK_STAMP(5);
        // vmcnt counts loads and stores in issue order: a load issued after a tile's stores cannot be waited for without
        // waiting for those stores to reach memory (2-3 us).  So the affine parameters of all column blocks are fetched
        // before the first store, and the residual of tile t + 1 before the stores of tile t.
        f32x4 sc[TN], sh[TN];
        bool nok[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const unsigned col = (j * WN + wn) * 32 + lcol4;
            nok[j] = (int)(c_n0 + col) < p.Cout;
            sc[j] = f32x4{1.f, 1.f, 1.f, 1.f};
            sh[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (affine && nok[j]) {
                sc[j] = scale[col >> 2];
                sh[j] = shift[col >> 2];
            }
        }
        PK_STAMP(6);
        // ... and the residual one row group ahead of its use, issued before the stores of the group in hand
        auto res_off = [&](int s) __attribute__((always_inline)) -> unsigned {
            const int tt = s >> 2, g4 = s & 3, j = tt / TM, i = tt % TM;
            return nok[j] ? ((wm * TM + i) * 32 + lrow + g4 * 8) * res_cs4 + ((j * WN + wn) * 32 + lcol4) * RES_ES : OOB;
        };
        f32x4 rv_next = {0.f, 0.f, 0.f, 0.f};
        if constexpr (RES) rv_next = H16IO ? buf_load4h(rs_res, res_off(0)) : buf_load4(rs_res, res_off(0));
        double s0 = 0.0, q0s = 0.0, s1 = 0.0, q1s = 0.0;
#pragma unroll
        for (int tt = 0; tt < TM * TN; ++tt) {
            const int j = tt / TM, i = tt % TM;
            const unsigned vo = nok[j] ? ((wm * TM + i) * 32 + lrow) * out_cs4 + ((j * WN + wn) * 32 + lcol4) * oes : OOB;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 rv = rv_next;
                if constexpr (RES) {
                    if (tt * 4 + g4 + 1 < TM * TN * 4)
                        rv_next = H16IO ? buf_load4h(rs_res, res_off(tt * 4 + g4 + 1)) : buf_load4(rs_res, res_off(tt * 4 + g4 + 1));
                }
                const f32x16& fin = TWO ? top[i][j] : acc[i][j];
                float a0 = fin[4 * g4], a1 = fin[4 * g4 + 1], a2 = fin[4 * g4 + 2], a3 = fin[4 * g4 + 3];
                quad_transpose(a0, a1, a2, a3, q0, q1);
                f32x4 v = {a0, a1, a2, a3};
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    float y = fmaf(v[x], sc[j][x], sh[j][x]);
                    if constexpr (RES) y += rv[x];
                    v[x] = fmaxf(y, lo);
                }
                if (H16IO && !c_raw) {            // rounded once to fp16; the GroupNorm sums are of the stored values
                    const h16x4_t hv = buf_store4h(v, rs_out, vo + g4 * 8 * out_cs4);
                    v = f32x4{(float)hv.x, (float)hv.y, (float)hv.z, (float)hv.w};
                } else {
                    buf_store4(v, rs_out, vo + g4 * 8 * out_cs4);
                }
                if constexpr (GN) {
                    const int row = (wm * TM + i) * 32 + (int)lrow + g4 * 8;
                    if (gn && nok[j] && row < rows) {
                        const double a = (double)v[0] + (double)v[1] + (double)v[2] + (double)v[3];
                        const double b = (double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2] + (double)v[3] * v[3];
                        if (row < rows0) { s0 += a; q0s += b; } else { s1 += a; q1s += b; }
                    }
                }
            }
            if constexpr (GN) {
                if (i == TM - 1) {               // this lane's 4 channels of column block j are done: one norm group
                    if (gn && nok[j]) {
                        const int grp = (c_n0 + (j * WN + wn) * 32 + (int)lcol4) / p.gn_cpg;
                        atomicAdd(&gacc[grp * 2], s0);
                        atomicAdd(&gacc[grp * 2 + 1], q0s);
                        if (s1 != 0.0 || q1s != 0.0) {
                            atomicAdd(&gacc[64 + grp * 2], s1);
                            atomicAdd(&gacc[64 + grp * 2 + 1], q1s);
                        }
                    }
                    s0 = q0s = s1 = q1s = 0.0;
                }
            }
            PK_STAMP(7 + tt);
        }
    };

    // ---- one loop over the K-slices of all segments ----
    int nseg = 0, kt = 0;          // K-slices of the segment being multiplied, the slice in LDS
    bool cur = false;              // false only before the first segment
    for (;;) {
        const bool last = kt + 1 >= nseg;
        bool more = true;
        if (last) {
            more = next_seg();
            if (more) setup();
        }
        if (more) gload();                 // the next K-slice: of this segment, or the first one of the next segment
        if (cur) {
            mma_slice();
            if constexpr (DT == 3) {
                if (last || --fold_in == 0) {  // a chunk is complete (or the tile: the epilogue reads `top`)
                    fold();
                    fold_in = p.acc_chunk;
                }
            }
        }
        if constexpr (NBUF == 1) __syncthreads();   // every wave is done with the slice in LDS
        if (last) PK_STAMP(1);
        if (more) lstore();
        if (last) {
            PK_STAMP(2);
            if (cur) epilogue();
            PK_STAMP(3);
#ifdef PK_STAMPS
            if (t == 0 && blockIdx.x < 2048) g_pk_span[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memrealtime();
            if (t == 0 && blockIdx.x < 2048 && !cur) g_pk_span[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
#endif
            if (!more) {
                if constexpr (GN) {
                    if (fin_gn) {
                        __syncthreads();
                        if (t < 128) {
                            const double v = gacc[t];
                            const int b = fin_b0 + (t >> 6);
                            if (v != 0.0 && b < p.B) atomicAdd(&p.gn_sum[(((long)fin_g * p.B + b) * p.gn_groups) * 2 + (t & 63)], v);
                        }
                    }
                }
                break;
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        if constexpr (TWO) top[i][j][e] = 0.f;            // (`acc` is zero after the fold that ended the last tile)
                        else acc[i][j][e] = 0.f;
                    }
            fold_in = p.acc_chunk;         // 0: never reaches zero by decrements - one chain over the whole K
            c_m0 = n_m0; c_n0 = n_n0; c_g = n_g; c_slot = s_slot;
            nseg = s_k1 - s_k0;
            c_raw = nseg != nkt;
            kt = 0;
            cur = true;
#ifdef PK_STAMPS
            ++stamp_tile;
#endif
        } else {
            ++kt;
        }
        __syncthreads();                   // the next slice is in LDS (NBUF == 2: and every wave is done with the one just multiplied)
        if constexpr (NBUF == 2) buf ^= 1;
        if constexpr (GN) {
            if (last && fin_gn) {          // every wave has added its sums: out to the layer's accumulators, LDS cleared for the next tile
                if (t < 128) {
                    const double v = gacc[t];
                    gacc[t] = 0.0;
                    const int b = fin_b0 + (t >> 6);
                    if (v != 0.0 && b < p.B) atomicAdd(&p.gn_sum[(((long)fin_g * p.B + b) * p.gn_groups) * 2 + (t & 63)], v);
                }
                fin_gn = false;
            }
        }
        if (kt == 0) PK_STAMP(0);
        if (kt == 1) PK_STAMP(4);
    }
}

// Sums the pieces of the remainder tiles that were computed by more than one block, in block order, and applies the
// fused epilogue.  grid = (PX, 8): block (j, xcd) owns remainder tile j of that XCD's run.
template <int BM, int BN>
__global__ __launch_bounds__(256) void pk_fixup_kernel(const ConvP p, int P) {
    const int xcd = blockIdx.y, j = blockIdx.x;
    const int nkt = p.Kpad / BK;
    const PkPlan pl = pk_plan(p.pk_T, P, xcd, nkt, p.pk_min);
    if (j >= pl.rem) return;
    const int ua = j * nkt;
    const int bf = pk_owner(pl, ua), bl = pk_owner(pl, ua + nkt - 1);
    if (bf == bl) return;                               // computed whole by block bf
    const int gt = pl.c0 + pl.R * pl.PX + j;
    const int g = gt / p.pk_tpg;
    const int tile = gt - g * p.pk_tpg;
    const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const float* __restrict__ scale = p.scale ? p.scale + g * p.ss_gs : nullptr;
    const float* __restrict__ shift = p.shift ? p.shift + g * p.ss_gs : nullptr;
    const float* __restrict__ prelu = p.prelu ? p.prelu + g * p.ss_gs : nullptr;
    const bool h16 = p.es == 2;                        // fp16 data path: out / res are fp16 (the partial tiles are fp32 always)
    const float* __restrict__ res = p.res ? (h16 ? reinterpret_cast<const float*>(reinterpret_cast<const _Float16*>(p.res) + (long)g * p.res_gs)
                                                 : p.res + (long)g * p.res_gs)
                                          : nullptr;
    float* __restrict__ out = h16 ? reinterpret_cast<float*>(reinterpret_cast<_Float16*>(p.out) + (long)g * p.out_gs) : p.out + (long)g * p.out_gs;
    // first piece: the tail of block bf's share (slot 1 unless that share starts exactly here); the others start a share
    const float* first = p.ws + (long)(2 * (bf * 8 + xcd) + (pk_u0(pl, bf) < ua ? 1 : 0)) * (BM * BN);
    __shared__ double gacc[2 * 32 * 2];
    const bool gn = p.gn_sum != nullptr;
    const int t = threadIdx.x;
    int b0 = 0, m_next = 0;
    if (gn) {
        if (t < 128) gacc[t] = 0.0;
        b0 = m0 / p.ohw;
        m_next = (b0 + 1) * p.ohw;
        __syncthreads();
    }
    constexpr int CPR = BN / 4;
    for (int c = t; c < BM * CPR; c += 256) {
        const int row = c / CPR, q = (c - row * CPR) * 4;
        const int m = m0 + row, n = n0 + q;
        if (m >= p.M || n >= p.Cout) continue;
        const long e = (long)row * BN + q;
        // the affine parameters, the slopes and the residual are requested with the pieces, not one after the other where they are used
        float4 sc, sh, sl, rv;
        h16x4_t rh;
        if (scale) { sc = *reinterpret_cast<const float4*>(scale + n); sh = *reinterpret_cast<const float4*>(shift + n); }
        if (prelu) sl = *reinterpret_cast<const float4*>(prelu + n);
        if (res) {
            if (h16) rh = *reinterpret_cast<const h16x4_t*>(reinterpret_cast<const _Float16*>(res) + (long)m * p.res_cs + n);
            else rv = *reinterpret_cast<const float4*>(res + (long)m * p.res_cs + n);
        }
        float4 v = *reinterpret_cast<const float4*>(first + e);
        for (int b = bf + 1; b <= bl; ++b) {
            const float4 w = *reinterpret_cast<const float4*>(p.ws + (long)(2 * (b * 8 + xcd)) * (BM * BN) + e);
            v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
        if (scale) {
            v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y);
            v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
        }
        if (res) {
            if (h16) rv = make_float4((float)rh.x, (float)rh.y, (float)rh.z, (float)rh.w);
            v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
        }
        if (p.relu) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        if (prelu) {
            v.x = v.x > 0.f ? v.x : v.x * sl.x; v.y = v.y > 0.f ? v.y : v.y * sl.y;
            v.z = v.z > 0.f ? v.z : v.z * sl.z; v.w = v.w > 0.f ? v.w : v.w * sl.w;
        }
        if (h16) {
            const h16x4_t hv = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
            *reinterpret_cast<h16x4_t*>(reinterpret_cast<_Float16*>(out) + (long)m * p.out_cs + n) = hv;
            v = make_float4((float)hv.x, (float)hv.y, (float)hv.z, (float)hv.w);
        } else {
            *reinterpret_cast<float4*>(out + (long)m * p.out_cs + n) = v;
        }
        if (gn) {
            const double a = (double)v.x + (double)v.y + (double)v.z + (double)v.w;
            const double b = (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
            const int slot = (m < m_next ? 0 : 64) + (n / p.gn_cpg) * 2;
            atomicAdd(&gacc[slot], a);
            atomicAdd(&gacc[slot + 1], b);
        }
    }
    if (gn) {
        __syncthreads();
        if (t < 128) {
            const double v = gacc[t];
            const int b = b0 + (t >> 6);
            if (v != 0.0 && b < p.B) atomicAdd(&p.gn_sum[(((long)g * p.B + b) * p.gn_groups) * 2 + (t & 63)], v);
        }
    }
}

}  // namespace


// Eligibility beyond conv_persistent_ok() is decided by the caller (conv_igemm.hip: run<>): no skipped filter rows, a
// workspace of 2 * P tiles.  p.mtiles / p.ntiles / p.vec_out are filled in.  Returns 0 on success.
template <int BM, int BN, int WM, int WN>
int launch_conv_persistent(ConvP p, int G, int bpc, hipStream_t st) {
    const int nk = p.Kpad / BK;
    p.pk_tpg = p.mtiles * p.ntiles;
    p.pk_T = p.pk_tpg * G;
    int P = 256 * bpc;
    // The remainder (tiles beyond whole rounds of P) is shared out in K only when a share is long enough to pay for
    // the two partial tiles a block then writes and the fix-up pass re-reads (key 14: shortest K, default 32 slices),
    // or when the launch cannot give every block a tile; otherwise the remainder tiles are computed whole.
    p.pk_min = (nk >= tune().persist_min_nk || p.pk_T < P) ? 4 : 0;
    if (p.pk_min) {
        // never more blocks than there are shares of pk_min K-slices
        const long shares = (long)p.pk_T * nk / p.pk_min;
        if (shares < P) P = (int)(shares < 8 ? 8 : shares / 8 * 8);
    }
    const dim3 block(WM * WN * 64);
    p.pk_in_bytes = (int)((long)p.B * p.H * p.W * p.in_cs * 4);
    p.pk_debug = tune().persist_debug;
    const int epi = p.in2 ? 3 : p.gn_sum ? 2 : p.res ? 1 : 0;       // host: never two of them (conv_persistent_ok)
    if (p.es == 2) {
        if constexpr (BM == 128 && BN == 128) {          // the fp16 data path goes persistent on 128x128 tiles only
            if (epi == 3) hipLaunchKernelGGL((conv_igemm_pk<BM, BN, WM, WN, 4, 3>), dim3(P), block, 0, st, p);
            else if (epi == 2) hipLaunchKernelGGL((conv_igemm_pk<BM, BN, WM, WN, 4, 2>), dim3(P), block, 0, st, p);
            else if (epi == 1) hipLaunchKernelGGL((conv_igemm_pk<BM, BN, WM, WN, 4, 1>), dim3(P), block, 0, st, p);
            else hipLaunchKernelGGL((conv_igemm_pk<BM, BN, WM, WN, 4, 0>), dim3(P), block, 0, st, p);
        } else {
            return fail("persistent convolution: the fp16 data path has 128x128 tiles only");
        }
    } else if (p.bf16 == 3) {
        if (epi == 3) hipLaunchKernelGGL((conv_igemm_pk<BM, BN, WM, WN, 3, 3>), dim3(P), block, 0, st, p);
        else if (epi == 2) hipLaunchKernelGGL((conv_igemm_pk<BM, BN, WM, WN, 3, 2>), dim3(P), block, 0, st, p);
        else if (epi == 1) hipLaunchKernelGGL((conv_igemm_pk<BM, BN, WM, WN, 3, 1>), dim3(P), block, 0, st, p);
        else hipLaunchKernelGGL((conv_igemm_pk<BM, BN, WM, WN, 3, 0>), dim3(P), block, 0, st, p);
    } else {
        if (epi == 3) hipLaunchKernelGGL((conv_igemm_pk<BM, BN, WM, WN, 0, 3>), dim3(P), block, 0, st, p);
        else if (epi == 2) hipLaunchKernelGGL((conv_igemm_pk<BM, BN, WM, WN, 0, 2>), dim3(P), block, 0, st, p);
        else if (epi == 1) hipLaunchKernelGGL((conv_igemm_pk<BM, BN, WM, WN, 0, 1>), dim3(P), block, 0, st, p);
        else hipLaunchKernelGGL((conv_igemm_pk<BM, BN, WM, WN, 0, 0>), dim3(P), block, 0, st, p);
    }
    // any tile shared between blocks?
    int max_rem = 0;
    bool shared = false;
    for (int x = 0; x < 8; ++x) {
        const PkPlan pl = pk_plan(p.pk_T, P, x, nk, p.pk_min);
        if (pl.rem > max_rem) max_rem = pl.rem;
        if (pl.rem && pl.U % nk == 0 && pl.PXs == pl.rem) continue;    // one whole tile per share
        if (pl.rem) shared = true;
    }
    if (shared) hipLaunchKernelGGL((pk_fixup_kernel<BM, BN>), dim3(max_rem, 8), dim3(256), 0, st, p, P);
    QB_CHECK(hipGetLastError());
    return 0;
}

#ifdef PK_STAMPS
int pk_read_span(unsigned long long* dst, int n) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_pk_span), sizeof(unsigned long long) * (size_t)(n < 8192 ? n : 8192), 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
int pk_read_stamps(unsigned long long* dst, int n) {
    const size_t bytes = sizeof(unsigned long long) * (size_t)(n < STAMP_BLOCKS * STAMP_TILES * STAMP_N ? n : STAMP_BLOCKS * STAMP_TILES * STAMP_N);
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_pk_stamps), bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
#endif

size_t conv_persistent_ws_floats(int BM, int BN, int bpc) { return (size_t)2 * 256 * bpc * BM * BN; }

// Host view of the work list of block `bid` of a persistent launch (the arithmetic conv_igemm_pk runs on the device), for
// the CPU tests: up to `cap` segments as (tile, first K-slice, end K-slice, workspace slot or -1 for a whole tile).
int conv_persistent_segments(int T, int P, int nk, int min_slices, int bid, int* out4, int cap) {
    const int xcd = bid & 7, l = bid >> 3;
    const PkPlan pl = pk_plan(T, P, xcd, nk, min_slices);
    int n = 0;
    for (int round = 0; round < pl.R && n < cap; ++round, ++n) {
        out4[4 * n] = pl.c0 + round * pl.PX + l; out4[4 * n + 1] = 0; out4[4 * n + 2] = nk; out4[4 * n + 3] = -1;
    }
    if (l < pl.PXs) {
        int u = pk_u0(pl, l);
        const int u_first = u, u_end = pk_u0(pl, l + 1);
        while (u < u_end && n < cap) {
            const int j = u / nk, k0 = u - j * nk, k1 = (nk < k0 + (u_end - u)) ? nk : k0 + (u_end - u);
            out4[4 * n] = pl.c0 + pl.R * pl.PX + j; out4[4 * n + 1] = k0; out4[4 * n + 2] = k1;
            out4[4 * n + 3] = (k1 - k0 == nk) ? -1 : 2 * bid + (u == u_first ? 0 : 1);
            u += k1 - k0;
            ++n;
        }
    }
    return n;
}
// ... and of the fix-up pass: the slots summed, in order, for remainder tile j of XCD run `xcd` (0 = computed whole)
int conv_persistent_fixup(int T, int P, int nk, int min_slices, int xcd, int j, int* tile, int* slots, int cap) {
    const PkPlan pl = pk_plan(T, P, xcd, nk, min_slices);
    if (j >= pl.rem) return -1;
    *tile = pl.c0 + pl.R * pl.PX + j;
    const int ua = j * nk;
    const int bf = pk_owner(pl, ua), bl = pk_owner(pl, ua + nk - 1);
    if (bf == bl) return 0;
    int n = 0;
    for (int b = bf; b <= bl && n < cap; ++b, ++n) slots[n] = 2 * (b * 8 + xcd) + ((b == bf && pk_u0(pl, bf) < ua) ? 1 : 0);
    return n;
}

// what the persistent kernel covers: fp32 / bf16x3 arithmetic, affine + ReLU epilogues with either a residual or GroupNorm
// sums, 16-byte-aligned channel counts, every view below 2 GiB (32-bit buffer offsets)
bool conv_persistent_ok(const ConvP& p) {
    const long lim = (long)1 << 31;
    // (fp16 data path, es == 2: in_cs / Kpad are in 4-byte units by now - launch_conv - so the operand sizes come out in bytes)
    return (p.bf16 == 0 || p.bf16 == 3 || (p.bf16 == 2 && p.es == 2)) && p.vec_out && !p.prelu && !(p.gn_sum && p.res) &&
           !(p.in2 && (p.res || p.gn_sum)) && (long)p.B * p.H * p.W * p.in_cs * 4 < lim && (long)p.M * p.out_cs * 4 < lim &&
           (!p.res || (long)p.M * p.res_cs * 4 < lim) && (long)p.Cout * p.Kpad * 4 < lim;
}

// conv3 + projection shortcut of a bottleneck as one 1x1 GEMM over two inputs (kernel: EPI 3).  Returns 1 when the
// launch is not covered (the caller then runs the two convolutions separately), 0 on success, -1 on a launch error.
int launch_conv_dual(ConvP p, int G, hipStream_t st) {
    if (!tune().persist || !p.in2 || !p.ws) return 1;
    p.acc_chunk = tune().acc_chunk;
    if (p.es == 2) {        // fp16 data path: K-side quantities in 4-byte units (two halfs), as launch_conv hands them over
        if (p.bf16 != 2 || p.Cin % 8 || p.in_cs % 8 || p.in2_cs % 8 || p.Kpad % 64 || p.K1 % 64 || (p.in_gs & 7) || (p.in2_gs & 7) || (p.w_gs & 1))
            return 1;
        p.Cin /= 2; p.in_cs /= 2; p.in2_cs /= 2; p.K /= 2; p.Kpad /= 2; p.K1 /= 2; p.in_gs /= 2; p.in2_gs /= 2; p.w_gs /= 2;
        if (p.kh == 1 && p.kw == 1 && p.pad == 0 && p.stride == 1 && !p.kmode && p.K == p.Kpad) {      // 256 x 256 tiles, LDS-DMA pipeline (conv_h8.hip)
            p.ohw = p.OH * p.OW;
            const int rc = launch_conv_h8(p, G, st);
            if (rc != 1) return rc;
        }
    } else {
        p.es = 4;
        if (p.bf16 == 3 && p.w3 && p.kh == 1 && p.kw == 1 && p.pad == 0 && p.stride == 1 && !p.kmode && p.K == p.Kpad) {      // bf16x3: 256 x 128 tiles, LDS-DMA pipeline (conv_x8.hip)
            p.ohw = p.OH * p.OW;
            const int rc = launch_conv_x8(p, G, st);
            if (rc != 1) return rc;
        }
    }
    if (p.kh != 1 || p.kw != 1 || p.pad != 0 || p.stride != 1 || p.kmode || p.K1 % BK || p.K1 <= 0 || p.K1 >= p.Kpad || p.K != p.Kpad ||
        p.Kpad % BK || p.in_cs % 4 || p.in2_cs % 4 || ((uintptr_t)p.in & 15) || ((uintptr_t)p.in2 & 15) || (p.in_gs & 3) || (p.in2_gs & 3))
        return 1;
    p.vec_out = (p.Cout % 4 == 0) && (p.out_cs % 4 == 0) && (((uintptr_t)p.out & 15) == 0) && (p.out_gs % 4 == 0) &&
                (!p.scale || ((((uintptr_t)p.scale & 15) == 0) && (p.ss_gs % 4 == 0)));
    const long in2_bytes = (long)p.B * p.H2 * p.W2 * p.in2_cs * 4;
    if (!conv_persistent_ok(p) || in2_bytes >= ((long)1 << 31)) return 1;
    p.pk_in2_bytes = (int)in2_bytes;
    const char* tag = p.tag ? p.tag : "conv_gemm";
    if (p.bf16 == 3 && p.Kpad / BK <= 8) {                 // as the separate launches: short K gains nothing from bf16x3
        p.bf16 = 0;
        tag = "conv_gemm_f32pipe";                         // profiled apart: this launch runs on the fp32 matrix pipe
    }
    const long tiles128 = (long)((p.M + 127) / 128) * ((p.Cout + 127) / 128) * G;
    // small batches: the two separate launches (64 x 64 tiles, split-K model).  With the launch rule of round 6 (key 42) the separate launches win up to
    // ~500 tiles: res3.0 at batch 1 (304 tiles) 85 us as one dual-input launch against 34 + 24 us as two (profiles/r15_final_b1_layers_480x640.md)
    // (twice key 15's threshold: 512 tiles by default; the tests lower key 15 to 0 to send a few tiles through this kernel)
    if (tiles128 < (tune().small_n_64 ? 2 : 1) * (long)tune().persist_min_tiles) return 1;
    const bool big = tiles128 >= 192 && p.Cout > 64;
    if (p.es == 2 && !big) return 1;                       // the fp16 data path has the 128x128 persistent kernel only
    const int bpc = big ? (p.es == 2 ? 3 : 2) : 5;         // pk_occupancy()
    const int BMs = big ? 128 : 64;
    if (p.ws_floats < conv_persistent_ws_floats(BMs, BMs, bpc)) return 1;
    p.mtiles = (p.M + BMs - 1) / BMs;
    p.ntiles = (p.Cout + BMs - 1) / BMs;
    const double out_bytes = 4.0 * G * (double)p.M * p.Cout;
    const double bytes = 4.0 * G * ((double)p.M * p.Kpad + (double)p.Cout * p.Kpad) + out_bytes;
    ProfScope prof(tag, bytes, 2.0 * G * (double)p.M * p.Kpad * p.Cout, st);
    const int rc = big ? launch_conv_persistent<128, 128, 2, 2>(p, G, bpc, st) : launch_conv_persistent<64, 64, 2, 2>(p, G, bpc, st);
    return rc ? -1 : 0;
}

template int launch_conv_persistent<64, 64, 2, 2>(ConvP, int, int, hipStream_t);
template int launch_conv_persistent<128, 128, 2, 2>(ConvP, int, int, hipStream_t);
template int launch_conv_persistent<256, 32, 4, 1>(ConvP, int, int, hipStream_t);

}  // namespace quber
