// Implicit-GEMM convolution on the CDNA4 matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32).
//
// Covers every convolution of the QuBER refiner: 1x1 / 3x3, stride 1-2,
// dilation 1-18, NHWC activations, with the per-channel affine (FrozenBN / folded BN / bias),
// residual add and ReLU of the reference's Conv2d wrapper fused into the epilogue
// (reference: detectron2 Conv2d + BottleneckBlock as used by maskrefiner/modeling/backbone/resnet.py:37-63,
//  441-447 and maskrefiner/modeling/mask_refiner/model.py:386-403, 431-451).
//
// GEMM view: M = B*OH*OW output pixels, N = Cout, K = kh*kw*Cin.  A rows are gathered on the fly
// (im2col never materialised), B rows are the packed weights [Cout][Kpad].
// Block = 256 threads = 4 waves; each wave owns a (TM x TN) grid of 32x32 MFMA tiles.
// K is consumed 32 at a time through LDS ([row][k], pitch 36 floats so that the 16-byte fragment
// reads of a wave hit 64 distinct banks); the next K-slice is prefetched into registers while
// the current one feeds the matrix pipe.
//
// MFMA operand mapping: lane (r = lane&31, h = lane>>5) supplies A[row r][kk = h] and
// B[kk = h][col r].  The order of k inside a K-slice is free as long as A and B agree, so each lane
// fetches 4 consecutive k with one ds_read_b128 and feeds them to 4 successive MFMAs:
// MFMA step s of group ks multiplies k = ks*8 + 4*h + s.
#include "common.h"

namespace quber {

using f32x16 = __attribute__((ext_vector_type(16))) float;
// native vector type for register staging: HIP's float4 is a struct whose copies can lower to memcpy and
// keep the staging arrays in scratch
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int BK = 32;
constexpr int PITCH = 36;

// Main-loop variants that were measured and rejected (profiles/r01c_conv_variants.md): double-buffered LDS with one
// barrier per K-slice, a a per-block s_setprio stagger, and 256x128 / 128x256 tiles (8 waves, or 4 waves of 128x64) were all 1-25 % slower
// than this single-buffer, register-prefetch loop at 3 blocks per CU.
template <int BM, int BN, int WM, int WN, bool DIAG = false>
__global__ __launch_bounds__(WM * WN * 64, (BM == 128 && BN == 128) ? 3 : 1) void conv_igemm_f32(const ConvP p) {
    constexpr int NTH = WM * WN * 64;
    constexpr int RPP = NTH / 8;      // tile rows covered by one pass of the loader
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int AL = BM / RPP;      // float4 loads per thread per K-slice (A)
    constexpr int BL = BN / RPP;      // (B)
    constexpr int NBUF = 1;
    static_assert(BM % RPP == 0 && BN % RPP == 0, "loader geometry");
    constexpr int SW = WN * 32;        // columns staged per epilogue pass
    constexpr int SP = SW + 4;         // staging pitch (floats)
    constexpr int KSLICE_FLOATS = NBUF * (BM + BN) * PITCH;
    constexpr int SMEM_FLOATS = KSLICE_FLOATS > BM * SP ? KSLICE_FLOATS : BM * SP;
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
    float* const As = smem;
    float* const Bs = smem + NBUF * BM * PITCH;

    const int t = threadIdx.x;
    const int g = blockIdx.z;
    unsigned long long blk_rt0 = 0;
    if (DIAG) blk_rt0 = __builtin_amdgcn_s_memrealtime();

    // XCD-aware tile order: blocks b and b+8 share an XCD (and its L2); give every XCD one
    // contiguous run of tiles, n-tile fastest, so neighbouring tiles reuse the same input rows.
    int tile;
    {
        const int bid = blockIdx.x, nblk = gridDim.x;
        const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        if (p.order == 1) tile = bid;
        tile += p.tile_begin;   // a launch may cover only a run of the tile sequence (full rounds / tail)
    }
    int nt = tile % p.ntiles;
    int mt = tile / p.ntiles;
    if (p.order == 2) {
        // groups of 8 m-tiles x all n-tiles, m fastest inside the group
        const int gsz = 8 * p.ntiles;
        const int grp = tile / gsz, within = tile - grp * gsz;
        const int rows = min(8, p.mtiles - grp * 8);
        mt = grp * 8 + within % rows;
        nt = within / rows;
    }
    const int m0 = mt * BM, n0 = nt * BN;

    const float* __restrict__ in = p.in + (long)g * p.in_gs;
    const float* __restrict__ wt = p.w + (long)g * p.w_gs;

    // ---- loader state ----
    // Each thread owns one 16-byte column (4 consecutive k) of AL A-rows and BL B-rows.  Its position inside the
    // filter window (ky, kx, channel c) advances by BK per K-slice and is tracked incrementally (no divisions in
    // the loop).  Loads are unconditional: an out-of-image tap reads a valid dummy address and is zeroed when
    // the slice is written to LDS, so the compiler can issue all loads of a slice back to back.
    const int kq = (t & 7) * 4;
    const int lrow = t >> 3;
    int iy0[AL], ix0[AL];
    const float* rowp[AL];
#pragma unroll
    for (int i = 0; i < AL; ++i) {
        const int m = m0 + lrow + RPP * i;
        if (m < p.M) {
            const int ohw = p.OH * p.OW;
            const int b = m / ohw;
            const int rem = m - b * ohw;
            const int oy = rem / p.OW;
            const int ox = rem - oy * p.OW;
            iy0[i] = oy * p.stride - p.pad;
            ix0[i] = ox * p.stride - p.pad;
            rowp[i] = in + ((long)b * p.H * p.W + (long)iy0[i] * p.W + ix0[i]) * p.in_cs;
        } else {
            iy0[i] = -(1 << 28);
            ix0[i] = 0;
            rowp[i] = in;
        }
    }
    const float* wrow[BL];
#pragma unroll
    for (int i = 0; i < BL; ++i) {
        const int n = n0 + lrow + RPP * i;
        wrow[i] = wt + (long)(n < p.Cout ? n : 0) * p.Kpad + kq;   // columns >= Cout are never stored
    }
    // split-K: blockIdx.y owns the K-slices [k_begin, k_end) and writes a raw partial tile (summed by
    // splitk_reduce_kernel); used when a layer has too few tiles to fill the chip (small batches)
    const int nk_all = p.Kpad / BK;
    const int k_begin = (int)((long)blockIdx.y * nk_all / p.ksplit);
    const int k_end = (int)((long)(blockIdx.y + 1) * nk_all / p.ksplit);
    int kc, kx, ky;
    if (p.kmode) {
        const int taps = p.kh * p.kw;
        const int cb = k_begin / taps, tap = k_begin - cb * taps;
        kc = cb * BK + kq;
        ky = tap / p.kw;
        kx = tap - ky * p.kw;
    } else {
        const int k = k_begin * BK + kq;
        const int tap = k / p.Cin;
        kc = k - tap * p.Cin;
        ky = tap / p.kw;
        kx = tap - ky * p.kw;
    }

    f32x4 ra[AL], rb[BL];
    bool aok[AL];
    auto gload = [&](int kt) __attribute__((always_inline)) {
        const bool kok = p.kmode ? kc < p.Cin : ky < p.kh;   // false for the redundant load after the last slice
        const int dy = ky * p.dil, dx = kx * p.dil;
        const long off = ((long)dy * p.W + dx) * p.in_cs + kc;
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            const bool ok = kok && (unsigned)(iy0[i] + dy) < (unsigned)p.H && (unsigned)(ix0[i] + dx) < (unsigned)p.W;
            aok[i] = ok;
            ra[i] = *reinterpret_cast<const f32x4*>(ok ? rowp[i] + off : in);
        }
#pragma unroll
        for (int i = 0; i < BL; ++i) rb[i] = *reinterpret_cast<const f32x4*>(wrow[i] + kt * BK);
        // advance to the next K-slice.  Both K orders are evaluated and selected (no branches: the loop body stays
        // one basic block, and LLVM cannot merge the branch tails into a pointer phi that would push kc/ky to scratch).
        // slice-major (kmode 1, k = (c/32, tap, c%32)): the taps of one 32-channel slice are consecutive K-slices, so
        // the 9 shifted reads of a 3x3 window hit the same cache lines back to back instead of 8+ slices apart.
        const int wx = (kx + 1 == p.kw) ? 1 : 0;
        const int s_kx = wx ? 0 : kx + 1;
        const int wy = (ky + wx == p.kh) ? 1 : 0;
        const int s_ky = wy ? 0 : ky + wx;
        const int s_kc = kc + (wy ? BK : 0);
        // tap-major (kmode 0, k = (tap, c)); Cin >= 8, so a K-slice crosses at most BK/8 taps
        int t_kc = kc + BK, t_kx = kx, t_ky = ky;
#pragma unroll
        for (int it = 0; it < BK / 8; ++it) {
            const int ge = t_kc >= p.Cin ? 1 : 0;
            t_kc -= ge ? p.Cin : 0;
            const int w = (t_kx + ge == p.kw) ? 1 : 0;
            t_kx = w ? 0 : t_kx + ge;
            t_ky += w;
        }
        const bool sm = p.kmode != 0;
        kc = sm ? s_kc : t_kc;
        kx = sm ? s_kx : t_kx;
        ky = sm ? s_ky : t_ky;
    };
    auto lstore = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < AL; ++i)
            *reinterpret_cast<f32x4*>(&As[buf * BM * PITCH + (lrow + RPP * i) * PITCH + kq]) =
                aok[i] ? ra[i] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < BL; ++i)
            *reinterpret_cast<f32x4*>(&Bs[buf * BN * PITCH + (lrow + RPP * i) * PITCH + kq]) = rb[i];
    };

    const int wave = t >> 6, lane = t & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto mma = [&](int buf, int ks) __attribute__((always_inline)) {
        const float* ap = &As[buf * BM * PITCH + (wm * TM * 32 + r) * PITCH + 4 * h];
        const float* bp = &Bs[buf * BN * PITCH + (wn * 32 + r) * PITCH + 4 * h];
        f32x4 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(ap + i * 32 * PITCH + ks * 8);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(bp + j * WN * 32 * PITCH + ks * 8);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
            }
    };
    gload(k_begin);
    lstore(0);
    __syncthreads();
    // The loop body is one basic block (the last iteration re-loads its own slice instead of branching), so the
    // scheduler may spread the next slice's address arithmetic and global loads between the MFMAs of the current
    // one: an MFMA occupies the matrix pipe for 64 cycles but the issue port only briefly, and VALU work
    // interleaved there is free, while the same work in front of the MFMA block leaves the pipe idle.
    unsigned long long tsum[5] = {0, 0, 0, 0, 0};
    unsigned long long clk0 = 0, rt0 = 0;
    if (DIAG) {
        clk0 = __builtin_amdgcn_s_memtime();
        rt0 = __builtin_amdgcn_s_memrealtime();
    }
    auto stamp = [&]() __attribute__((always_inline)) -> unsigned long long {
        unsigned long long tt;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt)::"memory");
        __builtin_amdgcn_sched_barrier(0);
        return tt;
    };
    for (int kt = k_begin; kt < k_end; ++kt) {
        unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
        if (DIAG) t0 = stamp();
        gload(kt + 1 < k_end ? kt + 1 : kt);
        if (DIAG) t1 = stamp();
#pragma unroll
        for (int ks = 0; ks < BK / 8; ++ks) mma(0, ks);
        if (!DIAG) {
#pragma unroll
            for (int i = 0; i < (BK / 8) * TM * TN * 4; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);   // a few VALU / SALU
                __builtin_amdgcn_sched_group_barrier(0x120, 1, 0);   // at most one VMEM / LDS read
            }
        }
        if (DIAG) t2 = stamp();
        __syncthreads();
        if (DIAG) t3 = stamp();
        lstore(0);
        if (DIAG) t4 = stamp();
        __syncthreads();
        if (DIAG) {
            t5 = stamp();
            tsum[0] += t1 - t0; tsum[1] += t2 - t1; tsum[2] += t3 - t2; tsum[3] += t4 - t3; tsum[4] += t5 - t4;
        }
    }
    if (DIAG && p.dbg && (t & 63) == 0) {
        unsigned long long* d = p.dbg + ((long)blockIdx.x * 4 + wave) * 12;
        for (int i = 0; i < 5; ++i) d[i] = tsum[i];
        d[5] = __builtin_amdgcn_s_memtime() - clk0;        // shader cycles spent in the K loop
        d[6] = __builtin_amdgcn_s_memrealtime() - rt0;     // the same interval in 100 MHz ticks
        d[7] = blk_rt0;                                    // block start (100 MHz ticks)
        d[8] = rt0;                                        // K loop start
        d[10] = ((unsigned long long)__builtin_amdgcn_s_getreg(63492) << 32) | __builtin_amdgcn_s_getreg(63508);  // HW_ID | XCC_ID
    }

    // ---- epilogue: y = acc*scale + shift (+ residual) (ReLU) ----
    // The accumulators are transposed through LDS one 32-column tile per wave at a time, so that global
    // stores (and residual loads) are 16 bytes per lane over WN*32 consecutive channels of a pixel.
    const bool partial = p.ksplit > 1;
    // partial tiles of the rows [m_begin, M): slab s of group g starts at ((s*G + g) * (M - m_begin)) * Cout
    float* __restrict__ out = partial ? p.ws + (((long)blockIdx.y * gridDim.z + g) * (long)(p.M - p.m_begin) - p.m_begin) * p.Cout
                                      : p.out + (long)g * p.out_gs;
    const int out_cs = partial ? p.Cout : p.out_cs;
    const float* __restrict__ res = (p.res && !partial) ? p.res + (long)g * p.res_gs : nullptr;
    const float* __restrict__ scale = (p.scale && !partial) ? p.scale + g * p.ss_gs : nullptr;
    const float* __restrict__ shift = (p.shift && !partial) ? p.shift + g * p.ss_gs : nullptr;
    const bool relu = p.relu && !partial;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                smem[row * SP + wn * 32 + r] = acc[i][j][e];
            }
        __syncthreads();
        const int nb = n0 + j * SW;    // first channel of this pass
        if (p.vec_out) {
            constexpr int CPR = SW / 4;  // float4 chunks per row
#pragma unroll
            for (int c = t; c < BM * CPR; c += NTH) {
                const int row = c / CPR, q = (c - row * CPR) * 4;
                const int m = m0 + row, n = nb + q;
                if (m < p.M && n < p.Cout) {
                    float4 v = *reinterpret_cast<const float4*>(&smem[row * SP + q]);
                    if (scale) {
                        const float4 sc = *reinterpret_cast<const float4*>(scale + n);
                        const float4 sh = *reinterpret_cast<const float4*>(shift + n);
                        v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y);
                        v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
                    }
                    if (res) {
                        const float4 rv = *reinterpret_cast<const float4*>(res + (long)m * p.res_cs + n);
                        v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                    }
                    if (relu) {
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    }
                    *reinterpret_cast<float4*>(out + (long)m * out_cs + n) = v;
                }
            }
        } else {
            for (int c = t; c < BM * SW; c += NTH) {
                const int row = c / SW, q = c - row * SW;
                const int m = m0 + row, n = nb + q;
                if (m < p.M && n < p.Cout) {
                    float v = smem[row * SP + q];
                    if (scale) v = fmaf(v, scale[n], shift[n]);
                    if (res) v += res[(long)m * p.res_cs + n];
                    if (relu) v = fmaxf(v, 0.f);
                    out[(long)m * out_cs + n] = v;
                }
            }
        }
        if (j + 1 < TN) __syncthreads();
    }
    if (DIAG && p.dbg && (t & 63) == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        p.dbg[((long)blockIdx.x * 4 + wave) * 12 + 9] = __builtin_amdgcn_s_memrealtime();   // block end (stores drained)
    }
}

// sums the split-K partial tiles in a fixed order (deterministic) and applies the fused epilogue
__global__ void splitk_reduce_kernel(const ConvP p, int S, int G) {
    const int g = blockIdx.y;
    const long MN = (long)(p.M - p.m_begin) * p.Cout;
    const float* __restrict__ scale = p.scale ? p.scale + g * p.ss_gs : nullptr;
    const float* __restrict__ shift = p.shift ? p.shift + g * p.ss_gs : nullptr;
    const float* __restrict__ res = p.res ? p.res + (long)g * p.res_gs : nullptr;
    float* __restrict__ out = p.out + (long)g * p.out_gs;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < MN; i += (long)gridDim.x * blockDim.x) {
        const long mr = i / p.Cout;
        const int n = (int)(i - mr * p.Cout);
        const long m = mr + p.m_begin;
        float v = 0.f;
        for (int s = 0; s < S; ++s) v += p.ws[((long)s * G + g) * MN + i];
        if (scale) v = fmaf(v, scale[n], shift[n]);
        if (res) v += res[m * p.res_cs + n];
        if (p.relu) v = fmaxf(v, 0.f);
        out[m * p.out_cs + n] = v;
    }
}

static int g_order = 0;
static float* g_splitk_ws = nullptr;
static size_t g_splitk_floats = 0;
void set_conv_splitk_workspace(float* ws, size_t floats) { g_splitk_ws = ws; g_splitk_floats = floats; }
static unsigned long long* g_dbg = nullptr;
void set_conv_dbg(void* p) { g_dbg = (unsigned long long*)p; }
void set_conv_order(int v) { g_order = v; }

template <int BM, int BN, int WM, int WN>
static int run(ConvP p, int G, hipStream_t st) {
    p.mtiles = (p.M + BM - 1) / BM;
    p.ntiles = (p.Cout + BN - 1) / BN;
    p.order = g_order;
    p.vec_out = (p.Cout % 4 == 0) && (p.out_cs % 4 == 0) && (((uintptr_t)p.out & 15) == 0) && (p.out_gs % 4 == 0) &&
                (!p.res || ((p.res_cs % 4 == 0) && (((uintptr_t)p.res & 15) == 0) && (p.res_gs % 4 == 0))) &&
                (!p.scale || ((((uintptr_t)p.scale & 15) == 0) && (p.ss_gs % 4 == 0)));
    // Work distribution.  `slots` blocks are resident at once (256 CUs x blocks per CU for this tile shape).  With
    // fewer than half a round of tiles (small batches) K is split so that about one round of blocks exists; the partial
    // tiles are combined in a fixed order by splitk_reduce_kernel (deterministic).  Splitting only the ragged last
    // round of larger launches was measured too (profiles/r01g_tail_split.md): +1..3 % on some layers, -2 % on
    // others, because a lone block on a CU already runs ~1.5x faster than one of three - not kept.
    constexpr int BPC = (BM == 64) ? 7 : (BM * BN == 256 * 64 ? 2 : 3);   // resident blocks per CU (registers / LDS)
    const long slots = 256L * BPC;
    const long T = (long)p.mtiles * p.ntiles;          // tiles per group
    const long blocks_all = T * G;
    const int nk = p.Kpad / BK;
    auto split_for = [&](long nblocks, long rows) {
        if (!g_splitk_ws || nk < 16) return 1;
        int S = (int)((slots + nblocks - 1) / nblocks);
        if (S > nk / 8) S = nk / 8;
        if (S > 16) S = 16;
        while (S > 1 && (size_t)S * G * rows * p.Cout > g_splitk_floats) --S;
        return S < 2 ? 1 : S;
    };
    auto launch = [&](long tile_begin, long ntile, int S, int m_begin) {
        ConvP q = p;
        q.tile_begin = (int)tile_begin;
        q.ksplit = S;
        q.m_begin = m_begin;
        q.ws = g_splitk_ws;
        hipLaunchKernelGGL((conv_igemm_f32<BM, BN, WM, WN>), dim3((unsigned)ntile, S, G), dim3(WM * WN * 64), 0, st, q);
        if (S > 1) {
            const long MN = (long)(q.M - m_begin) * q.Cout;
            int blocks = (int)((MN + 255) / 256);
            if (blocks > 2048) blocks = 2048;
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks, G), dim3(256), 0, st, q, S, G);
        }
    };
    launch(0, T, blocks_all * 2 < slots ? split_for(blocks_all, p.M) : 1, 0);
    QB_CHECK(hipGetLastError());
    return 0;
}

int launch_conv(const ConvP& p, int G, hipStream_t st) {
    if (p.Cin % 4 || p.in_cs % 4 || p.Kpad % BK || p.K > p.Kpad)
        return fail("conv: Cin / channel stride must be multiples of 4 and Kpad a multiple of 32");
    if (((uintptr_t)p.in & 15) || ((uintptr_t)p.w & 15) || (p.in_gs & 3) || (p.w_gs & 3))
        return fail("conv: operands must be 16-byte aligned");
    if (p.M <= 0 || p.Cout <= 0) return fail("conv: empty problem");
    if (p.kmode && (p.Cin % BK || p.K != p.Kpad)) return fail("conv: slice-major weights need Cin % 32 == 0");
    if ((p.scale == nullptr) != (p.shift == nullptr)) return fail("conv: scale and shift go together");
    if (p.Cout <= 32) return run<256, 32, 4, 1>(p, G, st);
    if (p.Cout <= 64) return run<256, 64, 4, 1>(p, G, st);
    const long tiles128 = (long)((p.M + 127) / 128) * ((p.Cout + 127) / 128) * G;
    if (tiles128 < 512) return run<64, 64, 2, 2>(p, G, st);
    if (g_dbg) {
        ConvP q = p;
        q.dbg = g_dbg;
        q.mtiles = (q.M + 127) / 128; q.ntiles = (q.Cout + 127) / 128; q.order = g_order; q.vec_out = (q.Cout % 4 == 0) && (q.out_cs % 4 == 0) && !q.scale && !q.res; q.ksplit = 1; q.tile_begin = 0; q.m_begin = 0;
        hipLaunchKernelGGL((conv_igemm_f32<128, 128, 2, 2, true>), dim3(q.mtiles * q.ntiles, 1, G), dim3(256), 0, st, q);
        return 0;
    }
    return run<128, 128, 2, 2>(p, G, st);
}

}  // namespace quber
