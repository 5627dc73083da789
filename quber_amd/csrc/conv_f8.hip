// Exact-fp32 1x1 convolutions / grouped GEMMs on 256 x 128 tiles with an LDS-DMA operand pipeline: the fp32 sibling of
// conv_h8.hip, for the launches whose tile count fills the chip - the P^2 = 36 position GEMMs of the wide Winograd layers
// (csrc/winograd.hip: M = tiles, K = Cin, N = Cout) and the wide 1x1 layers of the encoder (reference:
// maskrefiner/modeling/backbone/resnet.py:395-449 bottleneck conv1 / conv3, :472-485 the fusion reductions).
//
// Arithmetic = conv_igemm.hip's exact fp32 mode, bit for bit: v_mfma_f32_32x32x2_f32, lane (r, h) supplies k = 8 ks + 4 h + s to
// MFMA step s of group ks, the chain of a K-slice (32 k) starts from zero and is added to a second register set when the slice
// is done (two-level accumulation, tests/fp64_anchor.py).  Swapping the operand roles (weights as the row operand, so that a
// lane's accumulator registers are consecutive channels) transposes the MFMA's output and changes no sum.
//
// Structure: block = 8 waves (4 pixel quarters x 2 channel halves, 64 x 64 per wave), K-slice = 32 floats = 128-byte rows, two
// K-slice images of 48 KB, both operands by `buffer_load_dwordx4 ... lds` one slice ahead (the fp32 matrix pipe needs 8 192
// cycles per slice and SIMD: one slice of lookahead hides any latency), four phases per slice {4 fragment reads + DMA | barrier |
// 16 MFMAs | barrier} with the two waves of a SIMD half a phase apart, persistent tiles with the pipeline running across the tile
// boundary, 16-byte stores straight from the accumulators.  conv_igemm.hip's kernels hold MFMA busy 0.66-0.70 on these launches
// (profiles/r10_final_conv_mfma_busy_dtype0.md): its loader's vector instructions and the accumulator fold run on the wave that
// should be multiplying.
#include <algorithm>
#include <string>

#include "common.h"

namespace quber {

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int F8_BM = 256, F8_BN = 128;     // pixels x channels per tile
constexpr int F8_KB = 128;                  // bytes of one K-slice row (32 floats)
constexpr int F8_QBASE = F8_BM * F8_KB;     // channel rows start here inside a K-slice image
constexpr int F8_SLOT = (F8_BM + F8_BN) * F8_KB;
constexpr int F8_SS = 2048;                 // bytes of one tile's [scale (256 floats read, 128 used) | shift] image
constexpr int F8_OOB = (int)0x80000000;

__device__ __forceinline__ unsigned f8_div(unsigned n, unsigned m, unsigned s) {       // conv_h8.hip: h8_div
    const unsigned t = __umulhi(n, m);
    return (t + ((n - t) >> (s & 1u))) >> (s >> 1);
}

// DMA source state of a thread for one tile: 4 pixel rows (pieces 4 w .. 4 w + 3 of the 32) and 2 channel rows (pieces 2 w, 2 w + 1)
__device__ __forceinline__ void f8_tile_state(const ConvP& p, int tile, int wave, int lane, int (&aoff)[4], int (&boff)[2], int& m0, int& n0, int& g) {
    g = (int)f8_div((unsigned)tile, p.dv_m[2], p.dv_s[2]);             // tile / tiles per group
    const int rem = tile - g * p.pk_tpg;
    const int mt = (int)f8_div((unsigned)rem, p.dv_m[3], p.dv_s[3]);   // rem / channel tiles
    const int nt = rem - mt * p.ntiles;
    m0 = mt * F8_BM;
    n0 = nt * F8_BN;
    const int prow = lane >> 3, pc = lane & 7;
    const int gin = g * (int)p.in_gs * 4, gw = g * (int)p.w_gs * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int R = 32 * wave + 8 * j + prow;
        const int m = m0 + R;
        const int chunk = pc ^ ((R >> 1) & 7);
        int pix = m;                                  // stride 1: output pixel m reads input pixel m
        if (p.stride != 1) {
            const int b = (int)f8_div((unsigned)m, p.dv_m[0], p.dv_s[0]);       // m / ohw
            const int r2 = m - b * p.ohw;
            const int oy = (int)f8_div((unsigned)r2, p.dv_m[1], p.dv_s[1]);     // r2 / OW
            const int ox = r2 - oy * p.OW;
            pix = (b * p.H + oy * p.stride) * p.W + ox * p.stride;
        }
        aoff[j] = m < p.M ? gin + pix * p.in_cs * 4 + chunk * 16 : F8_OOB;     // rows past M: zeros
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int R = 16 * wave + 8 * j + prow;
        const int n = n0 + R;
        const int chunk = pc ^ ((R >> 1) & 7);
        boff[j] = n < p.Cout ? gw + n * p.Kpad * 4 + chunk * 16 : F8_OOB;
    }
}

template <bool AFFINE, bool RES, bool GN>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_f8_kernel(const ConvP p) {
    // ONE shared object (conv_h8.hip): [2 K-slice images][GroupNorm sums, f64 [2 images][32 groups][2]][2 scale | shift images]
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * F8_SLOT + 1024 + 2 * F8_SS];
    constexpr int SSBASE = 2 * F8_SLOT + 1024;

    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    const int wp = wave & 3, wq = wave >> 2;      // pixel quarter, channel half; waves w and w + 4 share a SIMD
    const int r = lane & 31, h = lane >> 5;
    const int nk = p.Kpad / 32;

    int tile, tile_step, tile_end;                // this block's tiles (XCD-aware, as conv_h8.hip)
    {
        const int bid = blockIdx.x, nblk = gridDim.x, T = p.pk_T;
        const int xcd = bid & 7, q = T >> 3, rr = T & 7;
        const int start = xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q;
        tile_end = start + q + (xcd < rr ? 1 : 0);
        tile_step = (nblk >> 3) + (xcd < (nblk & 7) ? 1 : 0);
        tile = start + (bid >> 3);
    }

    int aoff[4], aoffN[4], boff[2], boffN[2];
    int m0, n0, g, m0N = 0, n0N = 0, gN = 0;
    f8_tile_state(p, tile, wave, lane, aoff, boff, m0, n0, g);
#pragma unroll
    for (int j = 0; j < 4; ++j) aoffN[j] = F8_OOB;
    boffN[0] = boffN[1] = F8_OOB;
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.lean_in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.pk_in_bytes, 0x00020000);

    // K-slice kq of this tile (kq == nk: slice 0 of the block's next tile; past the last tile every offset is out of range)
    auto issue_p = [&](int half, int slot, int kq) __attribute__((always_inline)) {
        const bool nxt = kq >= nk;
        const int soff = nxt ? 0 : kq * F8_KB;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int j = 2 * half + jj;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lds_ptr_t)(smem + slot * F8_SLOT + (4 * wave + j) * 1024), 16, nxt ? aoffN[j] : aoff[j], soff, 0, 0);
        }
    };
    auto issue_q = [&](int slot, int kq) __attribute__((always_inline)) {
        const bool nxt = kq >= nk;
        const int soff = nxt ? 0 : kq * F8_KB;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, (lds_ptr_t)(smem + slot * F8_SLOT + F8_QBASE + (2 * wave + j) * 1024), 16, nxt ? boffN[j] : boff[j], soff, 0, 0);
    };
    const __amdgpu_buffer_rsrc_t rss = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.scale), 0, AFFINE ? p.h8_ss_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.shift), 0, AFFINE ? p.h8_ss_bytes : 0, 0x00020000);
    auto issue_ss = [&](int buf, int tg, int tn0) __attribute__((always_inline)) {
        if constexpr (AFFINE) {
            if (wave == 0) {
                const int off = (tg * p.ss_gs + tn0) * 4 + lane * 16;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rss, (lds_ptr_t)(smem + SSBASE + buf * F8_SS), 16, off, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsh, (lds_ptr_t)(smem + SSBASE + buf * F8_SS + 1024), 16, off, 0, 0, 0);
            }
        }
    };
    int ssb = 0;

    // fragment addresses: row R of an operand image at R * 128 B, logical chunk 2 ks + h at physical chunk (2 ks + h) ^ ((R >> 1) & 7)
    int paddr[4], qaddr[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int sw = ((2 * ks + h) ^ ((r >> 1) & 7)) << 4;
        paddr[ks] = (64 * wp + r) * F8_KB + sw;
        qaddr[ks] = F8_QBASE + (64 * wq + r) * F8_KB + sw;
    }

    f32x16 acc[2][2], top[2][2];                   // [channel tile][pixel tile]
    f32x4 qf[2], pf[2];
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // ---- prologue: K-slice 0 ----
    issue_ss(0, g, n0);
    issue_p(0, 0, 0); issue_p(1, 0, 0); issue_q(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int gk = 0;                                    // K-slices consumed so far: slice gk lives in image gk & 1

#define F8_READ(KS)                                                                                                       \
    _Pragma("unroll") for (int c = 0; c < 2; ++c) qf[c] = *reinterpret_cast<const f32x4*>(smem + sbase + qaddr[KS] + c * 32 * F8_KB); \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) pf[i] = *reinterpret_cast<const f32x4*>(smem + sbase + paddr[KS] + i * 32 * F8_KB);
#define F8_MMA(KS)                                                                                                        \
    __builtin_amdgcn_s_barrier();                                                                                          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                     \
    __builtin_amdgcn_s_setprio(1);                                                                                         \
    _Pragma("unroll") for (int c = 0; c < 2; ++c)                                                                          \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                    \
            acc[c][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[c].x, pf[i].x, (KS) == 0 ? zero16 : acc[c][i], 0, 0, 0);   \
            acc[c][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[c].y, pf[i].y, acc[c][i], 0, 0, 0);                        \
            acc[c][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[c].z, pf[i].z, acc[c][i], 0, 0, 0);                        \
            acc[c][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[c].w, pf[i].w, acc[c][i], 0, 0, 0);                        \
        }                                                                                                                  \
    __builtin_amdgcn_s_setprio(0);                                                                                         \
    __builtin_amdgcn_s_barrier();

    for (;;) {
        const bool has_next = tile + tile_step < tile_end;
        if (has_next) {
            f8_tile_state(p, tile + tile_step, wave, lane, aoffN, boffN, m0N, n0N, gN);
            issue_ss(ssb ^ 1, gN, n0N);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) aoffN[j] = F8_OOB;
            boffN[0] = boffN[1] = F8_OOB;
        }
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < 2; ++i) top[c][i] = zero16;
        if (wq == 1) __builtin_amdgcn_s_barrier();      // stagger: waves 4-7 run half a phase behind their SIMD partners

        for (int kt = 0; kt < nk; ++kt, ++gk) {
            const int s = gk & 1;
            const int sbase = s * F8_SLOT;
            // phase ks: the fragments of k-step ks (2 channel + 2 pixel tiles), then 16 MFMAs; the DMAs of the next slice go to the
            // other image in phases 0-2 (its last readers retired their reads before phase 3's first barrier of the slice before)
            F8_READ(0)
            issue_p(0, s ^ 1, kt + 1);
            F8_MMA(0)
            F8_READ(1)
            issue_p(1, s ^ 1, kt + 1);
            F8_MMA(1)
            F8_READ(2)
            issue_q(s ^ 1, kt + 1);
            F8_MMA(2)
            F8_READ(3)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // the next slice has landed; this image's last reads are done
            F8_MMA(3)
            // two-level accumulation: the slice's chain into the running sums (conv_igemm.hip `top`)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int i = 0; i < 2; ++i) top[c][i] += acc[c][i];
        }
        if (wq == 0) __builtin_amdgcn_s_barrier();      // the two halves level again

        // ---- epilogue: y = top * scale + shift (+ residual) (ReLU); a lane's registers 4 j .. 4 j + 3 of a tile are the 4
        //      consecutive channels 8 j + 4 h .. of pixel r: 16-byte stores (conv_h8.hip for the LDS / barrier rules) ----
        {
            const int rows = min(p.M - m0, F8_BM);
            const long org = (long)g * p.out_gs + (long)m0 * p.out_cs + n0;
            const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.out + org, 0, ((rows - 1) * p.out_cs + min(p.Cout - n0, F8_BN)) * 4, 0x00020000);
            const long rorg = (long)g * p.res_gs + (long)m0 * p.res_cs + n0;
            const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res) + (RES ? rorg : 0), 0,
                                                                                 RES ? ((rows - 1) * p.res_cs + min(p.Cout - n0, F8_BN)) * 4 : 0, 0x00020000);
            const float lo = p.relu ? 0.f : -__builtin_inff();
            const int b0 = GN ? (int)f8_div((unsigned)m0, p.dv_m[0], p.dv_s[0]) : 0;
            const int m_next = (b0 + 1) * p.ohw;
            const bool plain = m0 + F8_BM <= p.M && m0 + F8_BM <= m_next;      // one image, whole rows
            const unsigned gacc_b = 2 * F8_SLOT;
            if constexpr (GN) {
                if (t < 128) {
                    const unsigned long long z = 0;
                    asm volatile("ds_write_b64 %0, %1" :: "v"(gacc_b + t * 8), "v"(z) : "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            int r_e = r, h_e = h;                      // opaque copies: offsets recomputed per tile, not kept in registers across the K loop
            asm volatile("" : "+v"(r_e), "+v"(h_e));
            const int prow0 = 64 * wp + r_e;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int nl = 64 * wq + 32 * c + 8 * j + 4 * h_e;       // first of the lane's 4 channels inside the tile
                    const bool colok = n0 + nl < p.Cout;                     // (Cout is a multiple of 4)
                    const int obase = colok ? (prow0 * p.out_cs + nl) * 4 : F8_OOB;
                    u32x4 rbuf[2];
                    if constexpr (RES) {
                        const int rbase = colok ? (prow0 * p.res_cs + nl) * 4 : F8_OOB;
#pragma unroll
                        for (int i = 0; i < 2; ++i) rbuf[i] = __builtin_amdgcn_raw_buffer_load_b128(rsr, rbase + i * 32 * p.res_cs * 4, 0, 0);
                    }
                    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (AFFINE) {
                        const unsigned ad = SSBASE + ssb * F8_SS + nl * 4;
                        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)" : "=&v"(sc), "=&v"(sh) : "v"(ad) : "memory");
                    }
                    double gs = 0.0, gq = 0.0, gs1 = 0.0, gq1 = 0.0;         // GroupNorm sums (fp64, as conv_igemm.hip) of image b0 / b0 + 1 over this lane's 4 channels
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = top[c][i][4 * j + e];
                            if constexpr (AFFINE) v[e] = fmaf(v[e], sc[e], sh[e]);
                        }
                        if constexpr (RES) {
                            const f32x4 rv = __builtin_bit_cast(f32x4, rbuf[i]);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += rv[e];
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], lo);
                        const f32x4 o = {v[0], v[1], v[2], v[3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rso, obase + i * 32 * p.out_cs * 4, 0, 0);
                        if constexpr (GN) {
                            const double a = (double)v[0] + (double)v[1] + (double)v[2] + (double)v[3];
                            const double b = (double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2] + (double)v[3] * v[3];
                            if (plain) {
                                gs += a; gq += b;
                            } else {
                                const int m = m0 + prow0 + 32 * i;
                                const bool in0 = m < p.M && m < m_next, in1 = m < p.M && m >= m_next;
                                gs += in0 ? a : 0.0; gq += in0 ? b : 0.0; gs1 += in1 ? a : 0.0; gq1 += in1 ? b : 0.0;
                            }
                        }
                    }
                    if constexpr (GN) {
                        // the 16 lanes of a DPP row hold the same channels: four row_shr adds leave the row's sum in its lane 15
                        auto row_sum = [](double x) __attribute__((always_inline)) {
#define F8_DPP_STEP(CTL)                                                                                                   \
    {                                                                                                                      \
        const unsigned long long u = __builtin_bit_cast(unsigned long long, x);                                            \
        const unsigned lo32 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTL, 0xf, 0xf, true);             \
        const unsigned hi32 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTL, 0xf, 0xf, true);     \
        x += __builtin_bit_cast(double, ((unsigned long long)hi32 << 32) | lo32);                                          \
    }
                            F8_DPP_STEP(0x111) F8_DPP_STEP(0x112) F8_DPP_STEP(0x114) F8_DPP_STEP(0x118)
#undef F8_DPP_STEP
                            return x;
                        };
                        auto lds_add = [&](unsigned slot, double dx) __attribute__((always_inline)) {
                            asm volatile("ds_add_f64 %0, %1" :: "v"(gacc_b + slot * 8), "v"(dx) : "memory");
                        };
                        const int grp = (int)f8_div((unsigned)(n0 + nl), p.dv_m[4], p.dv_s[4]);     // / channels per group (a multiple of 4)
                        gs = row_sum(gs); gq = row_sum(gq);
                        if (!plain) { gs1 = row_sum(gs1); gq1 = row_sum(gq1); }
                        if ((r_e & 15) == 15 && colok) {
                            lds_add(grp * 2, gs); lds_add(grp * 2 + 1, gq);
                            if (!plain) { lds_add(64 + grp * 2, gs1); lds_add(64 + grp * 2 + 1, gq1); }
                        }
                    }
                }
            }
            if constexpr (GN) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (t < 128) {
                    double v;
                    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(gacc_b + t * 8) : "memory");
                    const int b = b0 + (t >> 6);
                    if (v != 0.0 && b < p.B) atomicAdd(&p.gn_sum[(((long)g * p.B + b) * p.gn_groups) * 2 + (t & 63)], v);
                }
            }
        }
        if (!has_next) break;
        tile += tile_step;
#pragma unroll
        for (int j = 0; j < 4; ++j) aoff[j] = aoffN[j];
        boff[0] = boffN[0]; boff[1] = boffN[1];
        m0 = m0N; n0 = n0N; g = gN;
        ssb ^= 1;
    }
#undef F8_READ
#undef F8_MMA
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the out-of-range tail DMAs still write (zeros) into the images
}

static void f8_magic(unsigned d, unsigned& m, unsigned& s) {      // conv_h8.hip: h8_magic
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    m = (unsigned)((((1ull << l) - d) << 32) / d + 1);
    s = ((l > 0 ? l - 1 : 0) << 1) | (l > 0 ? 1u : 0u);
}

}  // namespace

// exact fp32, 1x1 / pad 0 (any stride), K a multiple of 32, one input, 16-byte epilogue accesses, enough tiles to fill the chip
// several times (a block owns whole tiles: no K sharing of a ragged last round - those launches keep conv_igemm.hip's persistent kernel)
// returns 0 = launched, 1 = not covered, -1 = error
int launch_conv_f8(ConvP p, int G, hipStream_t st) {
    if (!tune().f8 || p.es != 4 || p.bf16 != 0 || p.in2 || p.prelu || p.acc_chunk == 0) return 1;
    if (p.kh != 1 || p.kw != 1 || p.pad != 0 || p.Cin % 32 || p.K != p.Kpad || p.Cin != p.K || p.Kpad / 32 < 2) return 1;
    if ((p.scale == nullptr) != (p.shift == nullptr)) return 1;
    const long in_bytes = ((long)p.B * p.H * p.W * p.in_cs) * 4, w_bytes = (long)p.Cout * p.Kpad * 4;
    const long in_all = in_bytes + (long)(G - 1) * p.in_gs * 4, w_all = w_bytes + (long)(G - 1) * p.w_gs * 4;
    if (in_all >= 0x7fffff00L || w_all >= 0x7fffff00L) return 1;
    const bool vec4 = p.Cout % 4 == 0 && p.out_cs % 4 == 0 && p.out_gs % 4 == 0 && (((uintptr_t)p.out & 15) == 0) && p.in_cs % 4 == 0 && (p.in_gs & 3) == 0 &&
                      (!p.res || (p.res_cs % 4 == 0 && p.res_gs % 4 == 0 && (((uintptr_t)p.res & 15) == 0))) &&
                      (!p.scale || (p.ss_gs % 4 == 0 && (((uintptr_t)p.scale & 15) == 0) && (((uintptr_t)p.shift & 15) == 0)));
    if (!vec4 || (long)p.out_cs * F8_BM * 4 >= 0x7fffff00L) return 1;
    p.mtiles = (p.M + F8_BM - 1) / F8_BM;
    p.ntiles = (p.Cout + F8_BN - 1) / F8_BN;
    const long tiles = (long)p.mtiles * p.ntiles * G;
    if (tiles > 0x3fffffff) return 1;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return fail("conv_f8: cannot query the device");
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    // rounds of tiles: at least f8_min_rounds, and the ragged last round at least 3/4 full unless there are many rounds
    const long rounds = (tiles + cus - 1) / cus;
    if (tune().f8 < 2) {         // (key 33 = 2: every covered launch, whatever its shape and tile count - the tests)
        // short K (the HBM-bound residual 1x1 layers live on conv_igemm.hip's 64 x 64 tiles, 7 blocks per CU) and narrow outputs
        // (half a tile of zero rows) are not for this kernel: measured slower in the network (profiles/r11_f8.md)
        if (p.Kpad / 32 < 8 || p.Cout < F8_BN) return 1;
        if (tiles < (long)tune().f8_min_rounds * cus) return 1;
        if (rounds < 8 && tiles % cus != 0 && tiles % cus < (3 * cus) / 4) return 1;
    }
    p.lean_in_bytes = (int)in_all;
    p.pk_in_bytes = (int)w_all;
    p.pk_T = (int)tiles;
    p.pk_tpg = p.mtiles * p.ntiles;
    f8_magic((unsigned)p.ohw, p.dv_m[0], p.dv_s[0]);
    f8_magic((unsigned)p.OW, p.dv_m[1], p.dv_s[1]);
    f8_magic((unsigned)p.pk_tpg, p.dv_m[2], p.dv_s[2]);
    f8_magic((unsigned)p.ntiles, p.dv_m[3], p.dv_s[3]);
    f8_magic((unsigned)(p.gn_sum && p.gn_cpg > 0 ? p.gn_cpg : 1), p.dv_m[4], p.dv_s[4]);
    p.h8_ss_bytes = ((G - 1) * p.ss_gs + p.Cout) * 4;
    const bool gn_sep = p.gn_sum && !(p.gn_cpg % 4 == 0 && p.gn_groups <= 32 && p.ohw >= F8_BM);
    double* const gn_sum = p.gn_sum;
    if (gn_sep) p.gn_sum = nullptr;
    {
        const double out_bytes = 4.0 * G * (double)p.M * p.Cout;
        const double conv_bytes = 4.0 * G * ((double)p.B * p.H * p.W * p.Cin + (double)p.Cout * p.K) + out_bytes * (p.res ? 2.0 : 1.0);
        const double conv_flops = 2.0 * G * (double)p.M * p.K * p.Cout;
        // (stages of their own: the bench prices them with the GEMM family, the tests see that the kernel is on the path)
        const char* tag = !p.tag ? "conv_gemm_f8" : std::string(p.tag) == "wino_gemm" ? "wino_gemm_f8" : p.tag;
        ProfScope prof(tag, conv_bytes, conv_flops, st);
        const dim3 grid((unsigned)std::min<long>(tiles, cus)), block(512);
        const int variant = (p.scale ? 4 : 0) | (p.res ? 2 : 0) | (p.gn_sum ? 1 : 0);
        switch (variant) {
            case 0: hipLaunchKernelGGL((conv_f8_kernel<false, false, false>), grid, block, 0, st, p); break;
            case 1: hipLaunchKernelGGL((conv_f8_kernel<false, false, true>), grid, block, 0, st, p); break;
            case 2: hipLaunchKernelGGL((conv_f8_kernel<false, true, false>), grid, block, 0, st, p); break;
            case 3: hipLaunchKernelGGL((conv_f8_kernel<false, true, true>), grid, block, 0, st, p); break;
            case 4: hipLaunchKernelGGL((conv_f8_kernel<true, false, false>), grid, block, 0, st, p); break;
            case 5: hipLaunchKernelGGL((conv_f8_kernel<true, false, true>), grid, block, 0, st, p); break;
            case 6: hipLaunchKernelGGL((conv_f8_kernel<true, true, false>), grid, block, 0, st, p); break;
            default: hipLaunchKernelGGL((conv_f8_kernel<true, true, true>), grid, block, 0, st, p); break;
        }
    }
    QB_CHECK(hipGetLastError());
    if (gn_sep) {
        View o;
        o.p = p.out; o.B = p.B; o.H = p.OH; o.W = p.OW; o.C = p.Cout; o.cs = p.out_cs; o.gs = p.out_gs; o.es = 4;
        return launch_gn_stats(o, p.B, G, p.gn_groups, gn_sum, st, false);
    }
    return 0;
}

}  // namespace quber
