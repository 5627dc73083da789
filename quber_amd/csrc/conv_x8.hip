// bf16x3 (quber_config.compute_dtype 3: fp32 operands as three bf16 terms, six partial products per multiply, fp32 accumulation -
// the fp32-equivalent mode that meets the exact mode's float64-anchor bars) for the wide 1x1 launches and the Winograd position
// GEMMs, on 256 x 128 tiles with the LDS-DMA operand pipeline of conv_h8.hip.
//
// conv_igemm.hip's bf16x3 kernels split BOTH operands as a K-slice is written to LDS and keep three bf16 planes per operand there:
// LDS-bound, MFMA busy 0.40 (DECISIONS.md section 4).  Here
//   * the WEIGHTS are split once, at plan time, into three bf16 planes in HBM (launch_split_bf16x3: x1 = bf16(x), x2 = bf16(x - x1),
//     x3 = bf16(x - x1 - x2), round to nearest even - the same terms the kernels derive) and arrive by DMA as planes of 64-byte rows;
//   * the ACTIVATIONS arrive by DMA as the fp32 they are (128-byte rows: 32 k) and are split in registers, fragment by fragment, by the
//     wave that is NOT multiplying: the two waves of a SIMD run half a phase apart, so the ~70 vector instructions of a k-step's split
//     execute beside the partner's 24 MFMAs;
//   * a K-slice (32 k) is two phases {reads + split + DMA | barrier | 2 x 2 tiles x 6 MFMAs | barrier}; v_mfma_f32_32x32x16_bf16 with the
//     weights as row operand; the six partial products in conv_igemm.hip's order (smallest first), the MFMA chain folded into a
//     second register set every p.acc_chunk slices (two-level accumulation).
// Reference layers: maskrefiner/modeling/backbone/resnet.py:395-449 (bottleneck 1x1s), :472-485 (fusion reductions), and through
// csrc/winograd.hip the 3x3 fusion / res4 / res5 / ASPP convolutions.
#include <algorithm>
#include <string>

#include "common.h"

namespace quber {

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int X8_BM = 256, X8_BN = 128;     // pixels x channels per tile
constexpr int X8_PB = 128;                  // bytes of a pixel row of a K-slice (32 floats)
constexpr int X8_QB = 64;                   // bytes of a channel row of one plane of a K-slice (32 bf16)
constexpr int X8_QBASE = X8_BM * X8_PB;     // the three weight planes start here inside a K-slice image
constexpr int X8_QPLANE = X8_BN * X8_QB;
constexpr int X8_SLOT = X8_QBASE + 3 * X8_QPLANE;
constexpr int X8_SS = 2048;
constexpr int X8_OOB = (int)0x80000000;

__device__ __forceinline__ unsigned x8_div(unsigned n, unsigned m, unsigned s) {       // conv_h8.hip: h8_div
    const unsigned t = __umulhi(n, m);
    return (t + ((n - t) >> (s & 1u))) >> (s >> 1);
}

// DMA source state of a thread for one tile: 4 pixel rows (pieces 4 w .. 4 w + 3 of the 32: 8 rows x 128 B each) and one channel
// row per plane (piece w of a plane's 8: 16 rows x 64 B)
// DUAL (bottleneck conv3 + projection shortcut as one GEMM, conv_persist.hip launch_conv_dual): K-slices [0, K1 / 32) come from `in`, the rest from `in2`
// sampled at stride2; aoff2 = the rows' offsets in `in2`
template <bool DUAL = false>
__device__ __forceinline__ void x8_tile_state(const ConvP& p, int tile, int wave, int lane, int (&aoff)[4], int& boff, int& m0, int& n0, int& g, int (&aoff2)[4]) {
    g = (int)x8_div((unsigned)tile, p.dv_m[2], p.dv_s[2]);
    const int rem = tile - g * p.pk_tpg;
    const int mt = (int)x8_div((unsigned)rem, p.dv_m[3], p.dv_s[3]);
    const int nt = rem - mt * p.ntiles;
    m0 = mt * X8_BM;
    n0 = nt * X8_BN;
    const int gin = g * (int)p.in_gs * 4, gw = g * (int)p.w_gs * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int R = 32 * wave + 8 * j + (lane >> 3);
        const int m = m0 + R;
        const int chunk = (lane & 7) ^ ((R >> 1) & 7);
        int pix = m;
        if (p.stride != 1) {
            const int b = (int)x8_div((unsigned)m, p.dv_m[0], p.dv_s[0]);
            const int r2 = m - b * p.ohw;
            const int oy = (int)x8_div((unsigned)r2, p.dv_m[1], p.dv_s[1]);
            const int ox = r2 - oy * p.OW;
            pix = (b * p.H + oy * p.stride) * p.W + ox * p.stride;
        }
        aoff[j] = m < p.M ? gin + pix * p.in_cs * 4 + chunk * 16 : X8_OOB;
        if constexpr (DUAL) {
            const int b = (int)x8_div((unsigned)m, p.dv_m[0], p.dv_s[0]);
            const int r2 = m - b * p.ohw;
            const int oy = (int)x8_div((unsigned)r2, p.dv_m[1], p.dv_s[1]);
            const int ox = r2 - oy * p.OW;
            aoff2[j] = m < p.M ? g * (int)p.in2_gs * 4 + ((b * p.H2 + oy * p.stride2) * p.W2 + ox * p.stride2) * p.in2_cs * 4 + chunk * 16 : X8_OOB;
        } else {
            aoff2[j] = 0;
        }
    }
    {
        const int R = 16 * wave + (lane >> 2);
        const int n = n0 + R;
        const int chunk = (lane & 3) ^ ((R >> 2) & 3);
        boff = n < p.Cout ? gw + n * p.Kpad * 2 + chunk * 16 : X8_OOB;
    }
}

// x = x1 + x2 + x3 in bf16 terms (round to nearest even), 8 values of a fragment
__device__ __forceinline__ void x8_split(const f32x4 lo, const f32x4 hi, bf16x8& p1, bf16x8& p2, bf16x8& p3) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = e < 4 ? lo[e] : hi[e - 4];
        const __bf16 x1 = (__bf16)x;
        const float r1 = x - (float)x1;
        const __bf16 x2 = (__bf16)r1;
        const float r2 = r1 - (float)x2;
        p1[e] = x1; p2[e] = x2; p3[e] = (__bf16)r2;
    }
}

#ifdef X8_STAMPS
// diagnostic build (make X8X=-DX8_STAMPS, tools/x8_stamps.py): s_memtime of waves 0 and 4 of the first blocks at the phase boundaries of K-slice 8 of every tile
__device__ unsigned long long g_x8_stamps[256 * 2 * 16];
#define X8_STAMP(i) do { if (lane == 0 && (wave == 0 || wave == 4) && blockIdx.x < 256 && kt == 8) g_x8_stamps[((int)blockIdx.x * 2 + (wave >> 2)) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define X8_STAMP(i) do {} while (0)
#endif

template <bool AFFINE, bool RES, bool GN, bool DUAL = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_x8_kernel(const ConvP p) {
    static_assert(!DUAL || (!RES && !GN), "the dual-input form is the 1x1 conv3 + shortcut GEMM");
    // ONE shared object (conv_h8.hip): [2 K-slice images][GroupNorm sums f64 [2][32][2]][3 scale | shift images]
    // (three, used in turn: the image a late wave may still be reading in the previous tile's epilogue is not the one wave 0 requests the next
    //  tile's vectors into - conv_h8.hip's note at its SSBASE, profiles/r20_h8_affine_race.md)
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * X8_SLOT + 1024 + 3 * X8_SS];
    constexpr int SSBASE = 2 * X8_SLOT + 1024;

    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    // A wave owns 32 pixels x ALL 128 channels of the tile (not 64 x 64): every pixel fragment is split into its bf16 terms by ONE wave instead of two - the
    // split's ~90 vector instructions per k-step were the longer half of a phase (the reading wave could not keep up with the multiplying one: matrix pipe 0.55 busy)
    const int wp = wave, late = wave >> 2;        // pixel rows 32 wp .. + 31; waves w and w + 4 share a SIMD
    const int r = lane & 31, h = lane >> 5;
    const int nk = p.Kpad / 32;

    int tile, tile_step, tile_end;
    {
        const int bid = blockIdx.x, nblk = gridDim.x, T = p.pk_T;
        const int xcd = bid & 7, q = T >> 3, rr = T & 7;
        const int start = xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q;
        tile_end = start + q + (xcd < rr ? 1 : 0);
        tile_step = (nblk >> 3) + (xcd < (nblk & 7) ? 1 : 0);
        tile = start + (bid >> 3);
    }

    int aoff[4], aoffN[4], aoff2[4], aoff2N[4], boff, boffN = X8_OOB;
    int m0, n0, g, m0N = 0, n0N = 0, gN = 0;
    x8_tile_state<DUAL>(p, tile, wave, lane, aoff, boff, m0, n0, g, aoff2);
#pragma unroll
    for (int j = 0; j < 4; ++j) { aoffN[j] = X8_OOB; aoff2N[j] = X8_OOB; }
    const __amdgpu_buffer_rsrc_t rsa2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(DUAL ? p.in2 : p.in), 0, DUAL ? p.ws_rows : 0, 0x00020000);
    const int nk1 = DUAL ? p.K1 / 32 : nk;
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, p.lean_in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w3), 0, p.pk_in_bytes, 0x00020000);
    const int plane_bytes = p.pk_in2_bytes;       // distance between the weight planes in HBM

    auto issue_p = [&](int half, int slot, int kq) __attribute__((always_inline)) {
        const bool nxt = kq >= nk;
        const bool second = DUAL && !nxt && kq >= nk1;          // this K-slice comes from the second input
        const int soff = nxt ? 0 : (second ? kq - nk1 : kq) * X8_PB;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int j = 2 * half + jj;
            if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa2, (lds_ptr_t)(smem + slot * X8_SLOT + (4 * wave + j) * 1024), 16, aoff2[j], soff, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lds_ptr_t)(smem + slot * X8_SLOT + (4 * wave + j) * 1024), 16, nxt ? aoffN[j] : aoff[j], soff, 0, 0);
        }
    };
    auto issue_q = [&](int slot, int kq) __attribute__((always_inline)) {
        const bool nxt = kq >= nk;
        const int soff = nxt ? 0 : kq * X8_QB;
#pragma unroll
        for (int q = 0; q < 3; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, (lds_ptr_t)(smem + slot * X8_SLOT + X8_QBASE + q * X8_QPLANE + wave * 1024), 16, nxt ? boffN : boff,
                                                     soff + q * plane_bytes, 0, 0);
    };
    const __amdgpu_buffer_rsrc_t rss = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.scale), 0, AFFINE ? p.h8_ss_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.shift), 0, AFFINE ? p.h8_ss_bytes : 0, 0x00020000);
    auto issue_ss = [&](int buf, int tg, int tn0) __attribute__((always_inline)) {
        if constexpr (AFFINE) {
            if (wave == 0) {
                const int off = (tg * p.ss_gs + tn0) * 4 + lane * 16;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rss, (lds_ptr_t)(smem + SSBASE + buf * X8_SS), 16, off, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsh, (lds_ptr_t)(smem + SSBASE + buf * X8_SS + 1024), 16, off, 0, 0, 0);
            }
        }
    };
    int ssb = 0;

    // fragment addresses.  Pixels: row R at R * 128 B, k-step s needs the floats k = 16 s + 8 h .. + 7 = logical 16-byte chunks 4 s + 2 h,
    // 4 s + 2 h + 1, at physical chunk (logical) ^ ((R >> 1) & 7).  Weights: row R of a plane at R * 64 B, k-step s needs the 8 bf16
    // k = 16 s + 8 h .. = logical chunk 2 s + h of 4, at physical chunk (logical) ^ ((R >> 2) & 3).
    int paddr[2][2], qaddr[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int j = 0; j < 2; ++j) paddr[s][j] = (32 * wp + r) * X8_PB + (((4 * s + 2 * h + j) ^ (((32 * wp + r) >> 1) & 7)) << 4);
        qaddr[s] = X8_QBASE + r * X8_QB + (((2 * s + h) ^ ((r >> 2) & 3)) << 4);
    }

    f32x16 acc[4], top[4];                         // [channel tile of 32]
    bf16x8 wf[4][3], xf[3];                        // [channel tile][term], [term]
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // ---- prologue: K-slice 0 ----
    issue_ss(0, g, n0);
    issue_p(0, 0, 0); issue_p(1, 0, 0); issue_q(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int gk = 0;

    // reads of k-step S, then the split of the pixel fragments (vector instructions that run beside the partner wave's MFMAs)
#define X8_READ(S)                                                                                                        \
    {                                                                                                                      \
        f32x4 raw[2];                                                                                                      \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) raw[j] = *reinterpret_cast<const f32x4*>(smem + sbase + paddr[S][j]); \
        _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                                      \
            _Pragma("unroll") for (int q = 0; q < 3; ++q)                                                                  \
                wf[c][q] = *reinterpret_cast<const bf16x8*>(smem + sbase + qaddr[S] + q * X8_QPLANE + c * 32 * X8_QB);     \
        x8_split(raw[0], raw[1], xf[0], xf[1], xf[2]);                                                                     \
    }
    // six partial products per output tile, smallest first: (a3 b1) (a1 b3) (a2 b2) (a2 b1) (a1 b2) (a1 b1), a = pixels, b = weights
#define X8_MMA()                                                                                                          \
    __builtin_amdgcn_s_barrier();                                                                                          \
    __builtin_amdgcn_s_setprio(1);                                                                                         \
    _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                                        \
            f32x16 d = acc[c];                                                                                             \
            d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c][0], xf[2], d, 0, 0, 0);                                      \
            d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c][2], xf[0], d, 0, 0, 0);                                      \
            d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c][1], xf[1], d, 0, 0, 0);                                      \
            d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c][0], xf[1], d, 0, 0, 0);                                      \
            d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c][1], xf[0], d, 0, 0, 0);                                      \
            d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[c][0], xf[0], d, 0, 0, 0);                                      \
            acc[c] = d;                                                                                                    \
        }                                                                                                                  \
    __builtin_amdgcn_s_setprio(0);                                                                                         \
    __builtin_amdgcn_s_barrier();

    for (;;) {
        const bool has_next = tile + tile_step < tile_end;
        if (has_next) {
            x8_tile_state<DUAL>(p, tile + tile_step, wave, lane, aoffN, boffN, m0N, n0N, gN, aoff2N);
            issue_ss(ssb == 2 ? 0 : ssb + 1, gN, n0N);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) { aoffN[j] = X8_OOB; aoff2N[j] = X8_OOB; }
            boffN = X8_OOB;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) { top[c] = zero16; acc[c] = zero16; }
        int fold_in = p.acc_chunk;
        if (late == 1) __builtin_amdgcn_s_barrier();      // stagger: waves 4-7 run half a phase behind their SIMD partners

        for (int kt = 0; kt < nk; ++kt, ++gk) {
            const int s = gk & 1;
            const int sbase = s * X8_SLOT;
            // phase 0: k-step 0; DMA: the pixel rows and the three weight planes of the next slice into the other image (its last readers retired their reads before
            // the first barrier of the previous slice's phase 1)
            X8_STAMP(0);
            X8_READ(0)
            X8_STAMP(1);
            issue_p(0, s ^ 1, kt + 1);
            issue_p(1, s ^ 1, kt + 1);
            issue_q(s ^ 1, kt + 1);
            X8_STAMP(2);
            X8_MMA()
            X8_STAMP(3);
            // phase 1: k-step 1; the next slice (issued a phase ago) has landed after this wait + barrier pair, and this image's last reads
            // are retired before the barrier
            // (the wait has to stand in front of this phase's FIRST barrier: the partner waves run one barrier behind, and what they read after their
            //  next barrier includes this wave's pieces - moved behind the MFMAs it raced, and bought nothing: the fill is not what this loop waits for)
            X8_READ(1)
            X8_STAMP(4);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            X8_STAMP(5);
            X8_MMA()
            X8_STAMP(6);
            // two-level accumulation: the MFMA chain of p.acc_chunk slices into the running sums (conv_igemm.hip / conv_persist.hip `fold`)
            if (kt + 1 == nk || --fold_in == 0) {
                fold_in = p.acc_chunk;
#pragma unroll
                for (int c = 0; c < 4; ++c) { top[c] += acc[c]; acc[c] = zero16; }
            }
        }
        if (late == 0) __builtin_amdgcn_s_barrier();    // the two halves level again

        // ---- epilogue (fp32 output, 4 consecutive channels = 16 bytes per lane and store) ----
        {
            const int rows = min(p.M - m0, X8_BM);
            const long org = (long)g * p.out_gs + (long)m0 * p.out_cs + n0;
            const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.out + org, 0, ((rows - 1) * p.out_cs + min(p.Cout - n0, X8_BN)) * 4, 0x00020000);
            const long rorg = (long)g * p.res_gs + (long)m0 * p.res_cs + n0;
            const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res) + (RES ? rorg : 0), 0,
                                                                                 RES ? ((rows - 1) * p.res_cs + min(p.Cout - n0, X8_BN)) * 4 : 0, 0x00020000);
            const float lo = p.relu ? 0.f : -__builtin_inff();
            const int b0 = GN ? (int)x8_div((unsigned)m0, p.dv_m[0], p.dv_s[0]) : 0;
            const int m_next = (b0 + 1) * p.ohw;
            const bool plain = m0 + X8_BM <= p.M && m0 + X8_BM <= m_next;
            const unsigned gacc_b = 2 * X8_SLOT;
            if constexpr (GN) {
                if (t < 128) {
                    const unsigned long long z = 0;
                    asm volatile("ds_write_b64 %0, %1" :: "v"(gacc_b + t * 8), "v"(z) : "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            int r_e = r, h_e = h;
            asm volatile("" : "+v"(r_e), "+v"(h_e));
            const int prow0 = 32 * wp + r_e;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int nl = 32 * c + 8 * j + 4 * h_e;
                    const bool colok = n0 + nl < p.Cout;
                    const int obase = colok ? (prow0 * p.out_cs + nl) * 4 : X8_OOB;
                    u32x4 rbuf[1];
                    if constexpr (RES) {
                        const int rbase = colok ? (prow0 * p.res_cs + nl) * 4 : X8_OOB;
                        rbuf[0] = __builtin_amdgcn_raw_buffer_load_b128(rsr, rbase, 0, 0);
                    }
                    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (AFFINE) {
                        const unsigned ad = SSBASE + ssb * X8_SS + nl * 4;
                        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)" : "=&v"(sc), "=&v"(sh) : "v"(ad) : "memory");
                    }
                    double gs = 0.0, gq = 0.0, gs1 = 0.0, gq1 = 0.0;
#pragma unroll
                    for (int i = 0; i < 1; ++i) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = top[c][4 * j + e];
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
                        auto row_sum = [](double x) __attribute__((always_inline)) {
#define X8_DPP_STEP(CTL)                                                                                                   \
    {                                                                                                                      \
        const unsigned long long u = __builtin_bit_cast(unsigned long long, x);                                            \
        const unsigned lo32 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTL, 0xf, 0xf, true);             \
        const unsigned hi32 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTL, 0xf, 0xf, true);     \
        x += __builtin_bit_cast(double, ((unsigned long long)hi32 << 32) | lo32);                                          \
    }
                            X8_DPP_STEP(0x111) X8_DPP_STEP(0x112) X8_DPP_STEP(0x114) X8_DPP_STEP(0x118)
#undef X8_DPP_STEP
                            return x;
                        };
                        auto lds_add = [&](unsigned slot, double dx) __attribute__((always_inline)) {
                            asm volatile("ds_add_f64 %0, %1" :: "v"(gacc_b + slot * 8), "v"(dx) : "memory");
                        };
                        const int grp = (int)x8_div((unsigned)(n0 + nl), p.dv_m[4], p.dv_s[4]);
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
        for (int j = 0; j < 4; ++j) { aoff[j] = aoffN[j]; aoff2[j] = aoff2N[j]; }
        boff = boffN;
        m0 = m0N; n0 = n0N; g = gN;
        ssb = ssb == 2 ? 0 : ssb + 1;
    }
#undef X8_READ
#undef X8_MMA
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

static void x8_magic(unsigned d, unsigned& m, unsigned& s) {      // conv_h8.hip: h8_magic
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    m = (unsigned)((((1ull << l) - d) << 32) / d + 1);
    s = ((l > 0 ? l - 1 : 0) << 1) | (l > 0 ? 1u : 0u);
}

// w [n] fp32 -> three planes of bf16 terms [3][n]
__global__ void split_bf16x3_kernel(const float* __restrict__ w, long n, unsigned short* __restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float x = w[i];
        const __bf16 x1 = (__bf16)x;
        const float r1 = x - (float)x1;
        const __bf16 x2 = (__bf16)r1;
        const float r2 = r1 - (float)x2;
        const __bf16 x3 = (__bf16)r2;
        out[i] = __builtin_bit_cast(unsigned short, x1);
        out[n + i] = __builtin_bit_cast(unsigned short, x2);
        out[2 * n + i] = __builtin_bit_cast(unsigned short, x3);
    }
}

}  // namespace

#ifdef X8_STAMPS
int x8_read_stamps(unsigned long long* dst, int n) {
    if (n > 256 * 2 * 16) n = 256 * 2 * 16;
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_x8_stamps), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#endif

int launch_split_bf16x3(const float* w, long n, void* planes, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(split_bf16x3_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 2048)), dim3(256), 0, st, w, n, (unsigned short*)planes);
    QB_CHECK(hipGetLastError());
    return 0;
}

// bf16x3 mode, 1x1 / pad 0, K a multiple of 32, pre-split weight planes present, one input, 16-byte epilogue accesses, a tile count that
// fills the chip's one-block-per-CU grid a few times.  returns 0 = launched, 1 = not covered, -1 = error
int launch_conv_x8(ConvP p, int G, hipStream_t st) {
    if (!tune().x8 || p.es != 4 || p.bf16 != 3 || !p.w3 || p.prelu || p.acc_chunk < 0) return 1;
    const bool dual = p.in2 != nullptr;          // launch_conv_dual: K = K1 channels of `in`, then the channels of `in2` sampled at stride2
    if (p.kh != 1 || p.kw != 1 || p.pad != 0 || p.Cin % 32 || p.K != p.Kpad || p.Cin != (dual ? p.K1 : p.K) || p.Kpad / 32 < 2) return 1;
    if (dual) {
        if (p.res || p.gn_sum || p.stride != 1 || !p.scale || p.K1 % 32 || p.K1 <= 0 || p.K1 >= p.Kpad || p.in2_cs % 4 || (p.in2_gs & 3) || ((uintptr_t)p.in2 & 15)) return 1;
        const long in2_all = ((long)p.B * p.H2 * p.W2 * p.in2_cs) * 4 + (long)(G - 1) * p.in2_gs * 4;
        if (in2_all >= 0x7fffff00L) return 1;
        p.ws_rows = (int)in2_all;               // (descriptor range of the second input: a field the LDS-DMA kernels do not use otherwise)
    }
    if ((p.scale == nullptr) != (p.shift == nullptr)) return 1;
    const long in_bytes = ((long)p.B * p.H * p.W * p.in_cs) * 4;
    const long in_all = in_bytes + (long)(G - 1) * p.in_gs * 4;
    const long plane_bytes = p.w3_plane * 2;
    if (in_all >= 0x7fffff00L || 3 * plane_bytes >= 0x7fffff00L || (long)(G - 1) * p.w_gs * 2 + (long)p.Cout * p.Kpad * 2 > plane_bytes) return 1;
    const bool vec4 = p.Cout % 4 == 0 && p.out_cs % 4 == 0 && p.out_gs % 4 == 0 && (((uintptr_t)p.out & 15) == 0) && p.in_cs % 4 == 0 && (p.in_gs & 3) == 0 &&
                      (((uintptr_t)p.w3 & 15) == 0) && (p.w_gs & 7) == 0 && (p.w3_plane & 7) == 0 &&
                      (!p.res || (p.res_cs % 4 == 0 && p.res_gs % 4 == 0 && (((uintptr_t)p.res & 15) == 0))) &&
                      (!p.scale || (p.ss_gs % 4 == 0 && (((uintptr_t)p.scale & 15) == 0) && (((uintptr_t)p.shift & 15) == 0)));
    if (!vec4 || (long)p.out_cs * X8_BM * 4 >= 0x7fffff00L) return 1;
    p.mtiles = (p.M + X8_BM - 1) / X8_BM;
    p.ntiles = (p.Cout + X8_BN - 1) / X8_BN;
    const long tiles = (long)p.mtiles * p.ntiles * G;
    if (tiles > 0x3fffffff) return 1;
    const int cus = device_cus();          // per device: a process may drive several
    if (cus <= 0) return fail("conv_x8: cannot query the device");
    if (tune().x8 < 2) {         // (key 35 = 2: every covered launch - the tests)
        if (p.Kpad / 32 < tune().x8_min_nk || p.Cout < X8_BN) return 1;      // short K: the HBM-bound residual layers keep conv_igemm.hip's 64 x 64 tiles
        if (tiles < (long)tune().x8_min_rounds * cus) return 1;
    }
    p.lean_in_bytes = (int)in_all;
    p.pk_in_bytes = (int)(3 * plane_bytes);
    p.pk_in2_bytes = (int)plane_bytes;
    p.pk_T = (int)tiles;
    p.pk_tpg = p.mtiles * p.ntiles;
    x8_magic((unsigned)p.ohw, p.dv_m[0], p.dv_s[0]);
    x8_magic((unsigned)p.OW, p.dv_m[1], p.dv_s[1]);
    x8_magic((unsigned)p.pk_tpg, p.dv_m[2], p.dv_s[2]);
    x8_magic((unsigned)p.ntiles, p.dv_m[3], p.dv_s[3]);
    x8_magic((unsigned)(p.gn_sum && p.gn_cpg > 0 ? p.gn_cpg : 1), p.dv_m[4], p.dv_s[4]);
    p.h8_ss_bytes = ((G - 1) * p.ss_gs + p.Cout) * 4;
    if (p.acc_chunk == 0) p.acc_chunk = 0x7fffffff;       // 0 = one chain over the whole K
    const bool gn_sep = p.gn_sum && !(p.gn_cpg % 4 == 0 && p.gn_groups <= 32 && p.ohw >= X8_BM);
    double* const gn_sum = p.gn_sum;
    if (gn_sep) p.gn_sum = nullptr;
    {
        const double out_bytes = 4.0 * G * (double)p.M * p.Cout;
        const double conv_bytes = 4.0 * G * ((double)p.B * p.H * p.W * p.Cin + (double)p.Cout * p.K) + out_bytes * (p.res ? 2.0 : 1.0);
        const double conv_flops = 2.0 * G * (double)p.M * p.K * p.Cout;
        const char* tag = !p.tag ? "conv_gemm_x8" : std::string(p.tag) == "wino_gemm" ? "wino_gemm_x8" : p.tag;
        ProfScope prof(tag, conv_bytes, conv_flops, st);
        const dim3 grid((unsigned)std::min<long>(tiles, cus)), block(512);
        const int variant = dual ? 8 : (p.scale ? 4 : 0) | (p.res ? 2 : 0) | (p.gn_sum ? 1 : 0);
        switch (variant) {
            case 8: hipLaunchKernelGGL((conv_x8_kernel<true, false, false, true>), grid, block, 0, st, p); break;
            case 0: hipLaunchKernelGGL((conv_x8_kernel<false, false, false>), grid, block, 0, st, p); break;
            case 1: hipLaunchKernelGGL((conv_x8_kernel<false, false, true>), grid, block, 0, st, p); break;
            case 2: hipLaunchKernelGGL((conv_x8_kernel<false, true, false>), grid, block, 0, st, p); break;
            case 3: hipLaunchKernelGGL((conv_x8_kernel<false, true, true>), grid, block, 0, st, p); break;
            case 4: hipLaunchKernelGGL((conv_x8_kernel<true, false, false>), grid, block, 0, st, p); break;
            case 5: hipLaunchKernelGGL((conv_x8_kernel<true, false, true>), grid, block, 0, st, p); break;
            case 6: hipLaunchKernelGGL((conv_x8_kernel<true, true, false>), grid, block, 0, st, p); break;
            default: hipLaunchKernelGGL((conv_x8_kernel<true, true, true>), grid, block, 0, st, p); break;
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
