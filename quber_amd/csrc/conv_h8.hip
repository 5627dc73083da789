// fp16 data path (quber_config.compute_dtype 2, BASELINE.json configs[4]): the implicit-GEMM convolution on 256 x 256 tiles
// with an LDS-DMA operand pipeline - the kernel of the wide layers (>= 256 output channels, K = taps * Cin a multiple of 64).
//
// Layers it runs (reference): the 3x3 fusion convolutions of maskrefiner/modeling/backbone/resnet.py:472-485, the bottleneck
// convolutions of res4 / res5 (resnet.py:395-449, detectron2 BottleneckBlock) and the 1x1 reductions in front of them; epilogue =
// the Conv2d wrapper's per-channel affine (FrozenBN / bias), residual add, ReLU, and the GroupNorm sums of the stored values.
//
// Structure (conv_igemm.hip's 128 x 128 fp16 kernel holds MFMA busy 0.43: one K-slice of register prefetch, two barriers per
// slice, ~5 vector instructions beside every MFMA; profiles/r10b_h16_loader.md):
//   * block = 8 waves = 2 per SIMD, tile = 256 pixels x 256 channels, K consumed 64 halfs (128 bytes per row) at a time;
//   * both operands go global -> LDS by `buffer_load_dwordx4 ... lds` (no staging registers, no ds_write): a wave-instruction
//     fills 8 rows x 128 bytes; the im2col gather is the per-lane SOURCE offset (pixel base + block-uniform tap offset), and a
//     padding tap is an out-of-range offset, for which the DMA writes zeros (tools/micro/dma_oob_probe.hip);
//   * two K-tile images of 64 KB; a K-tile is four 16 KB half-tiles (pixels P0 P1, channels Q0 Q1); one half-tile is issued per
//     phase, two K-tiles ahead for P and one ahead for Q, and retired by ONE counted `s_waitcnt vmcnt(4)` per K-tile - the
//     loop never drains the queue, and its barriers are raw s_barrier (a __syncthreads() would wait for vmcnt(0));
//   * four phases per K-tile, each {fragment reads + one half-tile of DMA | barrier | 16 MFMAs | barrier}; the two waves of a
//     SIMD run half a phase apart (waves 4-7 pass one extra barrier up front), so one multiplies while the other reads;
//   * LDS rows are 128 bytes with the 16-byte chunks XOR-swizzled (on the DMA source side: the LDS image of a DMA is
//     lane-linear) so that every 16-lane group of a ds_read_b128 covers all 64 banks once;
//   * v_mfma_f32_16x16x32_f16 with the WEIGHTS as the row operand: a lane's four accumulator values are four consecutive
//     output channels of one pixel, and with the channel rows of a tile pair interleaved a lane owns 8 consecutive channels -
//     the epilogue stores 16 bytes per lane straight from the accumulators, no transpose through LDS.
#include "common.h"

namespace quber {

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using h16x8 = __attribute__((ext_vector_type(8))) _Float16;
using h16x4 = __attribute__((ext_vector_type(4))) _Float16;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int H8_BM = 256;                  // pixels per tile
constexpr int H8_KB = 128;                  // bytes of one K-tile row (64 halfs)
constexpr int H8_HALF = 128 * H8_KB;        // one half-tile image: 128 rows
constexpr int H8_OOB = (int)0x80000000;     // a buffer offset past every descriptor range: the DMA writes zeros

// QT = 16-channel tiles per wave: 8 (256-channel tile: waves 4 (pixels) x 2 (channels), 64 x 128 each)
template <int QT>
struct H8Geo {
    static constexpr int BN = 2 * QT * 16;               // channels per tile
    static constexpr int QHALF = BN / 2 * H8_KB;         // bytes of one channel half-tile image
    static constexpr int SLOT = 2 * H8_HALF + 2 * QHALF; // one K-tile image
    static constexpr int QBASE = 2 * H8_HALF;            // channel rows start here inside a slot
};

template <int QT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_h8_kernel(const ConvP p) {
    using G = H8Geo<QT>;
    constexpr int BN = G::BN, SLOT = G::SLOT;
    constexpr int QPW = QT * 16 / 8 / 8;          // DMA pieces (8 rows) of a channel half-tile per wave: 2 (QT 8)
    static_assert(QT == 8, "geometry");
    // ONE shared object: a second one beside a DMA target makes hipcc wait vmcnt(0) before the fragment reads
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * SLOT];

    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    const int wp = wave & 3, wq = wave >> 2;      // pixel quarter, channel half of this wave; waves w and w + 4 share a SIMD
    const int fr = lane & 15, fq = lane >> 4;
    const int g = blockIdx.z;

    // XCD-aware tile order (as conv_igemm.hip): blocks b and b + 8 share an XCD; every XCD takes a contiguous run of tiles
    int tile;
    {
        const int bid = blockIdx.x, nblk = gridDim.x;
        const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int nt = tile % p.ntiles, mt = tile / p.ntiles;
    const int m0 = mt * H8_BM, n0 = nt * BN;
    const int nk = p.Kpad / 32;                   // K-tiles (ConvP of the fp16 path counts K in 4-byte units)

    // ---- DMA source state: 4 pixel rows and 4 channel rows per thread ----
    // piece u (8 rows x 128 B) of a half-tile goes to wave u / 2; lane l fills row l >> 3, physical chunk l & 7 of it and
    // fetches the LOGICAL chunk (l & 7) ^ swizzle(row)
    int aoff[4], boff[4];
    unsigned amask[4];
    const int prow = lane >> 3, pc = lane & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int R = 128 * (i >> 1) + 8 * (2 * wave + (i & 1)) + prow;         // pixel row of the tile
        const int m = m0 + R;
        const int chunk = pc ^ ((R >> 1) & 7);
        aoff[i] = 0;
        amask[i] = 0;
        if (m < p.M && p.kh == 1 && p.stride == 1 && p.pad == 0) {
            aoff[i] = (m * p.in_cs) * 4 + chunk * 16;
            amask[i] = 1;
        } else if (m < p.M) {
            const int b = m / p.ohw;
            const int rem = m - b * p.ohw;
            const int oy = rem / p.OW;
            const int ox = rem - oy * p.OW;
            const int y0 = oy * p.stride - p.pad, x0 = ox * p.stride - p.pad;
            aoff[i] = (((b * p.H + y0) * p.W + x0) * p.in_cs) * 4 + chunk * 16;
            unsigned xbits = 0, mk = 0;
            for (int tx = 0; tx < p.kw; ++tx)
                if ((unsigned)(x0 + tx * p.dil) < (unsigned)p.W) xbits |= 1u << tx;
            for (int ty = 0; ty < p.kh; ++ty)
                if ((unsigned)(y0 + ty * p.dil) < (unsigned)p.H) mk |= xbits << (ty * p.kw);
            amask[i] = mk;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int R = (BN / 2) * (i >> 1) + 8 * (QPW * wave + (i & 1)) + prow;  // channel row of the tile
        const int n = n0 + R;
        const int chunk = pc ^ (((R >> 1) & 1) | (((R >> 3) & 3) << 1));
        boff[i] = n < p.Cout ? (n * p.Kpad) * 4 + chunk * 16 : H8_OOB;          // rows past Cout: zeros
    }
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.in) + (long)g * p.in_gs * 4), 0, p.lean_in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.w) + (long)g * p.w_gs * 4), 0, p.Cout * p.Kpad * 4, 0x00020000);

    // block-uniform position of the next pixel K-tile to issue: tap (ky, kx), channel block
    int sk = 0, skc = 0, skx = 0, sky = 0;
    auto issue_p = [&](int half, int slot) __attribute__((always_inline)) {
        const int tap = sk < nk ? sky * p.kw + skx : 31;          // past the end of K: every lane out of range (no traffic)
        const int soff = (((sky * p.dil) * p.W + skx * p.dil) * p.in_cs + skc) * 4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = 2 * half + j;
            const bool ok = (amask[i] >> tap) & 1u;
            const int voff = ok ? aoff[i] + soff : H8_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lds_ptr_t)(smem + slot * SLOT + half * H8_HALF + (2 * wave + j) * 1024), 16, voff, 0, 0, 0);
        }
    };
    auto advance_p = [&]() __attribute__((always_inline)) {
        ++sk;
        if (p.kmode) {                    // k = (channel block, tap, channel): the taps of one 64-channel block are consecutive K-tiles
            if (++skx == p.kw) {
                skx = 0;
                if (++sky == p.kh) { sky = 0; skc += 32; }
            }
        } else {                          // k = (tap, channel)
            skc += 32;
            if (skc >= p.Cin) {
                skc = 0;
                if (++skx == p.kw) { skx = 0; ++sky; }
            }
        }
    };
    auto issue_q = [&](int half, int slot, int kt) __attribute__((always_inline)) {
        const bool live = kt < nk;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = 2 * half + j;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, (lds_ptr_t)(smem + slot * SLOT + G::QBASE + half * G::QHALF + (QPW * wave + j) * 1024), 16,
                                                     live ? boff[i] : H8_OOB, kt * H8_KB, 0, 0);
        }
    };

    // ---- fragment addresses ----
    // pixel tile i of this wave: row 64 wp + 16 i + fr; k-step ks: logical chunk 4 ks + fq
    // channel tile c = 2 gg + jj: tile row fr -> channel row 128 wq + 32 gg + 8 (fr >> 2) + 4 jj + (fr & 3): the accumulator rows
    // 4 fq + e of tiles 2 gg, 2 gg + 1 are then the 8 consecutive channels 32 gg + 8 fq .. + 7
    const int sp = fr >> 1, sq = ((fr & 3) >> 1) | ((fr >> 2) << 1);
    int paddr[2], qaddr[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        paddr[ks] = (64 * wp + fr) * H8_KB + (((4 * ks + fq) ^ sp) << 4);
        qaddr[ks] = G::QBASE + (QT * 16 * wq + 8 * (fr >> 2) + (fr & 3)) * H8_KB + (((4 * ks + fq) ^ sq) << 4);
    }

    f32x4 acc[QT][4];
#pragma unroll
    for (int c = 0; c < QT; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[c][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    h16x8 pf[2][4], qf[4];

    // ---- prologue: K-tile 0 whole, the pixel halves of K-tile 1 ----
    issue_p(0, 0); issue_p(1, 0); advance_p();
    issue_q(0, 0, 0); issue_q(1, 0, 0);
    issue_p(0, 1); issue_p(1, 1); advance_p();
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wq == 1) __builtin_amdgcn_s_barrier();      // stagger: waves 4-7 run half a phase behind their SIMD partners

#define H8_READ_Q(JH, KS)                                                                                              \
    _Pragma("unroll") for (int c4 = 0; c4 < 4; ++c4) {                                                                  \
        const int c = 4 * (JH) + c4;                                                                                    \
        qf[c4] = *reinterpret_cast<const h16x8*>(smem + sbase + qaddr[KS] + ((c >> 1) * 32 + (c & 1) * 4) * H8_KB);      \
    }
#define H8_MMA(JH, KS)                                                                                                 \
    __builtin_amdgcn_s_barrier();                                                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                  \
    __builtin_amdgcn_s_setprio(1);                                                                                      \
    _Pragma("unroll") for (int c4 = 0; c4 < 4; ++c4)                                                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                   \
            acc[4 * (JH) + c4][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qf[c4], pf[KS][i], acc[4 * (JH) + c4][i], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                                      \
    __builtin_amdgcn_s_barrier();

    for (int kt = 0; kt < nk; ++kt) {
        const int s = kt & 1;
        const int sbase = s * SLOT;
        // phase 0: every pixel fragment of the K-tile + channel tiles 0-3, k-step 0; DMA: channel half 0 of K-tile kt + 1
        H8_READ_Q(0, 0)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) pf[ks][i] = *reinterpret_cast<const h16x8*>(smem + sbase + paddr[ks] + i * 16 * H8_KB);
        issue_q(0, s ^ 1, kt + 1);
        H8_MMA(0, 0)
        // phase 1: channel tiles 4-7, k-step 0; DMA: channel half 1 of K-tile kt + 1
        H8_READ_Q(1, 0)
        issue_q(1, s ^ 1, kt + 1);
        H8_MMA(1, 0)
        // phase 2: channel tiles 4-7, k-step 1; DMA: pixel half 0 of K-tile kt + 2 (this image's pixel rows were read in phase 0)
        H8_READ_Q(1, 1)
        issue_p(0, s);
        H8_MMA(1, 1)
        // phase 3: channel tiles 0-3, k-step 1; DMA: pixel half 1 of K-tile kt + 2; everything older than the two pixel halves
        // just issued has landed after this wait + the next barrier pair
        H8_READ_Q(0, 1)
        issue_p(1, s);
        advance_p();
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        H8_MMA(0, 1)
    }
#undef H8_READ_Q
#undef H8_MMA
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the out-of-range tail DMAs still write (zeros) into the images
    if (wq == 0) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();

    // ---- epilogue: y = acc * scale + shift (+ residual) (ReLU), 8 channels = 16 bytes per lane ----
    _Float16* __restrict__ out = reinterpret_cast<_Float16*>(p.out) + (long)g * p.out_gs;
    const _Float16* __restrict__ res = p.res ? reinterpret_cast<const _Float16*>(p.res) + (long)g * p.res_gs : nullptr;
    const float* __restrict__ scale = p.scale ? p.scale + g * p.ss_gs : nullptr;
    const float* __restrict__ shift = p.shift ? p.shift + g * p.ss_gs : nullptr;
    double* const gacc = reinterpret_cast<double*>(smem);       // [image b0 / b0 + 1][group][sum, sum of squares]
    const bool gn = p.gn_sum != nullptr;
    int b0 = 0, m_next = 0;
    if (gn) {
        if (t < 128) gacc[t] = 0.0;
        b0 = m0 / p.ohw;
        m_next = (b0 + 1) * p.ohw;
        __syncthreads();
    }
#pragma unroll
    for (int gg = 0; gg < QT / 2; ++gg) {
        const int n = n0 + QT * 16 * wq + 32 * gg + 8 * fq;
        if (n >= p.Cout) continue;
        float sc[8], sh[8];
        if (scale) {
            *reinterpret_cast<float4*>(sc) = *reinterpret_cast<const float4*>(scale + n);
            *reinterpret_cast<float4*>(sc + 4) = *reinterpret_cast<const float4*>(scale + n + 4);
            *reinterpret_cast<float4*>(sh) = *reinterpret_cast<const float4*>(shift + n);
            *reinterpret_cast<float4*>(sh + 4) = *reinterpret_cast<const float4*>(shift + n + 4);
        }
        double s0[2] = {0.0, 0.0}, q0[2] = {0.0, 0.0}, s1[2] = {0.0, 0.0}, q1[2] = {0.0, 0.0};   // [channel half] of image b0 / b0 + 1
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + 64 * wp + 16 * i + fr;
            if (m >= p.M) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = acc[2 * gg][i][e]; v[4 + e] = acc[2 * gg + 1][i][e]; }
            if (scale) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], sc[e], sh[e]);
            }
            if (res) {
                const h16x8 rh = *reinterpret_cast<const h16x8*>(res + (long)m * p.res_cs + n);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)rh[e];
            }
            h16x8 hv;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (p.relu) v[e] = fmaxf(v[e], 0.f);
                hv[e] = (_Float16)v[e];
                v[e] = (float)hv[e];
            }
            *reinterpret_cast<h16x8*>(out + (long)m * p.out_cs + n) = hv;
            if (gn) {       // sums of the stored (rounded) values, as conv_igemm.hip
                const double a = (double)v[0] + (double)v[1] + (double)v[2] + (double)v[3];
                const double b = (double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2] + (double)v[3] * v[3];
                const double a2 = (double)v[4] + (double)v[5] + (double)v[6] + (double)v[7];
                const double b2 = (double)v[4] * v[4] + (double)v[5] * v[5] + (double)v[6] * v[6] + (double)v[7] * v[7];
                if (m < m_next) { s0[0] += a; q0[0] += b; s0[1] += a2; q0[1] += b2; } else { s1[0] += a; q1[0] += b; s1[1] += a2; q1[1] += b2; }
            }
        }
        if (gn) {
            // the 16 lanes fr = 0..15 of a row hold the same channels: reduce over them, one lane adds to the block's sums
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int d = 1; d < 16; d <<= 1) {
                    s0[h] += __shfl_xor(s0[h], d); q0[h] += __shfl_xor(q0[h], d);
                    s1[h] += __shfl_xor(s1[h], d); q1[h] += __shfl_xor(q1[h], d);
                }
            }
            if (fr == 0) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int grp = (n + 4 * h) / p.gn_cpg;
                    atomicAdd(&gacc[grp * 2], s0[h]); atomicAdd(&gacc[grp * 2 + 1], q0[h]);
                    if (s1[h] != 0.0 || q1[h] != 0.0) { atomicAdd(&gacc[64 + grp * 2], s1[h]); atomicAdd(&gacc[64 + grp * 2 + 1], q1[h]); }
                }
            }
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

// The launches this kernel takes: fp16 tensors, every K-tile inside one filter tap (Cin a multiple of 64 halfs), at most 31 taps,
// views below 2 GiB, 16-byte epilogue accesses, no second input, and enough tiles to fill the chip.
// returns 0 = launched, 1 = not covered (the caller runs conv_igemm.hip), -1 = error
int launch_conv_h8(ConvP p, int G, hipStream_t st) {
    if (!tune().h8 || p.es != 2 || p.in2 || p.prelu || p.skip_rows) return 1;
    // (ConvP of the fp16 path: Cin / in_cs / K / Kpad / in_gs / w_gs are in 4-byte units)
    if (p.Cin % 32 || p.K != p.Kpad || p.kh * p.kw > 31 || p.Kpad / 32 < 2) return 1;
    if (p.Cout < 256 && tune().h8 < 2) return 1;
    const long in_bytes = ((long)p.B * p.H * p.W * p.in_cs) * 4, w_bytes = (long)p.Cout * p.Kpad * 4;
    if (in_bytes >= 0x7fffff00L || w_bytes >= 0x7fffff00L) return 1;
    const bool vec8 = p.Cout % 8 == 0 && p.out_cs % 8 == 0 && p.out_gs % 8 == 0 && (((uintptr_t)p.out & 15) == 0) &&
                      (!p.res || (p.res_cs % 8 == 0 && p.res_gs % 8 == 0 && (((uintptr_t)p.res & 15) == 0))) &&
                      (!p.scale || (p.ss_gs % 4 == 0 && (((uintptr_t)p.scale & 15) == 0) && (((uintptr_t)p.shift & 15) == 0)));
    if (!vec8) return 1;
    p.mtiles = (p.M + H8_BM - 1) / H8_BM;
    p.ntiles = (p.Cout + 255) / 256;
    const long tiles = (long)p.mtiles * p.ntiles * G;
    if (tiles < tune().h8_min_tiles) return 1;
    p.lean_in_bytes = (int)in_bytes;
    // GroupNorm sums in the epilogue: whole 4-channel halves inside one norm group, at most 32 groups, images of at least one
    // tile of rows (a tile then meets at most two images); otherwise a separate pass over the output
    const bool gn_sep = p.gn_sum && !(p.gn_cpg % 4 == 0 && p.gn_groups <= 32 && p.ohw >= H8_BM);
    double* const gn_sum = p.gn_sum;
    if (gn_sep) p.gn_sum = nullptr;
    {
        const double out_bytes = 2.0 * G * (double)p.M * p.Cout;
        const double conv_bytes = 4.0 * G * ((double)p.B * p.H * p.W * p.Cin + (double)p.Cout * p.K) + out_bytes * (p.res ? 2.0 : 1.0);
        const double conv_flops = 2.0 * G * (double)p.M * p.K * p.Cout * 2.0;
        ProfScope prof(p.tag ? p.tag : "conv_gemm_h8", conv_bytes, conv_flops, st);
        hipLaunchKernelGGL((conv_h8_kernel<8>), dim3(p.mtiles * p.ntiles, 1, G), dim3(512), 0, st, p);
    }
    QB_CHECK(hipGetLastError());
    if (gn_sep) {
        View o;
        o.p = p.out; o.B = p.B; o.H = p.OH; o.W = p.OW; o.C = p.Cout; o.cs = p.out_cs; o.gs = p.out_gs; o.es = 2;
        return launch_gn_stats(o, p.B, G, p.gn_groups, gn_sum, st, false);
    }
    return 0;
}

}  // namespace quber
