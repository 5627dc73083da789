// a3 + the first stem convolution as ONE kernel ("a3 ... or 0 if fused into the stem loader").
//   reference: model.py:137-153 (normalise, concatenate image | initial-mask encoding) -> resnet.py:37-44 stem.conv1 (3x3, stride 2,
//   6 -> 32 channels, FrozenBN, ReLU), one per stream (rgb, depth).
// The two-kernel form writes the normalised 8-channel fp32 input of both streams (315 MB per 16-frame step) only for the convolution
// to read it back; here a block stages the u8 pixels and the three encoding planes of its input window in LDS (normalised on the
// way, the zero padding as zeros) and every thread computes one output pixel x 32 channels for BOTH streams (they share the
// encoding planes).  HBM traffic: 88 MB of inputs + 315 MB of outputs per step.
// Arithmetic: exactly the MFMA path's.  v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain; the implicit GEMM packs k = tap * 8 +
// channel, cuts it into K-slices of 32 (taps 0-3 | 4-7 | 8) that start from zero, and adds the slice sums in order (two-level
// accumulation); inside a group of 8 k the kernel's four MFMAs multiply the pairs (0, 4), (1, 5), (2, 6), (3, 7) - lane half h supplies
// k = 4 h + e - so a tap's channels enter the chain in the order 0, 4, 1, 5, 2, 3 (6 and 7 are the zero channels of the padded input);
// the epilogue is fmaf(acc, scale, shift), max(., 0).  The same chains here, in vector FMAs (54 per output channel: the multiply-add
// count of this layer is 0.1 % of the network's), as packed fp32 FMAs (v_pk_fma_f32: two output channels per instruction, the pixel value broadcast, the filter pair a scalar-register
// operand fetched by scalar loads - the filter pointer is a `const __restrict__` kernel parameter for exactly that).
#include "common.h"

namespace quber {
namespace {

constexpr int TY = 8, TX = 32;                   // output pixels per block: one per thread
constexpr int IY = 2 * TY + 1, IX = 2 * TX + 1;  // input window (stride 2, 3x3, pad 1)

using f32x2 = __attribute__((ext_vector_type(2))) float;

struct StemP {
    const uint8_t* bgr; const uint8_t* depth; const float* offs;
    float* out; long out_gs;                     // [streams][B][OH][OW][32]
    int B, H, W, OH, OW, streams;
    float mean[6], stdv[6];
};

// wq: filters [streams][9 taps][6 channels][32] (output channel fastest); scq / shq: [streams][32]
// H16 (fp16 data path, compute_dtype 2 with fp16 tensors): the operands the MFMA kernel would see - the normalised input rounded to
// fp16, filters rounded to fp16 by the host - multiplied exactly and summed in fp32, the output rounded to fp16.  The fp16 MFMA sums
// the 16 products of a step in an order of its own, so this form agrees with the preprocess kernel + implicit GEMM to fp32
// summation order (a last-place difference of a few fp16 outputs), not bit for bit; it replaces a 16-channel fp16 input tensor
// (537 MB per 8-frame step at 1024x1024) and a GEMM that multiplies 10 zero channels per tap.
template <bool H16>
__global__ __launch_bounds__(256) void stem_conv1_kernel(const StemP p, const float* __restrict__ wq, const float* __restrict__ scq,
                                                         const float* __restrict__ shq) {
    __shared__ float sx[9][IY][IX + 1];          // planes: image 0 (3), image 1 (3), heat, off_y, off_x
    const int t = threadIdx.x;
    const int b = blockIdx.z;
    const int oy0 = blockIdx.y * TY, ox0 = blockIdx.x * TX;
    const int iy0 = 2 * oy0 - 1, ix0 = 2 * ox0 - 1;
    const long HW = (long)p.H * p.W;
    // ---- stage the window: u8 pixels normalised as model.py:138 does ((x - mean) / std, IEEE division), encoding planes as they are ----
    for (int i = t; i < IY * IX; i += 256) {
        const int r = i / IX, c = i - r * IX;
        const int y = iy0 + r, x = ix0 + c;
        const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        const long pix = (long)b * HW + (long)y * p.W + x;
        float v[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (ok) {
            const uint8_t* q = p.bgr + pix * 3;
            v[0] = ((float)q[0] - p.mean[0]) / p.stdv[0];
            v[1] = ((float)q[1] - p.mean[1]) / p.stdv[1];
            v[2] = ((float)q[2] - p.mean[2]) / p.stdv[2];
            if (p.streams == 2) {
                const uint8_t* d = p.depth + pix * 3;
                v[3] = ((float)d[0] - p.mean[3]) / p.stdv[3];
                v[4] = ((float)d[1] - p.mean[4]) / p.stdv[4];
                v[5] = ((float)d[2] - p.mean[5]) / p.stdv[5];
            }
            const float* o = p.offs + (long)b * 3 * HW + (long)y * p.W + x;
            v[6] = o[0]; v[7] = o[HW]; v[8] = o[2 * HW];
        }
#pragma unroll
        for (int e = 0; e < 9; ++e) sx[e][r][c] = H16 ? (float)(_Float16)v[e] : v[e];
    }
    __syncthreads();
    const int ty = t >> 5, tx = t & 31;
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy >= p.OH || ox >= p.OW) return;
    for (int g = 0; g < p.streams; ++g) {
        const float* __restrict__ w = wq + g * (9 * 6 * 32);
        f32x2 top[16], acc[16];
        // K-slices of the packed GEMM: taps 0-3, taps 4-7, tap 8 - each chain starts from zero (SrcC = 0), the sums are added in order
#pragma unroll
        for (int slice = 0; slice < 3; ++slice) {
#pragma unroll
            for (int tap = slice * 4; tap < (slice == 2 ? 9 : slice * 4 + 4); ++tap) {
                const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    const int ci = q == 0 ? 0 : q == 1 ? 4 : q == 2 ? 1 : q == 3 ? 5 : q == 4 ? 2 : 3;      // the MFMA's order inside a group of 8 k
                    const float x = sx[ci < 3 ? 3 * g + ci : 3 + ci][2 * ty + ky][2 * tx + kx];
                    const float* wr = w + (tap * 6 + ci) * 32;
                    const bool first = tap == slice * 4 && q == 0;
#pragma unroll
                    for (int o = 0; o < 16; ++o)
                        acc[o] = __builtin_elementwise_fma(f32x2{x, x}, f32x2{wr[2 * o], wr[2 * o + 1]}, first ? f32x2{0.f, 0.f} : acc[o]);
                }
            }
#pragma unroll
            for (int o = 0; o < 16; ++o) top[o] = slice == 0 ? f32x2{0.f, 0.f} + acc[o] : top[o] + acc[o];
        }
        const float* __restrict__ sc = scq + g * 32;
        const float* __restrict__ sh = shq + g * 32;
        if constexpr (H16) {
            using h16x8 = __attribute__((ext_vector_type(8))) _Float16;
            _Float16* dst = reinterpret_cast<_Float16*>(p.out) + (long)g * p.out_gs + (((long)b * p.OH + oy) * p.OW + ox) * 32;
#pragma unroll
            for (int o = 0; o < 16; o += 4) {
                h16x8 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    y[2 * e] = (_Float16)fmaxf(fmaf(top[o + e].x, sc[2 * (o + e)], sh[2 * (o + e)]), 0.f);
                    y[2 * e + 1] = (_Float16)fmaxf(fmaf(top[o + e].y, sc[2 * (o + e) + 1], sh[2 * (o + e) + 1]), 0.f);
                }
                *reinterpret_cast<h16x8*>(dst + 2 * o) = y;
            }
            continue;
        }
        float* dst = p.out + (long)g * p.out_gs + (((long)b * p.OH + oy) * p.OW + ox) * 32;
#pragma unroll
        for (int o = 0; o < 16; o += 2) {
            float4 y;
            y.x = fmaxf(fmaf(top[o].x, sc[2 * o], sh[2 * o]), 0.f);
            y.y = fmaxf(fmaf(top[o].y, sc[2 * o + 1], sh[2 * o + 1]), 0.f);
            y.z = fmaxf(fmaf(top[o + 1].x, sc[2 * o + 2], sh[2 * o + 2]), 0.f);
            y.w = fmaxf(fmaf(top[o + 1].y, sc[2 * o + 3], sh[2 * o + 3]), 0.f);
            *reinterpret_cast<float4*>(dst + 2 * o) = y;
        }
    }
}


// ---- fp16 data path: the same layer on the matrix pipe ----
// The vector-FMA form above costs 54 x 32 multiply-adds per pixel on the vector lanes: 0.35 ms of the 14.6 ms fp16 step at 1024 x 1024 x 8,
// four times what the layer's HBM traffic takes.  Here a block stages its input window in LDS as 16-byte pixels per stream
// {image c0 c1 c2 | heat off_y off_x | 0 0} in fp16 (the u8 normalisation (q - mean) / std, rounded to fp16, comes from a 6 x 256-entry
// table built by the block: the same values as the expression), and one ds_read_b128 per (pixel tile, k-step) IS the im2col
// fragment of v_mfma_f32_16x16x32_f16: lane (fr, fq) supplies pixel fr of the tile and filter tap 4 ks + fq (8 channels), k = tap * 8 +
// channel, three k-steps (taps 9-11 meet zero filters).  Filters are the row operand with the channel rows of the tile pair interleaved
// (conv_h8.hip): a lane ends up with 8 consecutive output channels of one pixel = one 16-byte store; the 12 filter fragments of both
// streams stay in registers (wf16: [stream][tile][k-step][lane][8] halfs, packed by the host).  Operands as in the vector form (fp16
// inputs, fp16 filters, exact products, fp32 sums); the sums are added in the MFMA's order: agreement to fp32 summation order.
using h16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__global__ __launch_bounds__(256) void stem_conv1_h16_kernel(const StemP p, const h16x8* __restrict__ wf16, const float* __restrict__ scq,
                                                             const float* __restrict__ shq) {
    constexpr int PX = IX + 1;                   // pixels per staged row (66: even, keeps the 16-byte pixels of a row pair apart by 33 x 32 bytes)
    __shared__ h16x8 sp[2][IY][PX];
    __shared__ _Float16 lut[6][256];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int b = blockIdx.z;
    const int oy0 = blockIdx.y * TY, ox0 = blockIdx.x * TX;
    const int iy0 = 2 * oy0 - 1, ix0 = 2 * ox0 - 1;
    const long HW = (long)p.H * p.W;
#pragma unroll
    for (int c = 0; c < 6; ++c) lut[c][t] = (_Float16)(((float)t - p.mean[c]) / p.stdv[c]);
    __syncthreads();
    for (int i = t; i < IY * IX; i += 256) {
        const int r = i / IX, c = i - r * IX;
        const int y = iy0 + r, x = ix0 + c;
        const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        h16x8 v0 = {0, 0, 0, 0, 0, 0, 0, 0}, v1 = v0;
        if (ok) {
            const long pix = (long)b * HW + (long)y * p.W + x;
            const uint8_t* q = p.bgr + pix * 3;
            const float* o = p.offs + (long)b * 3 * HW + (long)y * p.W + x;
            const _Float16 e0 = (_Float16)o[0], e1 = (_Float16)o[HW], e2 = (_Float16)o[2 * HW];
            v0[0] = lut[0][q[0]]; v0[1] = lut[1][q[1]]; v0[2] = lut[2][q[2]];
            v0[3] = e0; v0[4] = e1; v0[5] = e2;
            if (p.streams == 2) {
                const uint8_t* d = p.depth + pix * 3;
                v1[0] = lut[3][d[0]]; v1[1] = lut[4][d[1]]; v1[2] = lut[5][d[2]];
                v1[3] = e0; v1[4] = e1; v1[5] = e2;
            }
        }
        sp[0][r][c] = v0;
        sp[1][r][c] = v1;
    }
    __syncthreads();
    const int fr = lane & 15, fq = lane >> 4;
    // this wave: output rows 2 wave, 2 wave + 1 of the block, two 16-pixel tiles each
    for (int g = 0; g < p.streams; ++g) {
        h16x8 wf[2][3];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) wf[jj][ks] = wf16[((g * 2 + jj) * 3 + ks) * 64 + lane];
        f32x4 sc[2], sh[2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            sc[jj] = *reinterpret_cast<const f32x4*>(scq + g * 32 + 8 * fq + 4 * jj);
            sh[jj] = *reinterpret_cast<const f32x4*>(shq + g * 32 + 8 * fq + 4 * jj);
        }
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            const int ty = 2 * wave + (pt >> 1), tx = 16 * (pt & 1) + fr;
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                const int tap = ks == 2 ? 8 : 4 * ks + fq;      // (k-step 2: taps 9-11 have zero filters; any finite pixel does)
                const int ky = tap / 3, kx = tap - 3 * ky;
                const h16x8 pf = sp[g][2 * ty + ky][2 * tx + kx];
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0][ks], pf, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[1][ks], pf, acc[1], 0, 0, 0);
            }
            const int oy = oy0 + ty, ox = ox0 + tx;
            if (oy < p.OH && ox < p.OW) {
                h16x8 y;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[4 * jj + e] = (_Float16)fmaxf(fmaf(acc[jj][e], sc[jj][e], sh[jj][e]), 0.f);
                _Float16* dst = reinterpret_cast<_Float16*>(p.out) + (long)g * p.out_gs + (((long)b * p.OH + oy) * p.OW + ox) * 32 + 8 * fq;
                *reinterpret_cast<h16x8*>(dst) = y;
            }
        }
    }
}

}  // namespace

// w: [streams][9][6][32] device floats; out: NHWC [streams][Bcap][OH][OW][32] (group stride out_gs elements) of fp32 (es 4) or fp16 (es 2)
// wf16 (fp16 output only, may be null): the filters as MFMA fragments [streams][2 tiles][3 k-steps][64 lanes][8 halfs] -> the matrix-pipe kernel
int launch_stem_conv1(const uint8_t* bgr, const uint8_t* depth, const float* offs, int B, int H, int W, int streams, const float* mean6,
                      const float* std6, const float* w, const float* scale, const float* shift, float* out, long out_gs, int es, hipStream_t st,
                      const void* wf16) {
    if (!bgr || !offs || !w || !scale || !shift || !out || (streams == 2 && !depth)) return fail("stem: null argument");
    StemP p{};
    p.bgr = bgr; p.depth = depth; p.offs = offs; p.out = out; p.out_gs = out_gs;
    p.B = B; p.H = H; p.W = W; p.OH = (H + 1) / 2; p.OW = (W + 1) / 2; p.streams = streams;
    for (int i = 0; i < 6; ++i) { p.mean[i] = mean6[i]; p.stdv[i] = std6[i]; }
    const double px = (double)B * H * W, opx = (double)B * p.OH * p.OW;
    ProfScope prof("stem_fused", px * (3.0 * streams + 12.0) + opx * 32.0 * es * streams, 2.0 * opx * 54.0 * 32.0 * streams, st);
    const dim3 grid((p.OW + TX - 1) / TX, (p.OH + TY - 1) / TY, B);
    if (es == 2 && wf16 && (((uintptr_t)scale | (uintptr_t)shift | (uintptr_t)out | (uintptr_t)wf16) & 15) == 0 && out_gs % 8 == 0)
        hipLaunchKernelGGL(stem_conv1_h16_kernel, grid, dim3(256), 0, st, p, reinterpret_cast<const h16x8*>(wf16), scale, shift);
    else if (es == 2) hipLaunchKernelGGL(stem_conv1_kernel<true>, grid, dim3(256), 0, st, p, w, scale, shift);
    else hipLaunchKernelGGL(stem_conv1_kernel<false>, grid, dim3(256), 0, st, p, w, scale, shift);
    QB_CHECK(hipGetLastError());
    return 0;
}

}  // namespace quber
