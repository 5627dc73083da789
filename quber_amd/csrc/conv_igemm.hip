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

constexpr int BK = 32;
constexpr int PITCH = 36;

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_f32(const ConvP p) {
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int AL = BM / 32;  // float4 loads per thread per K-slice (A)
    constexpr int BL = BN / 32;  // (B)
    static_assert(WM * WN == 4, "4 waves");
    __shared__ __attribute__((aligned(16))) float As[BM * PITCH];
    __shared__ __attribute__((aligned(16))) float Bs[BN * PITCH];

    const int t = threadIdx.x;
    const int g = blockIdx.z;

    // XCD-aware tile order: blocks b and b+8 share an XCD (and its L2); give every XCD one
    // contiguous run of tiles, n-tile fastest, so neighbouring tiles reuse the same input rows.
    int tile;
    {
        const int bid = blockIdx.x, nblk = gridDim.x;
        const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int nt = tile % p.ntiles;
    const int mt = tile / p.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;

    const float* __restrict__ in = p.in + (long)g * p.in_gs;
    const float* __restrict__ wt = p.w + (long)g * p.w_gs;

    // ---- loader state (loop invariant) ----
    const int kq = (t & 7) * 4;
    const int lrow = t >> 3;
    int iy0[AL], ix0[AL], pbase[AL];
#pragma unroll
    for (int i = 0; i < AL; ++i) {
        const int m = m0 + lrow + 32 * i;
        if (m < p.M) {
            const int ohw = p.OH * p.OW;
            const int b = m / ohw;
            const int rem = m - b * ohw;
            const int oy = rem / p.OW;
            const int ox = rem - oy * p.OW;
            iy0[i] = oy * p.stride - p.pad;
            ix0[i] = ox * p.stride - p.pad;
            pbase[i] = b * p.H * p.W;
        } else {
            iy0[i] = -(1 << 28);
            ix0[i] = 0;
            pbase[i] = 0;
        }
    }
    const float* wrow[BL];
    bool wok[BL];
#pragma unroll
    for (int i = 0; i < BL; ++i) {
        const int n = n0 + lrow + 32 * i;
        wok[i] = n < p.Cout;
        wrow[i] = wt + (long)(wok[i] ? n : 0) * p.Kpad + kq;
    }

    float4 ra[AL], rb[BL];
    auto gload = [&](int kt) {
        const int k = kt * BK + kq;
        int dy = 0, dx = 0, c = 0;
        const bool kok = k < p.K;
        if (kok) {
            const int tap = k / p.Cin;
            c = k - tap * p.Cin;
            const int ky = tap / p.kw;
            dy = ky * p.dil;
            dx = (tap - ky * p.kw) * p.dil;
        }
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            const int iy = iy0[i] + dy, ix = ix0[i] + dx;
            const bool ok = kok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            if (ok) {
                const long off = ((long)(pbase[i] + iy * p.W + ix)) * p.in_cs + c;
                ra[i] = *reinterpret_cast<const float4*>(in + off);
            } else {
                ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int i = 0; i < BL; ++i) {
            if (wok[i])
                rb[i] = *reinterpret_cast<const float4*>(wrow[i] + kt * BK);
            else
                rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < AL; ++i)
            *reinterpret_cast<float4*>(&As[(lrow + 32 * i) * PITCH + kq]) = ra[i];
#pragma unroll
        for (int i = 0; i < BL; ++i)
            *reinterpret_cast<float4*>(&Bs[(lrow + 32 * i) * PITCH + kq]) = rb[i];
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

    const int nk = p.Kpad / BK;
    gload(0);
    lstore();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) gload(kt + 1);
        const float* ap = &As[(wm * TM * 32 + r) * PITCH + 4 * h];
        const float* bp = &Bs[(wn * TN * 32 + r) * PITCH + 4 * h];
#pragma unroll
        for (int ks = 0; ks < BK / 8; ++ks) {
            float4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const float4*>(ap + i * 32 * PITCH + ks * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const float4*>(bp + j * 32 * PITCH + ks * 8);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
        if (kt + 1 < nk) {
            lstore();
            __syncthreads();
        }
    }

    // ---- epilogue: y = acc*scale + shift (+ residual) (ReLU) ----
    float* __restrict__ out = p.out + (long)g * p.out_gs;
    const float* __restrict__ res = p.res ? p.res + (long)g * p.res_gs : nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + r;
        const bool nok = n < p.Cout;
        float sc = 1.f, sh = 0.f;
        if (nok) {
            if (p.scale) sc = p.scale[g * p.ss_gs + n];
            if (p.shift) sh = p.shift[g * p.ss_gs + n];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (nok && m < p.M) {
                    float v = fmaf(acc[i][j][e], sc, sh);
                    if (res) v += res[(long)m * p.res_cs + n];
                    if (p.relu) v = fmaxf(v, 0.f);
                    out[(long)m * p.out_cs + n] = v;
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
static int run(ConvP p, int G, hipStream_t st) {
    p.mtiles = (p.M + BM - 1) / BM;
    p.ntiles = (p.Cout + BN - 1) / BN;
    dim3 grid(p.mtiles * p.ntiles, 1, G);
    hipLaunchKernelGGL((conv_igemm_f32<BM, BN, WM, WN>), grid, dim3(256), 0, st, p);
    QB_CHECK(hipGetLastError());
    return 0;
}

int launch_conv(const ConvP& p, int G, hipStream_t st) {
    if (p.Cin % 4 || p.in_cs % 4 || p.Kpad % BK || p.K > p.Kpad)
        return fail("conv: Cin / channel stride must be multiples of 4 and Kpad a multiple of 32");
    if (((uintptr_t)p.in & 15) || ((uintptr_t)p.w & 15) || (p.in_gs & 3) || (p.w_gs & 3))
        return fail("conv: operands must be 16-byte aligned");
    if (p.M <= 0 || p.Cout <= 0) return fail("conv: empty problem");
    if (p.Cout <= 32) return run<256, 32, 4, 1>(p, G, st);
    if (p.Cout <= 64) return run<256, 64, 4, 1>(p, G, st);
    const long tiles128 = (long)((p.M + 127) / 128) * ((p.Cout + 127) / 128) * G;
    if (tiles128 < 512) return run<64, 64, 2, 2>(p, G, st);
    return run<128, 128, 2, 2>(p, G, st);
}

}  // namespace quber
