// Winograd F(m x m, 3x3) path, m = 2, 4 or 6, for the wide 3x3 / stride 1 convolutions of the refiner (fp32 throughout).
//
// The reference computes these layers as plain convolutions (detectron2 Conv2d -> F.conv2d;
// maskrefiner/modeling/backbone/resnet.py:472-485 fusion_res*.conv0/1, resnet.py:441-447 res3-5 conv2, the
// DeepLabV3+ / head 3x3s of model.py:369-458).  The same sum is regrouped: an m x m output tile needs (m+2)^2
// multiplies per (cin, cout) pair instead of 9 m^2,
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A,       d = (m+2)^2 input patch, g = 3x3 filter,
// so the matrix pipe does 2.25x (m = 2), 4x (m = 4) or 5.06x (m = 6) less work.  Pipeline per layer, P = (m+2)^2:
//   1. wino_input_kernel   d -> V = B^T d B          NHWC input  -> V[g][P][tile][Cin]     (HBM-bound)
//   2. conv_igemm_f32      M[p] = V[p] x U[p]^T       P (x groups) independent GEMMs, launched as ONE grouped 1x1
//                                                      "convolution" over blockIdx.z (conv_igemm.hip, unchanged)
//   3. wino_output_kernel  Y = A^T M A, affine, ReLU  M[g][P][tile][Cout] -> NHWC output (any channel-slice view),
//                                                      GroupNorm sums of the stored values on request
// m = 2: all transform entries are in {0, +-1, +-1/2}; the result is as accurate as the direct kernel.
// m = 4 (interpolation points 0, +-3/4, +-3/2, inf - Lavin & Gray's 0, +-1, +-2 scaled by 3/4, which measured 2.7x less
// fp32 error): about half a decimal digit less accurate than the direct kernel
// (tests/test_gpu_parity.py::test_conv3x3_winograd_vs_float64 bounds it); see DESIGN.md section 4 for where each is used.
// m = 6 (points 0, +-1, +-2, +-1/2, inf): 64 GEMMs per layer, about a decimal digit less accurate than the direct
// kernel; taken only where the 6x6 tiles fit the map with little padding.
// With dilation d the layer is d*d independent dense convolutions on the phase sub-images in[d*Y + py][d*X + px]
// (TH x TW tiles each, the same for every phase; tiles beyond a shorter phase read zeros and store nothing).
#include "common.h"
#include "winograd_xf.h"

namespace quber {

using namespace wxf;


namespace {

// `np.stats` set: the input is the PRE-normalisation tensor of a GroupNorm + ReLU whose only consumer is this layer;
// the normalisation (same arithmetic as gn_apply_kernel) is applied to the in-range pixels as they are loaded, which
// saves the separate read + write pass over the activations.
// CV = C / V channel groups; a block holds 256 / CV tiles (CV <= 256) or a tile needs CV / 256 blocks (grid.y).
// waves per SIMD the transforms are compiled for (F(4x4) on 16-byte vectors: the input transform keeps a column's six loads - and,
// as far as the scheduler dares, the next column's - in flight: 182-205 registers, two waves; held to three waves (168) it spills 31
// registers and the 16-frame step loses 0.2 ms, profiles/r17_epilogue.md; the output transform's 36 loads in flight fill the
// register file of a single wave)
#ifndef QB_WINO_IN_WAVES44
#define QB_WINO_IN_WAVES44 2
#endif
constexpr int wino_in_waves(int O, int V) { return O == 4 ? (V == 4 ? QB_WINO_IN_WAVES44 : 5) : O == 2 ? 4 : 2; }
constexpr int wino_out_waves(int O, int V) { return O == 4 ? (V == 4 ? 2 : 3) : O == 2 ? 4 : 1; }

template <int O, int V, bool NORM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(wino_in_waves(O, V)))) void wino_input_kernel(const float* __restrict__ in, int B, int H, int W, int CV, int in_cs,
                                                         long in_gs, int TH, int TW, int d, float* __restrict__ v, long v_gs,
                                                         const WinoNorm np, int Ball, int boff) {
    constexpr int T = O + 2;
    using VT = Vec<V>;
    const int g = blockIdx.z;
    in += g * in_gs;
    v += g * v_gs;
    const int tpb = 256 / CV;
    const int cv = tpb ? threadIdx.x % CV : blockIdx.y * 256 + threadIdx.x;
    const long tiles = (long)B * d * d * TH * TW;
    // XCD-aware order: blocks b and b + 8 share an XCD (and its L2).  Every input pixel is read by up to (T/O)^2 = 2.25
    // tiles (the 2-pixel halo); with consecutive blocks on consecutive tiles the neighbours sit on other XCDs and every
    // halo read misses L2 (rocprofv3: 1.9x the input fetched from HBM).  Each XCD gets one contiguous run of tiles instead.
    const int nblk = gridDim.x, xcd = blockIdx.x & 7, bq = nblk >> 3, br = nblk & 7;
    const int vb = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + (blockIdx.x >> 3);
    const long tile = tpb ? (long)vb * tpb + threadIdx.x / CV : vb;
    if (tile >= tiles || cv >= CV || (tpb && (int)(threadIdx.x / CV) >= tpb)) return;
    const TileAt ta = locate(tile, TH, TW, d);
    const int c = cv * V;
    const float* base = in + (long)ta.b * H * W * in_cs + c;
    VT nsc, nbi;
    if constexpr (NORM) {
        const double* sb = np.stats + (((long)g * Ball + boff + ta.b) * np.groups + c / np.cpg) * 2;
        const double mean = sb[0] / np.n;
        double var = sb[1] / np.n - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)np.eps));
        const VT ga = vload<V>(np.gamma + g * np.param_gs + c), be = vload<V>(np.beta + g * np.param_gs + c);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            nsc.v[e] = rstd * ga.v[e];
            nbi.v[e] = be.v[e] - (float)mean * nsc.v[e];
        }
    }
    const float relu_lo = (NORM && np.relu) ? 0.f : -__builtin_inff();       // the fused norm's ReLU as max(v, lo)
    VT t[T][T];                                      // t = B^T d, one input column at a time
#pragma unroll
    for (int j = 0; j < T; ++j) {
        const int x = d * (O * ta.tx - 1 + j) + ta.px;
        VT col[T], tc[T];
        // No branch around a load: a pixel of the zero padding is read from the nearest pixel of the map (a line its neighbours read
        // anyway) and replaced by zero afterwards, and the six loads of a column are requested before the first is used.  With
        // `inside ? load : 0` every load sat in a branch of its own and the compiler waited for it there: 36 memory latencies in a row
        // per thread - at one frame the whole kernel (16 us per launch for 1-3 MB of traffic, profiles/r17_epilogue.md).
        const int xc = min(max(x, 0), W - 1);
        const bool xin = (unsigned)x < (unsigned)W;
#pragma unroll
        for (int i = 0; i < T; ++i) {
            const int y = d * (O * ta.ty - 1 + i) + ta.py;
            col[i] = vload<V>(base + ((long)min(max(y, 0), H - 1) * W + xc) * in_cs);
        }
#pragma unroll
        for (int i = 0; i < T; ++i) {
            const int y = d * (O * ta.ty - 1 + i) + ta.py;
            const bool inside = xin && (unsigned)y < (unsigned)H;
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float q = col[i].v[e];
                if constexpr (NORM) q = fmaxf(fmaf(q, nsc.v[e], nbi.v[e]), relu_lo);
                col[i].v[e] = inside ? q : 0.f;
            }
        }
        bt<O>(col, tc);
#pragma unroll
        for (int i = 0; i < T; ++i) t[i][j] = tc[i];
    }
    float* dst = v + tile * (long)(CV * V) + c;
    const long ps = tiles * (long)(CV * V);          // position stride
#pragma unroll
    for (int i = 0; i < T; ++i) {                    // (B^T d) B
        VT row[T];
        bt<O>(t[i], row);
#pragma unroll
        for (int j = 0; j < T; ++j) vstore<V>(dst + (i * T + j) * ps, row[j]);
    }
}

// A block handles `iters` (<= OUT_ITERS) groups of tiles - fewer when the launch would otherwise not fill the chip (one
// frame: 1200 tiles of 128 channels are 38 blocks at 4 groups each) - and with `gn_sum` it also accumulates the GroupNorm
// sums of what it stores (fp64, LDS per block, one global atomic per (image, group) per block - as the direct kernel's
// epilogue does).
constexpr int OUT_ITERS = 4;
template <int O, int V>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(wino_out_waves(O, V)))) void wino_output_kernel(const float* __restrict__ m, long m_gs, int B, int OH, int OW, int CV,
                                                          int TH, int TW, int d, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int ss_gs, int relu,
                                                          float* __restrict__ out, int out_cs, long out_gs,
                                                          double* __restrict__ gn_sum, int gn_groups, int gn_cpg, int Ball, int boff, int iters) {
    constexpr int T = O + 2;
    using VT = Vec<V>;
    const int g = blockIdx.z;
    m += g * m_gs;
    out += g * out_gs;
    const int tpb = 256 / CV;
    const int cv = tpb ? threadIdx.x % CV : blockIdx.y * 256 + threadIdx.x;
    const int c = cv * V;
    const long per_img = (long)d * d * TH * TW;
    const long tiles = (long)B * per_img;
    const int step = tpb ? tpb : 1;                  // tiles per iteration
    const long tile0 = (long)blockIdx.x * step * iters;
    const bool lane_ok = cv < CV && (!tpb || (int)(threadIdx.x / CV) < tpb);
    __shared__ double gacc[2 * 32 * 2];              // [image b0 / b0+1][group][sum, sum of squares]
    const int b0 = (int)(tile0 / per_img);
    if (gn_sum) {
        if (threadIdx.x < 128) gacc[threadIdx.x] = 0.0;
        __syncthreads();
    }
    double s0 = 0.0, q0 = 0.0, s1 = 0.0, q1 = 0.0;
    const long ps = tiles * (long)(CV * V);
    VT sc, sh;
    if (scale && lane_ok) {
        sc = vload<V>(scale + g * ss_gs + c);
        sh = vload<V>(shift + g * ss_gs + c);
    }
    for (int it = 0; it < iters; ++it) {
        const long tile = tile0 + (long)it * step + (tpb ? threadIdx.x / CV : 0);
        if (!lane_ok || tile >= tiles) continue;
        const TileAt ta = locate(tile, TH, TW, d);
        const float* src = m + tile * (long)(CV * V) + c;
        VT s[O][T];                                  // s = A^T M, one column of positions at a time
#pragma unroll
        for (int j = 0; j < T; ++j) {
            VT col[T], sj[O];
#pragma unroll
            for (int i = 0; i < T; ++i) col[i] = vload<V>(src + (i * T + j) * ps);
            at<O>(col, sj);
#pragma unroll
            for (int i = 0; i < O; ++i) s[i][j] = sj[i];
        }
        double a = 0.0, q = 0.0;
#pragma unroll
        for (int i = 0; i < O; ++i) {
            const int oy = d * (O * ta.ty + i) + ta.py;
            VT row[O];
            at<O>(s[i], row);                        // (A^T M) A
            if (oy >= OH) continue;
#pragma unroll
            for (int j = 0; j < O; ++j) {
                const int ox = d * (O * ta.tx + j) + ta.px;
                if (ox >= OW) continue;
                VT y = row[j];
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    if (scale) y.v[e] = fmaf(y.v[e], sc.v[e], sh.v[e]);
                    if (relu) y.v[e] = fmaxf(y.v[e], 0.f);
                    a += (double)y.v[e];
                    q += (double)y.v[e] * y.v[e];
                }
                vstore<V>(out + (((long)ta.b * OH + oy) * OW + ox) * out_cs + c, y);
            }
        }
        if (ta.b == b0) { s0 += a; q0 += q; } else { s1 += a; q1 += q; }
    }
    if (gn_sum) {
        if (lane_ok) {
            const int grp = c / gn_cpg;
            if (s0 != 0.0 || q0 != 0.0) { atomicAdd(&gacc[grp * 2], s0); atomicAdd(&gacc[grp * 2 + 1], q0); }
            if (s1 != 0.0 || q1 != 0.0) { atomicAdd(&gacc[64 + grp * 2], s1); atomicAdd(&gacc[64 + grp * 2 + 1], q1); }
        }
        __syncthreads();
        if (threadIdx.x < 128) {
            const double v = gacc[threadIdx.x];
            const int b = b0 + (threadIdx.x >> 6);
            if (v != 0.0 && b < B) atomicAdd(&gn_sum[(((long)g * Ball + boff + b) * gn_groups) * 2 + (threadIdx.x & 63)], v);
        }
    }
}

// one dimension of U = G g G^T, fp64
template <int O>
__host__ __device__ inline void gmul(double a, double b, double c, double* y) {
    if constexpr (O == 2) {
        y[0] = a;
        y[1] = 0.5 * (a + b + c);
        y[2] = 0.5 * (a - b + c);
        y[3] = c;
    } else if constexpr (O == 4) {
        y[0] = a * (64.0 / 81.0);
        y[1] = -a * (128.0 / 243.0) - b * (32.0 / 81.0) - c * (8.0 / 27.0);
        y[2] = -a * (128.0 / 243.0) + b * (32.0 / 81.0) - c * (8.0 / 27.0);
        y[3] = a * (32.0 / 243.0) + b * (16.0 / 81.0) + c * (8.0 / 27.0);
        y[4] = a * (32.0 / 243.0) - b * (16.0 / 81.0) + c * (8.0 / 27.0);
        y[5] = c;
    } else {
        y[0] = a;
        y[1] = -(a + b + c) * (2.0 / 9.0);
        y[2] = -(a - b + c) * (2.0 / 9.0);
        y[3] = a / 90.0 + b / 45.0 + c * (2.0 / 45.0);
        y[4] = a / 90.0 - b / 45.0 + c * (2.0 / 45.0);
        y[5] = (32.0 * a + 16.0 * b + 8.0 * c) / 45.0;
        y[6] = (32.0 * a - 16.0 * b + 8.0 * c) / 45.0;
        y[7] = c;
    }
}

// U[p = i*T+j] = (G g G^T)[i][j] of one 3x3 filter (row-major g9); computed in fp64 and rounded once by the caller
template <int O>
__host__ __device__ inline void wino_filter(const float* g9, double* u) {
    constexpr int T = O + 2;
    double t[T][3];
    for (int j = 0; j < 3; ++j) {
        double col[T];
        gmul<O>(g9[0 * 3 + j], g9[1 * 3 + j], g9[2 * 3 + j], col);
        for (int i = 0; i < T; ++i) t[i][j] = col[i];
    }
    for (int i = 0; i < T; ++i) gmul<O>(t[i][0], t[i][1], t[i][2], u + i * T);
}

template <int O>
__global__ void wino_weight_kernel(const float* __restrict__ w, int Cout, int Cin, float* __restrict__ u) {
    constexpr int P = (O + 2) * (O + 2);
    const long n = (long)Cout * Cin;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        double up[P];
        wino_filter<O>(w + i * 9, up);
        for (int p = 0; p < P; ++p) u[p * n + i] = (float)up[p];
    }
}

template <int O>
void weights_host(const float* w_oihw, int Cout, int Cin, float* u) {
    constexpr int P = (O + 2) * (O + 2);
    for (int o = 0; o < Cout; ++o)
        for (int c = 0; c < Cin; ++c) {
            double up[P];
            wino_filter<O>(w_oihw + ((size_t)o * Cin + c) * 9, up);
            for (int p = 0; p < P; ++p) u[((size_t)p * Cout + o) * Cin + c] = (float)up[p];
        }
}

}  // namespace

void winograd_weights_host(const float* w_oihw, int Cout, int Cin, int m, float* u) {
    if (m == 6) weights_host<6>(w_oihw, Cout, Cin, u);
    else if (m == 4) weights_host<4>(w_oihw, Cout, Cin, u);
    else weights_host<2>(w_oihw, Cout, Cin, u);
}

int launch_winograd_weights(const float* w_oihw, int Cout, int Cin, int m, float* u, hipStream_t st) {
    if (m == 6) hipLaunchKernelGGL(wino_weight_kernel<6>, dim3(256), dim3(256), 0, st, w_oihw, Cout, Cin, u);
    else if (m == 4) hipLaunchKernelGGL(wino_weight_kernel<4>, dim3(256), dim3(256), 0, st, w_oihw, Cout, Cin, u);
    else hipLaunchKernelGGL(wino_weight_kernel<2>, dim3(256), dim3(256), 0, st, w_oihw, Cout, Cin, u);
    QB_CHECK(hipGetLastError());
    return 0;
}

// m = 6 transforms work on channel pairs: C / 2 threads per tile
bool winograd_m6_channels_ok(int Cin, int Cout) {
    return (Cin / 2 <= 256 || (Cin / 2) % 256 == 0) && (Cout / 2 <= 256 || (Cout / 2) % 256 == 0);
}

bool winograd_eligible(int k, int stride, int pad, int dil, int Cin, int Cout) {
    return k == 3 && stride == 1 && dil >= 1 && pad == dil && Cin % 32 == 0 && Cout % 4 == 0 && Cin >= tune().wino_min_cin &&
           Cout >= tune().wino_min_cout && (Cin / 4 <= 256 || (Cin / 4) % 256 == 0) && (Cout / 4 <= 256 || (Cout / 4) % 256 == 0);
}

size_t winograd_ws_floats(int B, int H, int W, int Cin, int Cout, int G, int dil, int m) {
    const size_t tiles = (size_t)B * wino_tiles(H, W, dil, m);
    return (size_t)G * (m + 2) * (m + 2) * tiles * (size_t)(Cin + Cout);
}

// executed multiplies relative to the direct kernel: (m+2)^2 per tile against 9 per REAL output; the padded tiles of
// ragged frames and of a dilated layer's short phases eat into the 2.25x / 4x
double winograd_mac_ratio(int H, int W, int dil, int m) {
    return (double)((m + 2) * (m + 2)) * wino_tiles(H, W, dil, m) / (9.0 * H * W);
}


template <int O, int V>     // V: channels per thread in the transforms
static int run_winograd(const WinoP& q, int Ball, int G, hipStream_t st) {
    constexpr int P = (O + 2) * (O + 2);
    const View& in = q.in;
    const View& out = q.out;
    const int H = in.H, W = in.W, Cin = in.C, Cout = out.C, d = q.dil;
    const int TH = tiles_1d(H, d, O), TW = tiles_1d(W, d, O);
    const long tiles_pf = wino_tiles(H, W, d, O);
    if ((long)Ball * tiles_pf * P >= (1L << 31) / 2) return fail("winograd: too many tiles");
    // Frames per pass.  The intermediates V | M of a pass are rewritten in place by the next one; kept below the Infinity
    // Cache (256 MiB) they are written and re-read on the die instead of through HBM (tools/wino_subbatch_probe.py).
    int cb = Ball;
    if (tune().wino_chunk_mb > 0) {
        const double per_frame = 4.0 * G * P * (double)tiles_pf * (Cin + Cout);
        cb = (int)((double)tune().wino_chunk_mb * 1048576.0 / per_frame);
        if (cb < 1) cb = 1;
        if (cb > Ball) cb = Ball;
        cb = (Ball + (Ball + cb - 1) / cb - 1) / ((Ball + cb - 1) / cb);     // equal passes
    }
    if (winograd_ws_floats(cb, H, W, Cin, Cout, G, d, O) > q.ws_floats) return fail("winograd: workspace too small");
    WinoNorm np = q.norm;
    if (np.stats) {
        if (np.groups <= 0 || Cin % np.groups || (Cin / np.groups) % 4) return fail("winograd: fused GroupNorm needs 4 | channels per group");
        np.cpg = Cin / np.groups;
        np.n = (double)H * W * np.cpg;
    }
    // GroupNorm sums in the output transform when a block's tiles meet at most two images and vectors stay inside a group
    const int CVo = Cout / V, per_iter = CVo <= 256 ? 256 / CVo : 1;
    // groups of tiles per block of the output transform: 4, fewer while the grid would stay below ~8 blocks per CU
    int iters = OUT_ITERS;
    {
        const long groups = CVo <= 256 ? ((long)Ball * tiles_pf + per_iter - 1) / per_iter : (long)Ball * tiles_pf * (CVo / 256);
        while (iters > 1 && groups * G / iters < 2048) iters >>= 1;
    }
    const bool gn_here = q.gn_sum && q.gn_groups > 0 && q.gn_groups <= 32 && (Cout / q.gn_groups) % 4 == 0 &&
                         tiles_pf >= (long)per_iter * iters;
    for (int b0 = 0; b0 < Ball; b0 += cb) {
        const int B = Ball - b0 < cb ? Ball - b0 : cb;
        const long tiles = (long)B * tiles_pf;
        float* v = q.ws;
        auto grid = [&](int CV) {
            return CV <= 256 ? dim3((unsigned)((tiles + 256 / CV - 1) / (256 / CV)), 1, G) : dim3((unsigned)tiles, CV / 256, G);
        };
        const float* in_p = in.p + (size_t)b0 * H * W * in.cs;
        float* out_p = out.p + (size_t)b0 * H * W * out.cs;
        // Several groups reading ONE input (the heads of a hierarchy level: in.gs == 0): V is computed once and every
        // group's GEMMs read it (was: G identical copies written and re-read - 1 GB per step for the three heads)
        const bool shared_v = G > 1 && in.gs == 0 && !np.stats;
        const int Gv = shared_v ? 1 : G;
        float* m = q.ws + (size_t)Gv * P * tiles * Cin;
        auto grid_v = [&](int CV) {
            return CV <= 256 ? dim3((unsigned)((tiles + 256 / CV - 1) / (256 / CV)), 1, Gv) : dim3((unsigned)tiles, CV / 256, Gv);
        };
        {   // activations in once, V = P / m^2 times their size out
            ProfScope prof("wino_input", 4.0 * Gv * Cin * ((double)B * H * W + (double)P * tiles), 0.0, st);
            if (np.stats)
                hipLaunchKernelGGL((wino_input_kernel<O, V, true>), grid_v(Cin / V), dim3(256), 0, st, in_p, B, H, W, Cin / V, in.cs, in.gs, TH,
                                   TW, d, v, (long)P * tiles * Cin, np, Ball, b0);
            else
                hipLaunchKernelGGL((wino_input_kernel<O, V, false>), grid_v(Cin / V), dim3(256), 0, st, in_p, B, H, W, Cin / V, in.cs, in.gs, TH,
                                   TW, d, v, (long)P * tiles * Cin, np, Ball, b0);
        }
        QB_CHECK(hipGetLastError());
        ConvP p{};
        p.in = v; p.w = q.u; p.out = m;
        p.w3 = q.u3; p.w3_plane = q.u3_plane;          // bf16x3 mode: the same filters as three bf16 planes (conv_x8.hip)
        p.B = 1; p.H = (int)tiles; p.W = 1; p.Cin = Cin; p.in_cs = Cin;
        p.OH = (int)tiles; p.OW = 1; p.Cout = Cout; p.out_cs = Cout;
        p.K = Cin; p.Kpad = Cin;
        p.kh = 1; p.kw = 1; p.stride = 1; p.pad = 0; p.dil = 1;
        p.M = (int)tiles; p.ohw = (int)tiles;
        p.in_gs = tiles * Cin; p.out_gs = tiles * Cout; p.w_gs = (long)Cout * Cin;
        p.ws = q.splitk_ws; p.ws_floats = q.splitk_floats;
        p.tag = "wino_gemm";
        p.bf16 = q.dtype == 3 ? 3 : 0;
        if (shared_v) {
            for (int g = 0; g < G; ++g) {          // P position-GEMMs per group, all on the one V
                p.w = q.u + (size_t)g * P * Cout * Cin;
                if (q.u3) p.w3 = reinterpret_cast<const unsigned short*>(q.u3) + (size_t)g * P * Cout * Cin;
                p.out = m + (size_t)g * P * tiles * Cout;
                const int rc = launch_conv(p, P, st);
                if (rc) return rc;
            }
        } else {
            const int rc = launch_conv(p, G * P, st);
            if (rc) return rc;
        }
        dim3 og = grid(CVo);
        og.x = (og.x + iters - 1) / iters;
        {   // M in once, the layer's output out once
            ProfScope prof("wino_output", 4.0 * G * Cout * ((double)P * tiles + (double)B * H * W), 0.0, st);
            hipLaunchKernelGGL((wino_output_kernel<O, V>), og, dim3(256), 0, st, m, (long)P * tiles * Cout, B, H, W, CVo, TH, TW, d,
                               q.scale, q.shift, q.ss_gs, q.relu, out_p, out.cs, out.gs, gn_here ? q.gn_sum : nullptr, q.gn_groups,
                               q.gn_groups ? Cout / q.gn_groups : 1, Ball, b0, iters);
        }
        QB_CHECK(hipGetLastError());
    }
    if (q.gn_sum && !gn_here) return launch_gn_stats(out, Ball, G, q.gn_groups, q.gn_sum, st, false);
    return 0;
}

int launch_conv_winograd(const WinoP& q, int B, int G, hipStream_t st) {
    const View& in = q.in;
    const View& out = q.out;
    if (!winograd_eligible(3, 1, q.dil, q.dil, in.C, out.C) || out.H != in.H || out.W != in.W || (q.m != 2 && q.m != 4 && q.m != 6))
        return fail("winograd: unsupported geometry");
    if (q.m == 6 && !winograd_m6_channels_ok(in.C, out.C)) return fail("winograd: channel counts unsupported by the 6x6 variant");
    if (in.cs % 4 || out.cs % 4 || ((uintptr_t)in.p & 15) || ((uintptr_t)out.p & 15) || (in.gs & 3) || (out.gs & 3))
        return fail("winograd: operands must be 16-byte aligned");
    if (q.algo == 2 || (q.algo == 0 && winograd_fused_ok(q, B, G))) return launch_conv_winograd_fused(q, B, G, st);
    if (q.m == 4 && tune().wino_pairs && winograd_m6_channels_ok(in.C, out.C)) return run_winograd<4, 2>(q, B, G, st);
    return q.m == 6 ? run_winograd<6, 2>(q, B, G, st) : q.m == 4 ? run_winograd<4, 4>(q, B, G, st) : run_winograd<2, 4>(q, B, G, st);
}

}  // namespace quber
