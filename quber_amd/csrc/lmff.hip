// Element-wise / pooling / depthwise kernels of the LMFFNet foreground network (reference
// foreground_segmentation/lmffnet.py) and of the refiner's post-filter (eval/refiner_model.py:273-277).
// The network is ~3 GFLOP per 640x480 frame and its channel counts (38, 134, 262, 26, 3) are not multiples of 4,
// so these kernels are plain one-element-per-lane NHWC loops over (pixel, channel); the dense convolutions go
// through conv_igemm_f32 with a BN + PReLU epilogue.
#include "common.h"

namespace quber {

static inline int grid_for(long work) {
    long g = (work + 255) / 256;
    if (g < 1) g = 1;
    if (g > 4096) g = 4096;
    return (int)g;
}

// predictor.py:79-83: BGR standardised with the RGB statistics in float64 (numpy), depth / 255 in float32
__global__ void lmff_preprocess_kernel(const uint8_t* __restrict__ bgr, const uint8_t* __restrict__ depth, long pixels,
                                       float* __restrict__ x) {
    const double mean[3] = {0.485, 0.456, 0.406}, sd[3] = {0.229, 0.224, 0.225};
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < pixels; i += (long)gridDim.x * blockDim.x) {
        float v[8];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            v[c] = (float)(((double)bgr[3 * i + c] / 255.0 - mean[c]) / sd[c]);
            v[3 + c] = (float)depth[3 * i + c] / 255.f;
        }
        v[6] = 0.f; v[7] = 0.f;
        float4* dst = reinterpret_cast<float4*>(x + i * 8);
        dst[0] = make_float4(v[0], v[1], v[2], v[3]);
        dst[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
}

int launch_lmff_preprocess(const uint8_t* bgr, const uint8_t* depth, long pixels, float* x, hipStream_t st) {
    hipLaunchKernelGGL(lmff_preprocess_kernel, dim3(grid_for(pixels)), dim3(256), 0, st, bgr, depth, pixels, x);
    QB_CHECK(hipGetLastError());
    return 0;
}

// depthwise 3x3 (dilation d, padding d) + folded BN + PReLU.  V channels per thread (V = 4: 16-byte accesses when the
// channel count and strides allow; V = 1 otherwise, e.g. the 262-channel MAD layer); 32-bit index arithmetic.
template <int V>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W,
                                                        int C, int in_cs, int out_cs, int dil, const float* __restrict__ w9,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        const float* __restrict__ slope) {
    const unsigned CV = C / V, total = (unsigned)B * H * W * CV;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned cv = i % CV;
        unsigned pix = i / CV;
        const int x = pix % W;
        pix /= W;
        const int y = pix % H;
        const int b = pix / H;
        const int c = cv * V;
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = y + (ky - 1) * dil;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = x + (kx - 1) * dil;
                if ((unsigned)ix >= (unsigned)W) continue;
                const float* src = in + ((long)(b * H + iy) * W + ix) * in_cs + c;
                if constexpr (V == 4) {
                    const float4 v = *reinterpret_cast<const float4*>(src);
                    acc[0] = fmaf(v.x, w9[(c + 0) * 9 + ky * 3 + kx], acc[0]);
                    acc[1] = fmaf(v.y, w9[(c + 1) * 9 + ky * 3 + kx], acc[1]);
                    acc[2] = fmaf(v.z, w9[(c + 2) * 9 + ky * 3 + kx], acc[2]);
                    acc[3] = fmaf(v.w, w9[(c + 3) * 9 + ky * 3 + kx], acc[3]);
                } else {
                    acc[0] = fmaf(src[0], w9[c * 9 + ky * 3 + kx], acc[0]);
                }
            }
        }
        float o[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float v = fmaf(acc[e], scale[c + e], shift[c + e]);
            o[e] = v > 0.f ? v : v * slope[c + e];
        }
        float* dst = out + ((long)(b * H + y) * W + x) * out_cs + c;
        if constexpr (V == 4) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
        else dst[0] = o[0];
    }
}

static inline bool vec4_ok(const View& v) { return v.C % 4 == 0 && v.cs % 4 == 0 && ((uintptr_t)v.p & 15) == 0; }

int launch_dwconv3x3(const View& in, const View& out, int B, int dil, const float* w9, const float* scale,
                     const float* shift, const float* slope, hipStream_t st) {
    const long total = (long)B * in.H * in.W * in.C;
    if (total >= (1L << 31)) return fail("dwconv: tensor too large");
    if (vec4_ok(in) && vec4_ok(out))
        hipLaunchKernelGGL(dwconv3x3_kernel<4>, dim3(grid_for(total / 4)), dim3(256), 0, st, in.p, out.p, B, in.H, in.W, in.C,
                           in.cs, out.cs, dil, w9, scale, shift, slope);
    else
        hipLaunchKernelGGL(dwconv3x3_kernel<1>, dim3(grid_for(total)), dim3(256), 0, st, in.p, out.p, B, in.H, in.W, in.C, in.cs,
                           out.cs, dil, w9, scale, shift, slope);
    QB_CHECK(hipGetLastError());
    return 0;
}

// mode 0: AvgPool2d(3, stride 2, padding 1), count_include_pad (divide by 9);  mode 1: MaxPool2d(2, stride 2)
template <int V>
__global__ __launch_bounds__(256) void pool_s2_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W,
                                                      int C, int in_cs, int OH, int OW, int out_cs, int mode) {
    const unsigned CV = C / V, total = (unsigned)B * OH * OW * CV;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const int c = (i % CV) * V;
        unsigned pix = i / CV;
        const int ox = pix % OW;
        pix /= OW;
        const int oy = pix % OH;
        const int b = pix / OH;
        float r[V];
#pragma unroll
        for (int e = 0; e < V; ++e) r[e] = mode == 0 ? 0.f : -INFINITY;
        const int lo = mode == 0 ? -1 : 0, hi = 1;
        for (int dy = lo; dy <= hi; ++dy)
            for (int dx = lo; dx <= hi; ++dx) {
                const int iy = 2 * oy + dy, ix = 2 * ox + dx;
                if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) continue;   // (never out of range for the max pool)
                const float* src = in + ((long)(b * H + iy) * W + ix) * in_cs + c;
                float v[V];
                if constexpr (V == 4) {
                    const float4 t = *reinterpret_cast<const float4*>(src);
                    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
                } else {
                    v[0] = src[0];
                }
#pragma unroll
                for (int e = 0; e < V; ++e) r[e] = mode == 0 ? r[e] + v[e] : fmaxf(r[e], v[e]);
            }
        if (mode == 0) {
#pragma unroll
            for (int e = 0; e < V; ++e) r[e] = r[e] / 9.f;
        }
        float* dst = out + ((long)(b * OH + oy) * OW + ox) * out_cs + c;
        if constexpr (V == 4) *reinterpret_cast<float4*>(dst) = make_float4(r[0], r[1], r[2], r[3]);
        else dst[0] = r[0];
    }
}

int launch_pool_s2(const View& in, const View& out, int B, int mode, hipStream_t st) {
    const long total = (long)B * out.H * out.W * in.C;
    if (total >= (1L << 31)) return fail("pool: tensor too large");
    if (mode == 1 && (in.H < 2 * out.H || in.W < 2 * out.W)) return fail("pool: max pool window leaves the map");
    if (vec4_ok(in) && out.cs % 4 == 0 && ((uintptr_t)out.p & 15) == 0)
        hipLaunchKernelGGL(pool_s2_kernel<4>, dim3(grid_for(total / 4)), dim3(256), 0, st, in.p, out.p, B, in.H, in.W, in.C,
                           in.cs, out.H, out.W, out.cs, mode);
    else
        hipLaunchKernelGGL(pool_s2_kernel<1>, dim3(grid_for(total)), dim3(256), 0, st, in.p, out.p, B, in.H, in.W, in.C, in.cs,
                           out.H, out.W, out.cs, mode);
    QB_CHECK(hipGetLastError());
    return 0;
}

// y = prelu((a [+ b]) * scale + shift)   (BNPReLU on a concatenation, and SEM_B's bn_relu_1(output + input))
template <int V>
__global__ __launch_bounds__(256) void affine_prelu_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           float* __restrict__ out, unsigned pixels, int C, int a_cs, int b_cs,
                                                           int out_cs, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const float* __restrict__ slope) {
    const unsigned CV = C / V, total = pixels * CV;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned pix = i / CV;
        const int c = (int)(i - pix * CV) * V;
        float v[V];
        if constexpr (V == 4) {
            const float4 t = *reinterpret_cast<const float4*>(a + (long)pix * a_cs + c);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            if (b) {
                const float4 u = *reinterpret_cast<const float4*>(b + (long)pix * b_cs + c);
                v[0] += u.x; v[1] += u.y; v[2] += u.z; v[3] += u.w;
            }
        } else {
            v[0] = a[(long)pix * a_cs + c];
            if (b) v[0] += b[(long)pix * b_cs + c];
        }
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float y = fmaf(v[e], scale[c + e], shift[c + e]);
            v[e] = y > 0.f ? y : y * slope[c + e];
        }
        float* dst = out + (long)pix * out_cs + c;
        if constexpr (V == 4) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        else dst[0] = v[0];
    }
}

int launch_affine_prelu(const View& a, const View* b, const View& out, int B, const float* scale, const float* shift,
                        const float* slope, hipStream_t st) {
    const long pixels = (long)B * a.H * a.W;
    if (pixels * a.C >= (1L << 31)) return fail("affine: tensor too large");
    const bool vec = vec4_ok(a) && vec4_ok(out) && (!b || (b->cs % 4 == 0 && ((uintptr_t)b->p & 15) == 0));
    if (vec)
        hipLaunchKernelGGL(affine_prelu_kernel<4>, dim3(grid_for(pixels * a.C / 4)), dim3(256), 0, st, a.p, b ? b->p : nullptr,
                           out.p, (unsigned)pixels, a.C, a.cs, b ? b->cs : 0, out.cs, scale, shift, slope);
    else
        hipLaunchKernelGGL(affine_prelu_kernel<1>, dim3(grid_for(pixels * a.C)), dim3(256), 0, st, a.p, b ? b->p : nullptr, out.p,
                           (unsigned)pixels, a.C, a.cs, b ? b->cs : 0, out.cs, scale, shift, slope);
    QB_CHECK(hipGetLastError());
    return 0;
}

// PMCA (lmffnet.py:172-191): w = sigmoid(fc2(prelu(fc0(dw2x2(adaptive_pool_2x2(x)) + global_pool(x))))).
// Stage 1 (pmca_sums_kernel): per (frame, channel) the four adaptive-pool quadrant sums - adaptive bins of a 2-way split are
// [0, ceil(n/2)) and [floor(n/2), n), they overlap by one row / column for odd n - over a grid of row bands: a thread
// keeps one 16-byte channel column, fp64 partial sums, LDS reduce, one atomic per (channel, quadrant) per block.
// The global pool is the sum of the quadrants minus the overlaps, so it is accumulated as a fifth sum.
// Stage 2 (pmca_fc_kernel): the two tiny fully connected layers, one block per frame.
__global__ __launch_bounds__(256) void pmca_sums_kernel(const float* __restrict__ x, int H, int W, int C, int cs, int rows_per_block,
                                                        double* __restrict__ sums) {
    __shared__ double acc[128 * 5];
    const int b = blockIdx.y, t = threadIdx.x;
    const int C4 = C >> 2, lanes = 256 / C4;              // C in {64, 128}: 16 or 8 pixel lanes per channel column
    const int c4 = t % C4, lane = t / C4;
    for (int i = t; i < C * 5; i += 256) acc[i] = 0.0;
    __syncthreads();
    const int hy1 = (H + 1) / 2, ly1 = H / 2, hx1 = (W + 1) / 2, lx1 = W / 2;
    const int y0 = blockIdx.x * rows_per_block, y1 = min(H, y0 + rows_per_block);
    const float* base = x + (long)b * H * W * cs + c4 * 4;
    double q[5][4];
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) q[k][e] = 0.0;
    for (int y = y0; y < y1; ++y) {
        const bool t0 = y < hy1, t1 = y >= ly1;
        double l[4] = {0, 0, 0, 0}, r[4] = {0, 0, 0, 0}, a[4] = {0, 0, 0, 0};     // left bin, right bin, whole row
        for (int xx = lane; xx < W; xx += lanes) {
            const float4 v = *reinterpret_cast<const float4*>(base + ((long)y * W + xx) * cs);
            const double d[4] = {(double)v.x, (double)v.y, (double)v.z, (double)v.w};
            const bool l0 = xx < hx1, l1 = xx >= lx1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[e] += d[e];
                if (l0) l[e] += d[e];
                if (l1) r[e] += d[e];
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            q[4][e] += a[e];
            if (t0) { q[0][e] += l[e]; q[1][e] += r[e]; }
            if (t1) { q[2][e] += l[e]; q[3][e] += r[e]; }
        }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (q[k][e] != 0.0) atomicAdd(&acc[(c4 * 4 + e) * 5 + k], q[k][e]);
    __syncthreads();
    for (int i = t; i < C * 5; i += 256)
        if (acc[i] != 0.0) atomicAdd(&sums[(long)b * C * 5 + i], acc[i]);
}

__global__ __launch_bounds__(128) void pmca_fc_kernel(const double* __restrict__ sums, int H, int W, int C,
                                                      const float* __restrict__ w2x2, const float* __restrict__ fc0,
                                                      const float* __restrict__ alpha, const float* __restrict__ fc2,
                                                      float* __restrict__ wts) {
    __shared__ float s[128], hidden[16];
    const int b = blockIdx.x, t = threadIdx.x;
    const int hy1 = (H + 1) / 2, ly1 = H / 2, hx1 = (W + 1) / 2, lx1 = W / 2;
    if (t < C) {
        const double* a = sums + ((long)b * C + t) * 5;
        const double n00 = (double)hy1 * hx1, n01 = (double)hy1 * (W - lx1), n10 = (double)(H - ly1) * hx1,
                     n11 = (double)(H - ly1) * (W - lx1);
        const float p00 = (float)(a[0] / n00), p01 = (float)(a[1] / n01), p10 = (float)(a[2] / n10), p11 = (float)(a[3] / n11);
        float o1 = 0.f;
        o1 = fmaf(p00, w2x2[t * 4 + 0], o1);
        o1 = fmaf(p01, w2x2[t * 4 + 1], o1);
        o1 = fmaf(p10, w2x2[t * 4 + 2], o1);
        o1 = fmaf(p11, w2x2[t * 4 + 3], o1);
        s[t] = o1 + (float)(a[4] / ((double)H * W));
    }
    __syncthreads();
    const int R = C / 8;
    if (t < R) {
        float h = 0.f;
        for (int k = 0; k < C; ++k) h = fmaf(s[k], fc0[t * C + k], h);
        hidden[t] = h > 0.f ? h : h * alpha[0];
    }
    __syncthreads();
    if (t < C) {
        float o = 0.f;
        for (int k = 0; k < R; ++k) o = fmaf(hidden[k], fc2[t * R + k], o);
        wts[(long)b * C + t] = 1.f / (1.f + expf(-o));
    }
}

int launch_pmca(const View& x, int B, const float* w2x2, const float* fc0, const float* alpha, const float* fc2, float* wts,
                double* sums, hipStream_t st) {
    if (x.C != 64 && x.C != 128) return fail("pmca: expects 64 or 128 channels");
    if (x.cs % 4 || ((uintptr_t)x.p & 15)) return fail("pmca: channels must come in aligned groups of 4");
    int rc = launch_zero(sums, sizeof(double) * (size_t)B * x.C * 5, st);
    if (rc) return rc;
    // row bands: about 1000 blocks in flight, at least one row each
    int rpb = (int)(((long)x.H * B + 1023) / 1024);
    if (rpb < 1) rpb = 1;
    hipLaunchKernelGGL(pmca_sums_kernel, dim3((x.H + rpb - 1) / rpb, B), dim3(256), 0, st, x.p, x.H, x.W, x.C, x.cs, rpb, sums);
    hipLaunchKernelGGL(pmca_fc_kernel, dim3(B), dim3(128), 0, st, sums, x.H, x.W, x.C, w2x2, fc0, alpha, fc2, wts);
    QB_CHECK(hipGetLastError());
    return 0;
}

__global__ void scale_channels_kernel(const float* __restrict__ in, const float* __restrict__ wts, float* __restrict__ out,
                                      int B, long HW, int C, int in_cs, int out_cs) {
    const long total = (long)B * HW * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / C;
        const int c = (int)(i - pix * C);
        const long b = pix / HW;
        out[pix * out_cs + c] = wts[b * C + c] * in[pix * in_cs + c];
    }
}

int launch_scale_channels(const View& in, const float* wts, const View& out, int B, hipStream_t st) {
    const long HW = (long)in.H * in.W;
    hipLaunchKernelGGL(scale_channels_kernel, dim3(grid_for((long)B * HW * in.C)), dim3(256), 0, st, in.p, wts, out.p, B, HW,
                       in.C, in.cs, out.cs);
    QB_CHECK(hipGetLastError());
    return 0;
}

// MAD gate (lmffnet.py:268-277): q[b][c][pix] = o[pix][c] * sigmoid(att[pix][c])  (NHWC in, planar out)
__global__ void mad_gate_kernel(const float* __restrict__ o, const float* __restrict__ att, float* __restrict__ q, int B,
                                long HW, int C, int o_cs, int a_cs) {
    const long total = (long)B * HW * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / C;
        const int c = (int)(i - pix * C);
        const long b = pix / HW, p = pix - b * HW;
        const float a = 1.f / (1.f + expf(-att[pix * a_cs + c]));
        q[(b * C + c) * HW + p] = o[pix * o_cs + c] * a;
    }
}

int launch_mad_gate(const View& o, const View& att, float* q, int B, int C, hipStream_t st) {
    const long HW = (long)o.H * o.W;
    hipLaunchKernelGGL(mad_gate_kernel, dim3(grid_for((long)B * HW * C)), dim3(256), 0, st, o.p, att.p, q, B, HW, C, o.cs,
                       att.cs);
    QB_CHECK(hipGetLastError());
    return 0;
}

// predictor.py:85,98: argmax over the class planes (first maximum wins), foreground = class `cls`
__global__ void argmax_fg_kernel(const float* __restrict__ logits, int B, long HW, int nc, int cls, uint8_t* __restrict__ fg) {
    const long total = (long)B * HW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / HW, p = i - b * HW;
        const float* l = logits + b * nc * HW + p;
        int best = 0;
        float bv = l[0];
        for (int c = 1; c < nc; ++c) {
            const float v = l[c * HW];
            if (v > bv) { bv = v; best = c; }
        }
        fg[i] = best == cls ? 1 : 0;
    }
}

int launch_argmax_fg(const float* logits, int B, long HW, int nc, int cls, uint8_t* fg, hipStream_t st) {
    hipLaunchKernelGGL(argmax_fg_kernel, dim3(grid_for((long)B * HW)), dim3(256), 0, st, logits, B, HW, nc, cls, fg);
    QB_CHECK(hipGetLastError());
    return 0;
}

// eval/refiner_model.py:274-277: per refined mask, |mask & fg| and |mask|   -> counts[b][k][2]
__global__ __launch_bounds__(256) void mask_overlap_kernel(const uint8_t* __restrict__ masks, const uint8_t* __restrict__ fg,
                                                           long HW, int K, unsigned long long* __restrict__ counts) {
    const int k = blockIdx.y, b = blockIdx.z;
    const uint8_t* m = masks + ((long)b * K + k) * HW;
    const uint8_t* f = fg + (long)b * HW;
    unsigned inter = 0, area = 0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < HW; i += (long)gridDim.x * blockDim.x) {
        const bool on = m[i] != 0;
        area += on;
        inter += on && f[i] != 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        inter += __shfl_down(inter, o);
        area += __shfl_down(area, o);
    }
    if ((threadIdx.x & 63) == 0 && area) {
        atomicAdd(&counts[((long)b * K + k) * 2], (unsigned long long)inter);
        atomicAdd(&counts[((long)b * K + k) * 2 + 1], (unsigned long long)area);
    }
}

int launch_mask_overlap(const uint8_t* masks, const uint8_t* fg, int B, int K, long HW, unsigned long long* counts,
                        hipStream_t st) {
    if (int rc = launch_zero(counts, sizeof(unsigned long long) * 2 * B * K, st)) return rc;
    int bx = (int)((HW + 256 * 16 - 1) / (256 * 16));
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(mask_overlap_kernel, dim3(bx, K, B), dim3(256), 0, st, masks, fg, HW, K, counts);
    QB_CHECK(hipGetLastError());
    return 0;
}

}  // namespace quber
