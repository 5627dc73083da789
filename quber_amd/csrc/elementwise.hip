// HBM-bound glue kernels of the refiner network (NHWC fp32, 16-byte accesses, grid-stride).
//  * preprocess      - maskrefiner/modeling/mask_refiner/model.py:137-153 + backbone/resnet.py:493-498
//  * maxpool 3x3/2   - backbone/resnet.py:75
//  * GroupNorm(32)   - nn.GroupNorm in resnet.py:473,482 and [d2] get_norm("GN") in model.py:386-403
//  * bilinear resize - F.interpolate(align_corners=False) in [d2] DeepLabV3PlusHead.layers / ASPP
//  * global avg pool - [d2] ASPP image-pooling branch
//  * predictor       - SinglePredictor 1x1 conv (+ channel softmax, model.py:413-422, 752-759)
//  * logits x4       - model.py:689-708
#include "common.h"

namespace quber {

// Activation element type of a tensor (View::es): float, or _Float16 in the fp16 data path (quber_config.compute_dtype 2).
// Every kernel below computes in fp32 whatever the storage type; ldv4 / stv4 move 4 consecutive channels (16 or 8 bytes).
using half_t = _Float16;
template <class T> __device__ inline float4 ldv4(const T* p);
template <> __device__ inline float4 ldv4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ inline float4 ldv4<half_t>(const half_t* p) {
    using h4 = __attribute__((ext_vector_type(4))) half_t;
    const h4 v = *reinterpret_cast<const h4*>(p);
    return make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w);
}
template <class T> __device__ inline void stv4(T* p, float4 v);
template <> __device__ inline void stv4<float>(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
template <> __device__ inline void stv4<half_t>(half_t* p, float4 v) {
    using h4 = __attribute__((ext_vector_type(4))) half_t;
    *reinterpret_cast<h4*>(p) = h4{(half_t)v.x, (half_t)v.y, (half_t)v.z, (half_t)v.w};
}
template <class T> static inline const T* cptr(const View& v) { return reinterpret_cast<const T*>(v.p); }
// 16 bytes per lane whatever the storage type: 4 floats, or 8 halfs (the fp16 data path's pooling / resizing kernels)
template <class T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    static __device__ inline void load(const float* p, float* v) { *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(p); }
    static __device__ inline void store(float* p, const float* v) { *reinterpret_cast<float4*>(p) = *reinterpret_cast<const float4*>(v); }
};
template <> struct Vec16<half_t> {
    static constexpr int N = 8;
    using h8 = __attribute__((ext_vector_type(8))) half_t;
    static __device__ inline void load(const half_t* p, float* v) {
        const h8 x = *reinterpret_cast<const h8*>(p);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)x[e];
    }
    static __device__ inline void store(half_t* p, const float* v) {
        h8 x;
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = (half_t)v[e];
        *reinterpret_cast<h8*>(p) = x;
    }
};


static inline int cap_grid(long work, int per_block) {
    long g = (work + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > 256 * 8) g = 256 * 8;
    return (int)g;
}

// ------------------------------------------------------------------------------------------------
// u8 HWC rgb + u8 HWC depth + f32 planar offsets -> two 8-channel NHWC stream inputs
// x[0] = [(bgr - mean)/std, heat, off_y, off_x, 0, 0],  x[1] = [(depth - mean)/std, heat, off_y, off_x, 0, 0]
template <class T>
__global__ void preprocess_kernel(const uint8_t* __restrict__ rgb, const uint8_t* __restrict__ depth,
                                  const float* __restrict__ offs, T* __restrict__ x, int B, long gstride,
                                  int streams, int HW, int xc, float m0, float m1, float m2, float m3, float m4, float m5, float s0,
                                  float s1, float s2, float s3, float s4, float s5) {
    const long total = (long)B * HW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / HW;
        const long pix = i - b * HW;
        const float* o = offs + b * 3 * HW + pix;
        const float heat = o[0], oy = o[HW], ox = o[2 * (long)HW];
        const uint8_t* r = rgb + i * 3;
        const uint8_t* d = streams == 2 ? depth + i * 3 : r;
        float4 a, c;
        a.x = ((float)r[0] - m0) / s0;
        a.y = ((float)r[1] - m1) / s1;
        a.z = ((float)r[2] - m2) / s2;
        a.w = heat;
        c.x = oy; c.y = ox; c.z = 0.f; c.w = 0.f;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        T* dst = x + i * xc;
        stv4(dst, a);
        stv4(dst + 4, c);
        for (int e = 8; e < xc; e += 4) stv4(dst + e, z);
        if (streams == 2) {
            a.x = ((float)d[0] - m3) / s3;
            a.y = ((float)d[1] - m4) / s4;
            a.z = ((float)d[2] - m5) / s5;
            dst = x + gstride + i * xc;
            stv4(dst, a);
            stv4(dst + 4, c);
            for (int e = 8; e < xc; e += 4) stv4(dst + e, z);
        }
    }
}

int launch_preprocess(const uint8_t* rgb, const uint8_t* depth, const float* offs, const View& x, int B, int Bcap,
                      int H, int W, const float* mean6, const float* std6, int streams, hipStream_t st) {
    const long total = (long)B * H * W;
    ProfScope prof("preprocess", (double)total * (3.0 * streams + 12.0 + (double)x.es * x.C * streams), 0.0, st);
    if (x.es == 2)
        hipLaunchKernelGGL(preprocess_kernel<half_t>, dim3(cap_grid(total, 256)), dim3(256), 0, st, rgb, depth, offs, (half_t*)x.p, B,
                           (long)Bcap * H * W * x.C, streams, H * W, x.C, mean6[0], mean6[1], mean6[2], mean6[3], mean6[4], mean6[5],
                           std6[0], std6[1], std6[2], std6[3], std6[4], std6[5]);
    else
    hipLaunchKernelGGL(preprocess_kernel<float>, dim3(cap_grid(total, 256)), dim3(256), 0, st, rgb, depth, offs, x.p, B,
                       (long)Bcap * H * W * x.C, streams, H * W, x.C, mean6[0], mean6[1], mean6[2], mean6[3], mean6[4], mean6[5],
                       std6[0], std6[1], std6[2], std6[3], std6[4], std6[5]);
    QB_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// grid (ceil(OW * CV / 256), OH, G * B): a block row is one output row - one 32-bit division per thread (the grid-stride form took four
// 64-bit divisions per element); V = 16 bytes of channels per lane (4 floats / 8 halfs), falling back to 4 channels when C % 8 != 0
template <class T, int V>
__global__ __launch_bounds__(256) void maxpool_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int H, int W, int CV,
                                                      int in_cs, int OH, int OW, int out_cs, long in_gs, long out_gs) {
    const int g = blockIdx.z / B, b = blockIdx.z - g * B;
    in += g * in_gs;
    out += g * out_gs;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= OW * CV) return;
    const int ox = idx / CV, cv = idx - ox * CV;
    const int oy = blockIdx.y;
    float m[V];
#pragma unroll
    for (int e = 0; e < V; ++e) m[e] = -INFINITY;
    // A tap outside the map is read from the nearest pixel inside instead of being skipped: that pixel belongs to the same window (row
    // 2 oy - 1 < 0 -> row 0 = 2 oy; row 2 oy + 1 = H -> row H - 1 = 2 oy; columns alike), so the maximum is the same - and the nine
    // loads are unconditional, requested together, where nine `if inside: load` made nine memory latencies in a row.
    float v[9][V];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int iy = min(max(oy * 2 - 1 + dy, 0), H - 1);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int ix = min(max(ox * 2 - 1 + dx, 0), W - 1);
            const T* src = in + ((long)(b * H + iy) * W + ix) * in_cs + cv * V;
            if constexpr (V == Vec16<T>::N) Vec16<T>::load(src, v[dy * 3 + dx]);
            else { const float4 q = ldv4(src); v[dy * 3 + dx][0] = q.x; v[dy * 3 + dx][1] = q.y; v[dy * 3 + dx][2] = q.z; v[dy * 3 + dx][3] = q.w; }
        }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int e = 0; e < V; ++e) m[e] = fmaxf(m[e], v[k][e]);
    T* dst = out + ((long)(b * OH + oy) * OW + ox) * out_cs + cv * V;
    if constexpr (V == Vec16<T>::N) Vec16<T>::store(dst, m);
    else stv4(dst, make_float4(m[0], m[1], m[2], m[3]));
}

int launch_maxpool3x3s2(const View& in, const View& out, int B, int G, hipStream_t st) {
    ProfScope prof("maxpool", (double)in.es * G * B * in.C * ((double)in.H * in.W + (double)out.H * out.W), 0.0, st);
    if (in.es != out.es) return fail("maxpool: mixed element types");
    if (in.es == 2) {
        const bool wide = in.C % 8 == 0 && in.cs % 8 == 0 && out.cs % 8 == 0 && (in.gs & 7) == 0 && (out.gs & 7) == 0 &&
                          (((uintptr_t)in.p | (uintptr_t)out.p) & 15) == 0;      // 16-byte accesses: a channel-slice view may start 8 bytes in
        const int V = wide ? 8 : 4, CV = in.C / V;
        const dim3 grid((out.W * CV + 255) / 256, out.H, G * B);
        if (wide)
            hipLaunchKernelGGL((maxpool_kernel<half_t, 8>), grid, dim3(256), 0, st, cptr<half_t>(in), (half_t*)out.p, B, in.H, in.W, CV, in.cs,
                               out.H, out.W, out.cs, in.gs, out.gs);
        else
            hipLaunchKernelGGL((maxpool_kernel<half_t, 4>), grid, dim3(256), 0, st, cptr<half_t>(in), (half_t*)out.p, B, in.H, in.W, CV, in.cs,
                               out.H, out.W, out.cs, in.gs, out.gs);
    } else {
        const int CV = in.C / 4;
        hipLaunchKernelGGL((maxpool_kernel<float, 4>), dim3((out.W * CV + 255) / 256, out.H, G * B), dim3(256), 0, st, in.p, out.p, B, in.H,
                           in.W, CV, in.cs, out.H, out.W, out.cs, in.gs, out.gs);
    }
    QB_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// GroupNorm statistics: sum and sum of squares per (group-of-launch g, sample b, norm group), in fp64.
// grid = (chunks, B, G); every block streams a contiguous run of pixels with float4 loads.
template <class T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ in, int HW, int C, int cs, long gs,
                                                       long bstride, int groups, int ppb, double* __restrict__ stats,
                                                       int B) {
    __shared__ double acc[64 * 2];
    const int t = threadIdx.x;
    if (t < 128) acc[t] = 0.0;
    __syncthreads();
    const int b = blockIdx.y, g = blockIdx.z;
    const T* base = in + g * gs + b * bstride;
    const int C4 = C >> 2;
    const int cpg = C / groups;
    const int p0 = blockIdx.x * ppb;
    const int p1 = min(HW, p0 + ppb);
    // column passes keep the norm group of a thread fixed inside a pass
    const int colsper = min(C4, 256);
    const int rows = 256 / colsper;
    const int col = t % colsper, row = t / colsper;
    for (int cp = 0; cp < C4; cp += colsper) {
        const int c4 = cp + col;
        if (row >= rows || c4 >= C4) continue;   // idle lanes when 256 is not a multiple of the row width
        // a lane keeps its four channels over all its pixels: four (sum, sum of squares) pairs in registers, folded into the norm
        // groups once at the end - whatever the channels per group (32-channel heads have ONE channel per group: the per-element
        // LDS atomics of the first version ran those passes at 1 TB/s)
        double s[4] = {0.0, 0.0, 0.0, 0.0}, ss[4] = {0.0, 0.0, 0.0, 0.0};
        int pix = p0 + row;
        for (; pix + 3 * rows < p1; pix += 4 * rows) {        // four independent 16-byte loads in flight per lane
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = ldv4(base + (long)(pix + u * rows) * cs + c4 * 4);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s[0] += (double)v[u].x; s[1] += (double)v[u].y; s[2] += (double)v[u].z; s[3] += (double)v[u].w;
                ss[0] += (double)v[u].x * v[u].x; ss[1] += (double)v[u].y * v[u].y; ss[2] += (double)v[u].z * v[u].z; ss[3] += (double)v[u].w * v[u].w;
            }
        }
        for (; pix < p1; pix += rows) {
            const float4 v = ldv4(base + (long)pix * cs + c4 * 4);
            s[0] += (double)v.x; s[1] += (double)v.y; s[2] += (double)v.z; s[3] += (double)v.w;
            ss[0] += (double)v.x * v.x; ss[1] += (double)v.y * v.y; ss[2] += (double)v.z * v.z; ss[3] += (double)v.w * v.w;
        }
        if (cpg % 4 == 0) {          // the lane's four channels lie in one norm group (cpg = 10, 320 channels in 32 groups, does not: found by tests/test_gpu_h8.py's random geometries)
            const int grp = (c4 * 4) / cpg;
            atomicAdd(&acc[grp * 2], (s[0] + s[1]) + (s[2] + s[3]));
            atomicAdd(&acc[grp * 2 + 1], (ss[0] + ss[1]) + (ss[2] + ss[3]));
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int grp = (c4 * 4 + j) / cpg;
                atomicAdd(&acc[grp * 2], s[j]);
                atomicAdd(&acc[grp * 2 + 1], ss[j]);
            }
        }
    }
    __syncthreads();
    if (t < groups * 2) atomicAdd(&stats[((long)(g * B + b) * groups) * 2 + t], acc[t]);
}

static int gn_pixels_per_block(int HW, int C, int B, int G, bool stats = false) {
    // ~64K floats per block at large batch, but never fewer than ~1000 blocks in flight at small batch.
    // stats: one block per ~8K floats at most (128 blocks at least): every block of the statistics pass ends in 2 x groups fp64 atomics on
    // the SAME addresses, and a thousand blocks of 19 pixels each (a 32-channel head tensor of one frame) spent 16-25 us queueing on them
    // (profiles/r13d_b1_kernels_480x640_before.txt)
    int ppb = 65536 / C;
    const long pixels = (long)HW * B * G;
    long blocks = 1024;
    if (stats) blocks = std::min<long>(1024, std::max<long>(128, pixels * C / 8192));
    const long want = (pixels + blocks - 1) / blocks;
    if (ppb > want) ppb = (int)want;
    if (ppb < 8) ppb = 8;
    return ppb;
}

// Zero fill as a plain kernel: cheaper than the runtime's fill path for the small accumulators of this library and,
// unlike hipMemsetAsync nodes, replayed correctly when the step is captured in a hipGraph (ROCm 7.2: the captured
// memsets did not clear the buffers on replay - tests/test_gpu_network.py::test_config2_...).
__global__ void zero_kernel(unsigned* __restrict__ p, size_t words) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}

int launch_zero(void* p, size_t bytes, hipStream_t st) {
    if (((uintptr_t)p & 3) || (bytes & 3)) return fail("zero fill: 4-byte granularity");
    if (!bytes) return 0;
    const size_t words = bytes / 4;
    ProfScope prof("zero", (double)bytes, 0.0, st);
    hipLaunchKernelGGL(zero_kernel, dim3(cap_grid((long)words, 256)), dim3(256), 0, st, (unsigned*)p, words);
    QB_CHECK(hipGetLastError());
    return 0;
}

int launch_gn_stats(const View& in, int B, int G, int groups, double* stats, hipStream_t st, bool zero) {
    if (groups > 64 || in.C % groups || in.C % 4) return fail("groupnorm: unsupported channel/group count");
    const int HW = in.H * in.W;
    if (zero) {
        const int rc = launch_zero(stats, sizeof(double) * 2 * groups * B * G, st);
        if (rc) return rc;
    }
    const int ppb = gn_pixels_per_block(HW, in.C, B, G, true);
    const int chunks = (HW + ppb - 1) / ppb;
    ProfScope prof("gn_stats", (double)in.es * G * B * (double)HW * in.C, 0.0, st);
    if (in.es == 2)
        hipLaunchKernelGGL(gn_stats_kernel<half_t>, dim3(chunks, B, G), dim3(256), 0, st, cptr<half_t>(in), HW, in.C, in.cs, in.gs,
                           (long)HW * in.cs, groups, ppb, stats, B);
    else
        hipLaunchKernelGGL(gn_stats_kernel<float>, dim3(chunks, B, G), dim3(256), 0, st, in.p, HW, in.C, in.cs, in.gs,
                           (long)HW * in.cs, groups, ppb, stats, B);
    QB_CHECK(hipGetLastError());
    return 0;
}

// y = relu(x*scale + bias), scale = rstd*gamma, bias = beta - mean*scale  (torch's GroupNorm CPU form).
// mean / rstd come from the fp64 sums of gn_stats_kernel.  A thread keeps one 16-byte channel column: its four
// (scale, bias) pairs are computed once, the pixel loop is load - fma - store with no index arithmetic beyond an add.
// V channels = 16 bytes per lane (4 floats / 8 halfs; 4 halfs when C % 8 != 0); four pixels of a thread's column in flight per step
template <class T, int V>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int HW,
                                                       int C, int in_cs, int out_cs, long in_gs, long out_gs, int groups,
                                                       const double* __restrict__ stats, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int param_gs, int relu, int ppb,
                                                       double n, float eps) {
    const int t = threadIdx.x;
    const int b = blockIdx.y, g = blockIdx.z;
    in += g * in_gs + (long)b * HW * in_cs;
    out += g * out_gs + (long)b * HW * out_cs;
    gamma += g * param_gs;
    beta += g * param_gs;
    const double* sbase = stats + ((long)(g * B + b) * groups) * 2;
    const int CV = C / V, cpg = C / groups;
    const int p0 = blockIdx.x * ppb;
    const int p1 = min(HW, p0 + ppb);
    const int colsper = min(CV, 256);
    const int rows = 256 / colsper;
    const int col = t % colsper, row = t / colsper;
    auto ld = [&](const T* p, float* v) __attribute__((always_inline)) {
        if constexpr (V == Vec16<T>::N) Vec16<T>::load(p, v);
        else { const float4 q = ldv4(p); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
    };
    auto stt = [&](T* p, const float* v) __attribute__((always_inline)) {
        if constexpr (V == Vec16<T>::N) Vec16<T>::store(p, v);
        else stv4(p, make_float4(v[0], v[1], v[2], v[3]));
    };
    for (int cp = 0; cp < CV; cp += colsper) {
        const int cv = cp + col;
        if (row >= rows || cv >= CV) continue;
        float sc[V], bi[V];
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int ch = cv * V + j;
            const int grp = ch / cpg;
            const double mean = sbase[2 * grp] / n;
            double var = sbase[2 * grp + 1] / n - mean * mean;
            if (var < 0.0) var = 0.0;
            const float rstd = (float)(1.0 / sqrt(var + (double)eps));
            sc[j] = rstd * gamma[ch];
            bi[j] = beta[ch] - (float)mean * sc[j];
        }
        auto norm = [&](float* v) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < V; ++j) {
                v[j] = fmaf(v[j], sc[j], bi[j]);
                if (relu) v[j] = fmaxf(v[j], 0.f);
            }
        };
        int pix = p0 + row;
        for (; pix + 3 * rows < p1; pix += 4 * rows) {
            float v[4][V];
#pragma unroll
            for (int u = 0; u < 4; ++u) ld(in + (long)(pix + u * rows) * in_cs + cv * V, v[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                norm(v[u]);
                stt(out + (long)(pix + u * rows) * out_cs + cv * V, v[u]);
            }
        }
        for (; pix < p1; pix += rows) {
            float v[V];
            ld(in + (long)pix * in_cs + cv * V, v);
            norm(v);
            stt(out + (long)pix * out_cs + cv * V, v);
        }
    }
}

int launch_gn_apply(const View& in, const View& out, int B, int G, int groups, const double* stats,
                    const float* gamma, const float* beta, int param_gs, float eps, int relu, hipStream_t st) {
    const int HW = in.H * in.W;
    const int ppb = gn_pixels_per_block(HW, in.C, B, G);
    ProfScope prof("gn_apply", 2.0 * in.es * G * B * (double)HW * in.C, 0.0, st);
    if (in.es != out.es) return fail("groupnorm: mixed element types");
    const dim3 grid((HW + ppb - 1) / ppb, B, G);
    if (in.es == 2) {
        const bool wide = in.C % 8 == 0 && in.cs % 8 == 0 && out.cs % 8 == 0 && (in.gs & 7) == 0 && (out.gs & 7) == 0 &&
                          (((uintptr_t)in.p | (uintptr_t)out.p) & 15) == 0;
        if (wide)
            hipLaunchKernelGGL((gn_apply_kernel<half_t, 8>), grid, dim3(256), 0, st, cptr<half_t>(in), (half_t*)out.p, B, HW, in.C, in.cs, out.cs,
                               in.gs, out.gs, groups, stats, gamma, beta, param_gs, relu, ppb, (double)HW * (in.C / groups), eps);
        else
            hipLaunchKernelGGL((gn_apply_kernel<half_t, 4>), grid, dim3(256), 0, st, cptr<half_t>(in), (half_t*)out.p, B, HW, in.C, in.cs, out.cs,
                               in.gs, out.gs, groups, stats, gamma, beta, param_gs, relu, ppb, (double)HW * (in.C / groups), eps);
    } else
        hipLaunchKernelGGL((gn_apply_kernel<float, 4>), grid, dim3(256), 0, st, in.p, out.p, B, HW, in.C,
                           in.cs, out.cs, in.gs, out.gs, groups, stats, gamma, beta, param_gs, relu, ppb,
                           (double)HW * (in.C / groups), eps);
    QB_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// torch's area_pixel_compute_source_index (align_corners=False): src = max(scale*(dst+0.5)-0.5, 0)
__device__ inline void bilin_src(int o, float scale, int in_size, int& i0, int& i1, float& l1) {
    float s = scale * ((float)o + 0.5f) - 0.5f;
    if (s < 0.f) s = 0.f;
    i0 = (int)s;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = s - (float)i0;
}

// grid (ceil(OW * CV / 256), OH, B): as maxpool above - one output row per block row, 16 bytes of channels per lane
template <class T, int V>
__global__ __launch_bounds__(256) void bilinear_kernel(const T* __restrict__ in, T* __restrict__ out, int H, int W, int CV,
                                                       int in_cs, int OH, int OW, int out_cs, float sy, float sx) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= OW * CV) return;
    const int ox = idx / CV, cv = idx - ox * CV;
    const int oy = blockIdx.y, b = blockIdx.z;
    int y0, y1, x0, x1;
    float ly, lx;
    bilin_src(oy, sy, H, y0, y1, ly);
    bilin_src(ox, sx, W, x0, x1, lx);
    const float hy = 1.f - ly, hx = 1.f - lx;
    const T* base = in + (long)b * H * W * in_cs + cv * V;
    float v00[V], v01[V], v10[V], v11[V], r[V];
    auto ld = [&](const T* p, float* v) __attribute__((always_inline)) {
        if constexpr (V == Vec16<T>::N) Vec16<T>::load(p, v);
        else { const float4 q = ldv4(p); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
    };
    ld(base + ((long)y0 * W + x0) * in_cs, v00);
    ld(base + ((long)y0 * W + x1) * in_cs, v01);
    ld(base + ((long)y1 * W + x0) * in_cs, v10);
    ld(base + ((long)y1 * W + x1) * in_cs, v11);
#pragma unroll
    for (int e = 0; e < V; ++e) r[e] = hy * (hx * v00[e] + lx * v01[e]) + ly * (hx * v10[e] + lx * v11[e]);
    T* dst = out + ((long)(b * OH + oy) * OW + ox) * out_cs + cv * V;
    if constexpr (V == Vec16<T>::N) Vec16<T>::store(dst, r);
    else stv4(dst, make_float4(r[0], r[1], r[2], r[3]));
}

int launch_bilinear(const View& in, const View& out, int B, hipStream_t st) {
    ProfScope prof("bilinear", (double)in.es * B * in.C * ((double)in.H * in.W + (double)out.H * out.W), 0.0, st);
    if (in.es != out.es) return fail("bilinear: mixed element types");
    const float sy = (float)in.H / (float)out.H, sx = (float)in.W / (float)out.W;
    if (in.es == 2) {
        const bool wide = in.C % 8 == 0 && in.cs % 8 == 0 && out.cs % 8 == 0 && (((uintptr_t)in.p | (uintptr_t)out.p) & 15) == 0;
        const int V = wide ? 8 : 4, CV = in.C / V;
        const dim3 grid((out.W * CV + 255) / 256, out.H, B);
        if (wide)
            hipLaunchKernelGGL((bilinear_kernel<half_t, 8>), grid, dim3(256), 0, st, cptr<half_t>(in), (half_t*)out.p, in.H, in.W, CV, in.cs,
                               out.H, out.W, out.cs, sy, sx);
        else
            hipLaunchKernelGGL((bilinear_kernel<half_t, 4>), grid, dim3(256), 0, st, cptr<half_t>(in), (half_t*)out.p, in.H, in.W, CV, in.cs,
                               out.H, out.W, out.cs, sy, sx);
    } else {
        const int CV = in.C / 4;
        hipLaunchKernelGGL((bilinear_kernel<float, 4>), dim3((out.W * CV + 255) / 256, out.H, B), dim3(256), 0, st, in.p, out.p, in.H, in.W,
                           CV, in.cs, out.H, out.W, out.cs, sy, sx);
    }
    QB_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// global average pool: grid (C/64, B); 256 threads = 16 float4 channel columns x 16 pixel lanes, fp64 sums
template <class T>
__global__ __launch_bounds__(256) void avgpool_kernel(const T* __restrict__ in, T* __restrict__ out, int HW,
                                                      int C, int in_cs, int out_cs) {
    __shared__ double part[16][64];
    const int col = threadIdx.x & 15, lane = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + col * 4;
    const int b = blockIdx.y;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    if (c < C) {    // C % 4 == 0: a float4 column is inside the tensor or entirely outside
        // eight loads in flight per thread (the sums keep their pixel order: same bits as one load at a time, 4x the bandwidth of it)
        int p = lane;
        for (; p + 7 * 16 < HW; p += 8 * 16) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ldv4(in + ((long)b * HW + p + 16 * u) * in_cs + c);
#pragma unroll
            for (int u = 0; u < 8; ++u) { s[0] += (double)v[u].x; s[1] += (double)v[u].y; s[2] += (double)v[u].z; s[3] += (double)v[u].w; }
        }
        for (; p < HW; p += 16) {
            const float4 v = ldv4(in + ((long)b * HW + p) * in_cs + c);
            s[0] += (double)v.x; s[1] += (double)v.y; s[2] += (double)v.z; s[3] += (double)v.w;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) part[lane][col * 4 + e] = s[e];
    __syncthreads();
    if (threadIdx.x < 64 && blockIdx.x * 64 + threadIdx.x < C) {
        double t = 0.0;
#pragma unroll
        for (int l = 0; l < 16; ++l) t += part[l][threadIdx.x];
        out[(long)b * out_cs + blockIdx.x * 64 + threadIdx.x] = (T)(float)(t / (double)HW);
    }
}

int launch_avgpool(const View& in, const View& out, int B, hipStream_t st) {
    if (in.C % 4 || in.cs % 4 || ((uintptr_t)in.p & 15)) return fail("avgpool: channels must come in aligned groups of 4");
    if (in.es != out.es) return fail("avgpool: mixed element types");
    ProfScope prof("avgpool", (double)in.es * B * in.C * ((double)in.H * in.W + 1.0), 0.0, st);
    if (in.es == 2)
        hipLaunchKernelGGL(avgpool_kernel<half_t>, dim3((in.C + 63) / 64, B), dim3(256), 0, st, cptr<half_t>(in), (half_t*)out.p,
                           in.H * in.W, in.C, in.cs, out.cs);
    else
        hipLaunchKernelGGL(avgpool_kernel<float>, dim3((in.C + 63) / 64, B), dim3(256), 0, st, in.p, out.p, in.H * in.W, in.C,
                           in.cs, out.cs);
    QB_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// 1x1 predictors on the head features (32 or 64 channels) -> planar quarter-resolution logits q[B][q_nch][h*w],
// optionally also the softmax / sigmoid of a head's `cout` outputs into an NHWC channel slice (the 'pred' fusion target).
// All heads of a hierarchy level in ONE launch: blockIdx.y = head.
template <class T, int CIN>
__global__ void predictor_kernel(const PredHeads hs, int in_cs, float* __restrict__ q, int q_nch, int sm_cs, int B, int HW) {
    const int hd = blockIdx.y;
    const T* __restrict__ in = reinterpret_cast<const T*>(hs.in[hd]);
    const float* __restrict__ w = hs.w[hd];
    const float* __restrict__ bias = hs.bias[hd];
    T* __restrict__ sm = reinterpret_cast<T*>(hs.sm[hd]);
    const int cout = hs.cout[hd], q_ch0 = hs.q_ch0[hd], act = hs.act[hd];
    __shared__ float ws[4 * CIN + 4];
    for (int i = threadIdx.x; i < cout * CIN; i += blockDim.x) ws[i] = w[i];
    if ((int)threadIdx.x < cout) ws[4 * CIN + threadIdx.x] = bias[threadIdx.x];
    __syncthreads();
    const long total = (long)B * HW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        float x[CIN];
        const T* src = in + i * in_cs;
#pragma unroll
        for (int j = 0; j < CIN / 4; ++j) {
            const float4 v = ldv4(src + 4 * j);
            x[4 * j] = v.x; x[4 * j + 1] = v.y; x[4 * j + 2] = v.z; x[4 * j + 3] = v.w;
        }
        const long b = i / HW, pix = i - b * HW;
        float o[4];
        for (int k = 0; k < cout; ++k) {
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < CIN; ++j) a = fmaf(x[j], ws[k * CIN + j], a);
            a += ws[4 * CIN + k];
            o[k] = a;
            q[((long)b * q_nch + q_ch0 + k) * HW + pix] = a;
        }
        if (sm && act == 2) {
            for (int k = 0; k < cout; ++k) sm[i * sm_cs + k] = (T)(1.f / (1.f + expf(-o[k])));
        } else if (sm) {
            float mx = o[0];
            for (int k = 1; k < cout; ++k) mx = fmaxf(mx, o[k]);
            float e[4], s = 0.f;
            for (int k = 0; k < cout; ++k) { e[k] = expf(o[k] - mx); s += e[k]; }
            for (int k = 0; k < cout; ++k) sm[i * sm_cs + k] = (T)(e[k] / s);
        }
    }
}

int launch_predictors(const PredHeads& hs, int C, int in_cs, int es, int H, int W, float* q, int q_nch, int sm_cs, int B, hipStream_t st) {
    // hs.sm[j]: a channel slice of an activation buffer - same element type as the head features
    if ((C != 32 && C != 64) || hs.n < 1 || hs.n > 5) return fail("predictor: expects 32 or 64 input channels and 1..5 heads");
    double couts = 0.0, acts = 0.0;
    for (int j = 0; j < hs.n; ++j) {
        if (hs.cout[j] > 4) return fail("predictor: at most 4 outputs per head");
        couts += hs.cout[j];
        if (hs.sm[j]) acts += hs.cout[j];
    }
    const long total = (long)B * H * W;
    ProfScope prof("predictor", total * ((double)es * C * hs.n + 4.0 * couts + (double)es * acts), 2.0 * total * C * couts, st);
    const dim3 grid(cap_grid(total, 256), hs.n), block(256);
    if (es == 2 && C == 32) hipLaunchKernelGGL((predictor_kernel<half_t, 32>), grid, block, 0, st, hs, in_cs, q, q_nch, sm_cs, B, H * W);
    else if (es == 2) hipLaunchKernelGGL((predictor_kernel<half_t, 64>), grid, block, 0, st, hs, in_cs, q, q_nch, sm_cs, B, H * W);
    else if (C == 32) hipLaunchKernelGGL((predictor_kernel<float, 32>), grid, block, 0, st, hs, in_cs, q, q_nch, sm_cs, B, H * W);
    else hipLaunchKernelGGL((predictor_kernel<float, 64>), grid, block, 0, st, hs, in_cs, q, q_nch, sm_cs, B, H * W);
    QB_CHECK(hipGetLastError());
    return 0;
}

template <class T>
__global__ void copy_channels_kernel(const T* __restrict__ in, T* __restrict__ out, long pixels, int C4,
                                     int in_cs, int out_cs) {
    const long total = pixels * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / C4;
        const int c4 = (int)(i - pix * C4);
        stv4(out + pix * out_cs + c4 * 4, ldv4(in + pix * in_cs + c4 * 4));
    }
}

template <class T>
__global__ void add_channels_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out,
                                    long pixels, int C4, int a_cs, int b_cs, int out_cs) {
    const long total = pixels * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pix = i / C4;
        const int c4 = (int)(i - pix * C4);
        const float4 x = ldv4(a + pix * a_cs + c4 * 4);
        const float4 y = ldv4(b + pix * b_cs + c4 * 4);
        stv4(out + pix * out_cs + c4 * 4, make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w));
    }
}

int launch_add_channels(const View& a, const View& b, const View& out, int B, hipStream_t st) {
    const long pixels = (long)B * a.H * a.W;
    ProfScope prof("add_channels", 3.0 * a.es * pixels * a.C, 0.0, st);
    if (a.es != b.es || a.es != out.es) return fail("add: mixed element types");
    if (a.es == 2)
        hipLaunchKernelGGL(add_channels_kernel<half_t>, dim3(cap_grid(pixels * (a.C / 4), 256)), dim3(256), 0, st, cptr<half_t>(a),
                           cptr<half_t>(b), (half_t*)out.p, pixels, a.C / 4, a.cs, b.cs, out.cs);
    else
        hipLaunchKernelGGL(add_channels_kernel<float>, dim3(cap_grid(pixels * (a.C / 4), 256)), dim3(256), 0, st, a.p, b.p, out.p,
                           pixels, a.C / 4, a.cs, b.cs, out.cs);
    QB_CHECK(hipGetLastError());
    return 0;
}

int launch_copy_channels(const View& in, const View& out, int B, hipStream_t st) {
    const long pixels = (long)B * in.H * in.W;
    ProfScope prof("copy_channels", 2.0 * in.es * pixels * in.C, 0.0, st);
    if (in.es != out.es) return fail("copy: mixed element types");
    if (in.es == 2)
        hipLaunchKernelGGL(copy_channels_kernel<half_t>, dim3(cap_grid(pixels * (in.C / 4), 256)), dim3(256), 0, st, cptr<half_t>(in),
                           (half_t*)out.p, pixels, in.C / 4, in.cs, out.cs);
    else
        hipLaunchKernelGGL(copy_channels_kernel<float>, dim3(cap_grid(pixels * (in.C / 4), 256)), dim3(256), 0, st, in.p, out.p,
                           pixels, in.C / 4, in.cs, out.cs);
    QB_CHECK(hipGetLastError());
    return 0;
}

// planar bilinear x`scale` of the predictor logits, cropped to the frame (sem_seg_postprocess, model.py:266-289:
// when H or W is not a multiple of 16 the x4 map is larger than the image and only its top-left H x W part is
// kept); channels whose bit is set in mul_mask are multiplied by `scale` afterwards (offsets, model.py:695-700)
// One thread = V consecutive output pixels of one row of one plane (V = 4 with 16-byte stores when the row length
// allows); all index arithmetic is 32-bit and per thread, not per element.
template <int V>
__global__ __launch_bounds__(256) void upsample_logits_kernel(const float* __restrict__ q, float* __restrict__ out, int h,
                                                              int w, int nch, int scale, int OH, int OW,
                                                              unsigned mul_mask) {
    const int pl = blockIdx.y;                       // plane = frame * nch + channel
    const unsigned per_row = (unsigned)OW / V;
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= per_row * (unsigned)OH) return;
    const int oy = idx / per_row;
    const int ox0 = (idx - oy * per_row) * V;
    const float inv = 1.f / (float)scale;
    int y0, y1;
    float ly;
    bilin_src(oy, inv, h, y0, y1, ly);
    const float hy = 1.f - ly;
    const float* s0 = q + (long)pl * h * w + (long)y0 * w;
    const float* s1 = q + (long)pl * h * w + (long)y1 * w;
    const float mul = ((mul_mask >> (pl % nch)) & 1u) ? (float)scale : 1.f;
    float o[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        int x0, x1;
        float lx;
        bilin_src(ox0 + e, inv, w, x0, x1, lx);
        const float hx = 1.f - lx;
        float v = hy * (hx * s0[x0] + lx * s0[x1]) + ly * (hx * s1[x0] + lx * s1[x1]);
        if (mul != 1.f) v *= mul;
        o[e] = v;
    }
    float* dst = out + ((long)pl * OH + oy) * OW + ox0;
    if constexpr (V == 4) {
        *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
        dst[0] = o[0];
    }
}

int launch_upsample_logits(const float* q, float* out, int B, int nch, int h, int w, int scale, int OH, int OW,
                           unsigned mul_mask, hipStream_t st) {
    if (OH > h * scale || OW > w * scale) return fail("upsample: frame larger than the scaled head map");
    if ((long)OH * OW >= (1L << 31)) return fail("upsample: frame too large");
    const int planes = B * nch;
    ProfScope prof("upsample_logits", 4.0 * planes * ((double)h * w + (double)OH * OW), 0.0, st);
    if (OW % 4 == 0 && ((uintptr_t)out & 15) == 0) {
        const unsigned n = (unsigned)(OW / 4) * OH;
        hipLaunchKernelGGL(upsample_logits_kernel<4>, dim3((n + 255) / 256, planes), dim3(256), 0, st, q, out, h, w, nch,
                           scale, OH, OW, mul_mask);
    } else {
        const unsigned n = (unsigned)OW * OH;
        hipLaunchKernelGGL(upsample_logits_kernel<1>, dim3((n + 255) / 256, planes), dim3(256), 0, st, q, out, h, w, nch,
                           scale, OH, OW, mul_mask);
    }
    QB_CHECK(hipGetLastError());
    return 0;
}

}  // namespace quber
