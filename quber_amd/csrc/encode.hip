// Initial-mask encoder: N instance masks -> (centre heat-map, y-offset, x-offset) planes.
// Replaces the per-mask numpy loop of maskrefiner/predictor.py:304-357 (same arithmetic as
// explicit_error_estimation/util.py:142-228):
//   centroid  = float64 mean of the mask's pixel coordinates,
//   heat-map  = max over masks of a 63x63 Gaussian (sigma 10) pasted at the banker's-rounded centroid,
//   offsets   = f32((centroid - coord) / extent) for the LAST mask covering the pixel, 0 elsewhere.
//
// Pass 1 streams every mask byte exactly once (16 B per lane, N rows per pixel strip), reducing
// count / sum(y) / sum(x) per mask with wave shuffles and recording the last covering mask per pixel.
// Pass 2 is a per-pixel gather against the <=254 centroids held in LDS.
#include "common.h"

namespace quber {

struct MaskStat {
    unsigned long long cnt, sy, sx;
};

constexpr int ENC_PIX = 16;  // pixels per lane per mask row

__global__ __launch_bounds__(256) void encode_reduce_kernel(const uint8_t* __restrict__ masks, int N, long frame_stride,
                                                            int H, int W, MaskStat* __restrict__ stats,
                                                            uint8_t* __restrict__ last) {
    extern __shared__ unsigned int sm[];  // [N][3]
    const int b = blockIdx.y;
    const long HW = (long)H * W;
    for (int i = threadIdx.x; i < N * 3; i += blockDim.x) sm[i] = 0;
    __syncthreads();
    const long p = ((long)blockIdx.x * blockDim.x + threadIdx.x) * ENC_PIX;
    const bool active = p < HW;
    const int y = active ? (int)(p / W) : 0;
    const int x0 = active ? (int)(p - (long)y * W) : 0;
    uint8_t lastv[ENC_PIX];
#pragma unroll
    for (int j = 0; j < ENC_PIX; ++j) lastv[j] = 0;
    const uint8_t* src = masks + (long)b * frame_stride + p;
    for (int n = 0; n < N; ++n) {
        unsigned cnt = 0, sx = 0;
        if (active) {
            const uint4 v = *reinterpret_cast<const uint4*>(src + (long)n * HW);
            const unsigned wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < ENC_PIX; ++j) {
                const bool on = ((wds[j >> 2] >> (8 * (j & 3))) & 0xffu) != 0;
                if (on) {
                    cnt += 1;
                    sx += x0 + j;
                    lastv[j] = (uint8_t)(n + 1);
                }
            }
        }
        unsigned sy = cnt * (unsigned)y;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            cnt += __shfl_down(cnt, o);
            sy += __shfl_down(sy, o);
            sx += __shfl_down(sx, o);
        }
        if ((threadIdx.x & 63) == 0 && cnt) {
            atomicAdd(&sm[n * 3], cnt);
            atomicAdd(&sm[n * 3 + 1], sy);
            atomicAdd(&sm[n * 3 + 2], sx);
        }
    }
    if (active) {
        uint4 o;
        unsigned* ow = &o.x;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            ow[k] = lastv[4 * k] | (lastv[4 * k + 1] << 8) | (lastv[4 * k + 2] << 16) | ((unsigned)lastv[4 * k + 3] << 24);
        *reinterpret_cast<uint4*>(last + (long)b * HW + p) = o;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < N * 3; i += blockDim.x)
        if (sm[i]) atomicAdd(&(&stats[(long)b * N].cnt)[i], (unsigned long long)sm[i]);
}

// Fallback for frame widths that are not a multiple of 16 (rows are then not 16-byte aligned): one pixel per lane.
__global__ __launch_bounds__(256) void encode_reduce_generic_kernel(const uint8_t* __restrict__ masks, int N, long frame_stride,
                                                                    int H, int W, MaskStat* __restrict__ stats,
                                                                    uint8_t* __restrict__ last) {
    extern __shared__ unsigned int sm[];  // [N][3]
    const int b = blockIdx.y;
    const long HW = (long)H * W;
    for (int i = threadIdx.x; i < N * 3; i += blockDim.x) sm[i] = 0;
    __syncthreads();
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = p < HW;
    const int y = active ? (int)(p / W) : 0;
    const int x = active ? (int)(p - (long)y * W) : 0;
    unsigned lastv = 0;
    const uint8_t* src = masks + (long)b * frame_stride + p;
    for (int n = 0; n < N; ++n) {
        const bool on = active && src[(long)n * HW] != 0;
        if (on) lastv = n + 1;
        unsigned cnt = on ? 1u : 0u, sy = on ? (unsigned)y : 0u, sx = on ? (unsigned)x : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            cnt += __shfl_down(cnt, o);
            sy += __shfl_down(sy, o);
            sx += __shfl_down(sx, o);
        }
        if ((threadIdx.x & 63) == 0 && cnt) {
            atomicAdd(&sm[n * 3], cnt);
            atomicAdd(&sm[n * 3 + 1], sy);
            atomicAdd(&sm[n * 3 + 2], sx);
        }
    }
    if (active) last[(long)b * HW + p] = (uint8_t)lastv;
    __syncthreads();
    for (int i = threadIdx.x; i < N * 3; i += blockDim.x)
        if (sm[i]) atomicAdd(&(&stats[(long)b * N].cnt)[i], (unsigned long long)sm[i]);
}

__global__ __launch_bounds__(256) void encode_paint_kernel(const MaskStat* __restrict__ stats,
                                                           const uint8_t* __restrict__ last,
                                                           const float* __restrict__ gauss, int N, int H, int W,
                                                           int radius, int legacy_f32, int accumulate,
                                                           float* __restrict__ out) {
    extern __shared__ double smd[];  // [N] cy, [N] cx, then int [N] iy, [N] ix
    double* cy = smd;
    double* cx = smd + N;
    int* iy = reinterpret_cast<int*>(smd + 2 * N);
    int* ix = iy + N;
    const int b = blockIdx.y;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const MaskStat s = stats[(long)b * N + n];
        if (s.cnt) {
            const double my = (double)s.sy / (double)s.cnt;
            const double mx = (double)s.sx / (double)s.cnt;
            cy[n] = my;
            cx[n] = mx;
            iy[n] = (int)rint(my);  // round-half-even, like Python's round()
            ix[n] = (int)rint(mx);
        } else {
            cy[n] = 0.0;
            cx[n] = 0.0;
            iy[n] = -(1 << 20);     // empty masks are skipped (predictor.py:315-317)
            ix[n] = -(1 << 20);
        }
    }
    __syncthreads();
    const long HW = (long)H * W;
    const int side = 2 * radius + 1;
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const int y = (int)(p / W), x = (int)(p - (long)y * W);
    float heat = 0.f;
    for (int n = 0; n < N; ++n) {
        const int dy = y - iy[n] + radius, dx = x - ix[n] + radius;
        if ((unsigned)dy < (unsigned)side && (unsigned)dx < (unsigned)side) heat = fmaxf(heat, gauss[dy * side + dx]);
    }
    float oy = 0.f, ox = 0.f;
    const int l = last[(long)b * HW + p];
    if (l) {
        if (legacy_f32) {
            // numpy < 2 value-based casting (the reference pins numpy==1.23.1, INSTALL.md:14): the float64 centroid scalar
            // does not upcast the float32 coordinate array, so predictor.py:345-346 evaluates entirely in float32
            oy = ((float)cy[l - 1] - (float)y) / (float)H;
            ox = ((float)cx[l - 1] - (float)x) / (float)W;
        } else {
            // numpy >= 2 (NEP 50): float64 scalar - float32 array -> float64, rounded once on the store
            oy = (float)((cy[l - 1] - (double)y) / (double)H);
            ox = (float)((cx[l - 1] - (double)x) / (double)W);
        }
    }
    float* o = out + (long)b * 3 * HW + p;
    if (accumulate) {          // a further chunk of > 254 masks: max-paste on the earlier heat-map, overwrite only covered pixels
        o[0] = fmaxf(o[0], heat);
        if (l) { o[HW] = oy; o[2 * HW] = ox; }
        return;
    }
    o[0] = heat;
    o[HW] = oy;
    o[2 * HW] = ox;
}

constexpr int ENC_CHUNK = 254;     // masks per pass: the last-covering-mask map is one byte per pixel (0 = none)

size_t encode_ws_bytes(int B, int N, int H, int W) {
    if (N > ENC_CHUNK) N = ENC_CHUNK;
    return (size_t)B * N * sizeof(MaskStat) + (size_t)B * H * W + 16;
}

// Any number of masks, like the reference's Python loop (predictor.py:310): more than 254 are encoded in chunks, each
// later chunk max-pasting its Gaussians and overwriting the offsets of the pixels it covers ("later masks win").
int launch_encode(const uint8_t* masks, int B, int N, int H, int W, const float* gauss, int sigma, int legacy_f32, void* ws,
                  float* out, hipStream_t st) {
    if (N < 0) return fail("encode: negative mask count");
    const long HW = (long)H * W;
    // workspace: [last-index map B*H*W bytes][MaskStat B*min(N, 254)]  (the map first keeps its 16-byte alignment)
    uint8_t* last = reinterpret_cast<uint8_t*>(ws);
    MaskStat* stats = reinterpret_cast<MaskStat*>(last + (((size_t)B * HW + 15) & ~(size_t)15));
    if (N == 0) {
        return launch_zero(out, sizeof(float) * 3 * HW * B, st);
    }
    for (int c0 = 0; c0 < N; c0 += ENC_CHUNK) {
        const int n = N - c0 < ENC_CHUNK ? N - c0 : ENC_CHUNK;
        const uint8_t* m = masks + (long)c0 * HW;
        if (int rc = launch_zero(stats, sizeof(MaskStat) * (size_t)B * n, st)) return rc;
        {
            ProfScope prof("encode_reduce", (double)B * HW * (n + 1.0), 0.0, st);      // every mask byte once + the 1 B/px index map
            if (W % ENC_PIX == 0 && ((uintptr_t)m & 15) == 0 && (HW & 15) == 0) {
                const int blocks = (int)((HW / ENC_PIX + 255) / 256);
                hipLaunchKernelGGL(encode_reduce_kernel, dim3(blocks, B), dim3(256), sizeof(unsigned) * 3 * n, st, m, n,
                                   (long)N * HW, H, W, stats, last);
            } else {
                hipLaunchKernelGGL(encode_reduce_generic_kernel, dim3((int)((HW + 255) / 256), B), dim3(256),
                                   sizeof(unsigned) * 3 * n, st, m, n, (long)N * HW, H, W, stats, last);
            }
        }
        QB_CHECK(hipGetLastError());
        ProfScope prof("encode_paint", (double)B * HW * 13.0, 0.0, st);                // index map in, three f32 planes out
        const size_t sm2 = (size_t)n * (2 * sizeof(double) + 2 * sizeof(int));
        hipLaunchKernelGGL(encode_paint_kernel, dim3((int)((HW + 255) / 256), B), dim3(256), sm2, st, stats, last, gauss, n,
                           H, W, 3 * sigma + 1, legacy_f32, c0 > 0 ? 1 : 0, out);
        QB_CHECK(hipGetLastError());
    }
    return 0;
}

}  // namespace quber
