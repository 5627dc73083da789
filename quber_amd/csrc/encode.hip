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

constexpr int ENC_R = 4;     // 16-pixel groups per lane per mask: one wave reduction per 4 KiB of mask bytes

// grid (ceil(HW / (256 * 16 * ENC_R)), B).  A lane owns ENC_R groups of 16 pixels, 1 KiB (one wave load) apart, so every
// load instruction of a wave is one contiguous KiB; per mask it issues its ENC_R loads back to back, folds them into
// (count, sum y, sum x) in registers and the wave reduces once.  The last-covering-mask bytes stay in registers throughout.
__global__ __launch_bounds__(256) void encode_reduce_kernel(const uint8_t* __restrict__ masks, int N, long frame_stride,
                                                            int H, int W, MaskStat* __restrict__ stats,
                                                            uint8_t* __restrict__ last) {
    extern __shared__ unsigned int sm[];  // [N][3]
    const int b = blockIdx.y;
    const long HW = (long)H * W;
    for (int i = threadIdx.x; i < N * 3; i += blockDim.x) sm[i] = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // wave w of block x covers pixels [((x * 4 + w) * ENC_R + r) * 1024 + lane * 16, +16), r = 0..ENC_R-1
    const long base = ((long)blockIdx.x * 4 + wave) * ENC_R * 1024 + lane * ENC_PIX;
    int y[ENC_R], x0[ENC_R];
    bool act[ENC_R];
    uint4 lastv[ENC_R];
#pragma unroll
    for (int r = 0; r < ENC_R; ++r) {
        const long p = base + (long)r * 1024;
        act[r] = p < HW;
        y[r] = act[r] ? (int)(p / W) : 0;
        x0[r] = act[r] ? (int)(p - (long)y[r] * W) : 0;
        lastv[r] = make_uint4(0, 0, 0, 0);
    }
    const uint8_t* src = masks + (long)b * frame_stride + base;
    // the next mask's ENC_R loads are issued before this mask's reduction: 8 KiB per wave in flight
    uint4 nxt[ENC_R];
#pragma unroll
    for (int r = 0; r < ENC_R; ++r)
        nxt[r] = act[r] ? *reinterpret_cast<const uint4*>(src + (long)r * 1024) : make_uint4(0, 0, 0, 0);
    for (int n = 0; n < N; ++n) {
        uint4 v[ENC_R];
#pragma unroll
        for (int r = 0; r < ENC_R; ++r) v[r] = nxt[r];
        if (n + 1 < N) {
#pragma unroll
            for (int r = 0; r < ENC_R; ++r)
                nxt[r] = act[r] ? *reinterpret_cast<const uint4*>(src + (long)(n + 1) * HW + (long)r * 1024) : make_uint4(0, 0, 0, 0);
        }
        unsigned cnt = 0, sx = 0, sy = 0;
        const unsigned tag = (unsigned)(n + 1);
#pragma unroll
        for (int r = 0; r < ENC_R; ++r) {
            const unsigned wds[4] = {v[r].x, v[r].y, v[r].z, v[r].w};
            unsigned* lw = &lastv[r].x;
            unsigned c = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned wd = wds[j];
                // per byte: 0xff where the mask byte is non-zero
                const unsigned nz = ((((wd & 0x7f7f7f7fu) + 0x7f7f7f7fu) | wd) & 0x80808080u) >> 7;     // 0x01 per non-zero byte
                const unsigned full = nz * 0xffu;
                lw[j] = (lw[j] & ~full) | (full & (tag * 0x01010101u));
                const unsigned k = __popc(nz);
                c += k;
                // sum of the byte positions (0..3) of the set bytes: bits 0, 8, 16, 24 of nz
                sx += (unsigned)(x0[r] + 4 * j) * k + ((nz >> 8) & 1u) + 2u * ((nz >> 16) & 1u) + 3u * ((nz >> 24) & 1u);
            }
            cnt += c;
            sy += c * (unsigned)y[r];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            cnt += __shfl_down(cnt, o);
            sy += __shfl_down(sy, o);
            sx += __shfl_down(sx, o);
        }
        if (lane == 0 && cnt) {
            atomicAdd(&sm[n * 3], cnt);
            atomicAdd(&sm[n * 3 + 1], sy);
            atomicAdd(&sm[n * 3 + 2], sx);
        }
    }
#pragma unroll
    for (int r = 0; r < ENC_R; ++r)
        if (act[r]) *reinterpret_cast<uint4*>(last + (long)b * HW + base + (long)r * 1024) = lastv[r];
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

// Label-map input: labels[p] in 1..N names the one mask covering pixel p (0 = none), i.e. N non-overlapping
// masks m_i = (labels == i + 1).  A lane walks 16 consecutive pixels, folds runs of equal labels in registers and adds each
// run's (count, sum y, sum x) to the block's LDS table; the index map is the label itself.
__global__ __launch_bounds__(256) void encode_label_reduce_kernel(const int* __restrict__ labels, int N, int H, int W,
                                                                  MaskStat* __restrict__ stats, uint8_t* __restrict__ last,
                                                                  int* __restrict__ bad) {
    extern __shared__ unsigned int sm[];  // [N][3]
    const int b = blockIdx.y;
    const long HW = (long)H * W;
    for (int i = threadIdx.x; i < N * 3; i += blockDim.x) sm[i] = 0;
    __syncthreads();
    const long p0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    int run = 0;
    unsigned cnt = 0, sy = 0, sx = 0;
    auto flush = [&]() {
        if (run > 0 && cnt) {
            atomicAdd(&sm[(run - 1) * 3], cnt);
            atomicAdd(&sm[(run - 1) * 3 + 1], sy);
            atomicAdd(&sm[(run - 1) * 3 + 2], sx);
        }
        cnt = sy = sx = 0;
    };
    for (int j = 0; j < 16; ++j) {
        const long p = p0 + j;
        if (p >= HW) break;
        int l = labels[(long)b * HW + p];
        if (l < 0 || l > N) { *bad = 1; l = 0; }
        last[(long)b * HW + p] = (uint8_t)l;
        if (l != run) { flush(); run = l; }
        if (l > 0) {
            const int y = (int)(p / W);
            cnt += 1;
            sy += (unsigned)y;
            sx += (unsigned)(p - (long)y * W);
        }
    }
    flush();
    __syncthreads();
    for (int i = threadIdx.x; i < N * 3; i += blockDim.x)
        if (sm[i]) atomicAdd(&(&stats[(long)b * N].cnt)[i], (unsigned long long)sm[i]);
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
                const int blocks = (int)((HW + 256L * ENC_PIX * ENC_R - 1) / (256L * ENC_PIX * ENC_R));
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

int launch_encode_labels(const int* labels, int B, int N, int H, int W, const float* gauss, int sigma, int legacy_f32, void* ws,
                         float* out, int* bad, hipStream_t st) {
    if (N < 0 || N > ENC_CHUNK) return fail("encode (label map): 0..254 instances per frame");
    const long HW = (long)H * W;
    uint8_t* last = reinterpret_cast<uint8_t*>(ws);
    MaskStat* stats = reinterpret_cast<MaskStat*>(last + (((size_t)B * HW + 15) & ~(size_t)15));
    if (N == 0) return launch_zero(out, sizeof(float) * 3 * HW * B, st);
    if (int rc = launch_zero(stats, sizeof(MaskStat) * (size_t)B * N, st)) return rc;
    if (int rc = launch_zero(bad, sizeof(int), st)) return rc;
    {
        ProfScope prof("encode_label_reduce", (double)B * HW * 5.0, 0.0, st);
        hipLaunchKernelGGL(encode_label_reduce_kernel, dim3((int)((HW + 4095) / 4096), B), dim3(256), sizeof(unsigned) * 3 * N, st,
                           labels, N, H, W, stats, last, bad);
    }
    QB_CHECK(hipGetLastError());
    ProfScope prof("encode_paint", (double)B * HW * 13.0, 0.0, st);
    const size_t sm2 = (size_t)N * (2 * sizeof(double) + 2 * sizeof(int));
    hipLaunchKernelGGL(encode_paint_kernel, dim3((int)((HW + 255) / 256), B), dim3(256), sm2, st, stats, last, gauss, N, H, W,
                       3 * sigma + 1, legacy_f32, 0, out);
    QB_CHECK(hipGetLastError());
    return 0;
}

}  // namespace quber
