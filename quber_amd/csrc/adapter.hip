// Adapter-side pre-processing on the device: depth normalisation and the cv2.resize calls.
// Replaces eval/preprocess_utils.py:12-28 `normalize_depth` (clamp to [min,max], scale to 0..255, truncate to
// uint8, replicate to 3 channels) and records which pixels had zero depth (eval/refiner_model.py:250, used for the
// OCID zero-depth masking at :279-288).  Arithmetic follows numpy: integer depth (uint16 PNG, millimetres) is
// promoted to float64, float32 depth (npy, metres) stays float32.
#include "common.h"

namespace quber {

template <typename T, typename F>
__global__ void normalize_depth_kernel(const T* __restrict__ depth, long n, F lo, F hi, uint8_t* __restrict__ out3,
                                       uint8_t* __restrict__ zero) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const T raw = depth[i];
        F d = (F)raw;
        d = d < lo ? lo : d;
        d = d > hi ? hi : d;
        const F t = (d - lo) / (hi - lo) * (F)255;
        const uint8_t v = (uint8_t)t;               // np.uint8(): truncation
        out3[3 * i] = v;
        out3[3 * i + 1] = v;
        out3[3 * i + 2] = v;
        if (zero) zero[i] = raw == (T)0 ? 1 : 0;
    }
}

int launch_normalize_depth(const void* depth, int is_float, long n, double lo, double hi, uint8_t* out3, uint8_t* zero,
                           hipStream_t st) {
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    if (is_float)
        hipLaunchKernelGGL((normalize_depth_kernel<float, float>), dim3(blocks), dim3(256), 0, st, (const float*)depth, n,
                           (float)lo, (float)hi, out3, zero);
    else
        hipLaunchKernelGGL((normalize_depth_kernel<uint16_t, double>), dim3(blocks), dim3(256), 0, st,
                           (const uint16_t*)depth, n, lo, hi, out3, zero);
    QB_CHECK(hipGetLastError());
    return 0;
}

// ---- cv2.resize on uint8 images (eval/refiner_model.py:229, 232, 246, 254) -------------------------------------------
// OpenCV is not in the image, so these restate its published 8-bit algorithms (parity unpinned; oracle/adapter_np.py
// holds the same restatement, tests derive small cases by hand):
//  INTER_NEAREST   sx = min(floor(dx * (1 / (dw / sw))), sw - 1)                         (double arithmetic)
//  INTER_LINEAR    fx = float((dx + 0.5) * scale - 0.5); sx = floor(fx); fx -= sx; clamped at both ends with fx = 0;
//                  11-bit fixed-point weights a = round_half_even(w * 2048) (float product), horizontal pass in int32,
//                  vertical pass  dst = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2
//                  and, when both scale factors are exactly 2, cv::resize switches to the area filter
//                  dst = (p00 + p01 + p10 + p11 + 2) >> 2.
struct LinCoef { int s0, s1, a0, a1; };

__device__ inline LinCoef lin_coef(int d, double scale, int ssize) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
    LinCoef c;
    c.s0 = s;
    c.s1 = s + 1 < ssize ? s + 1 : ssize - 1;
    c.a0 = __float2int_rn((1.f - f) * 2048.f);
    c.a1 = __float2int_rn(f * 2048.f);
    return c;
}

__global__ void resize_u8_kernel(const uint8_t* __restrict__ src, int sh, int sw, int ch, uint8_t* __restrict__ dst, int dh,
                                 int dw, int mode, double scale_y, double scale_x) {
    const long total = (long)dh * dw * ch;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % ch);
        const int dx = (int)((i / ch) % dw);
        const int dy = (int)(i / ((long)ch * dw));
        if (mode == 0) {
            int sx = (int)floor((double)dx * scale_x), sy = (int)floor((double)dy * scale_y);
            sx = sx < sw - 1 ? sx : sw - 1;
            sy = sy < sh - 1 ? sy : sh - 1;
            dst[i] = src[((long)sy * sw + sx) * ch + c];
        } else if (mode == 2) {
            const uint8_t* p = src + ((long)(2 * dy) * sw + 2 * dx) * ch + c;
            dst[i] = (uint8_t)((p[0] + p[ch] + p[(long)sw * ch] + p[(long)sw * ch + ch] + 2) >> 2);
        } else {
            const LinCoef x = lin_coef(dx, scale_x, sw), y = lin_coef(dy, scale_y, sh);
            const uint8_t* r0 = src + (long)y.s0 * sw * ch + c;
            const uint8_t* r1 = src + (long)y.s1 * sw * ch + c;
            const int h0 = r0[(long)x.s0 * ch] * x.a0 + r0[(long)x.s1 * ch] * x.a1;
            const int h1 = r1[(long)x.s0 * ch] * x.a0 + r1[(long)x.s1 * ch] * x.a1;
            dst[i] = (uint8_t)((((y.a0 * (h0 >> 4)) >> 16) + ((y.a1 * (h1 >> 4)) >> 16) + 2) >> 2);
        }
    }
}

int launch_resize_u8(const uint8_t* src, int sh, int sw, int ch, uint8_t* dst, int dh, int dw, int linear, hipStream_t st) {
    if (sh < 1 || sw < 1 || dh < 1 || dw < 1 || ch < 1 || ch > 4) return fail("resize: bad geometry");
    // cv::resize: inv_scale = dsize / ssize (double), scale = 1 / inv_scale
    const double scale_x = 1.0 / ((double)dw / (double)sw), scale_y = 1.0 / ((double)dh / (double)sh);
    int mode = linear ? 1 : 0;
    if (linear && sw == 2 * dw && sh == 2 * dh) mode = 2;        // INTER_LINEAR at exactly 1/2 scale = the area filter
    const long total = (long)dh * dw * ch;
    ProfScope prof("resize_u8", (double)sh * sw * ch + (double)total, 0.0, st);
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(resize_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, st, src, sh, sw, ch, dst, dh, dw, mode, scale_y, scale_x);
    QB_CHECK(hipGetLastError());
    return 0;
}

}  // namespace quber
