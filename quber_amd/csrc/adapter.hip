// Adapter-side pre-processing on the device: depth normalisation.
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

}  // namespace quber
