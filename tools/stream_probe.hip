// Micro-benchmark (GPU box only, not part of the library): what does the MI355X deliver on the initial-mask stream of
// the encoder (B x N planes of H*W bytes, read once)?  Variants isolate the access pattern from the per-byte work:
//   copy     : float4 read of the whole buffer, one block-sum written                      (ceiling at this size)
//   planes   : the encoder's pattern (lane = 16 B of plane n, loop over the N planes), loads only
//   planes4  : 4 x 1 KiB per wave per plane (ENC_R = 4), loads only
//   decode4  : planes4 + the SWAR byte tests / counts of encode_reduce_kernel, no cross-lane reduction
//   reduce4  : decode4 + the three wave reductions per plane
// build: hipcc -O3 --offload-arch=gfx950 tools/stream_probe.hip -o gpurun_out/stream_probe ; run: gpurun_out/stream_probe [B] [N]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void copy_k(const uint4* __restrict__ p, long n, unsigned* out) {
    unsigned acc = 0;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) { const uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int R, int MODE>
__global__ __launch_bounds__(256) void planes_k(const uint8_t* __restrict__ masks, int N, long HW, unsigned* out) {
    const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long base = ((long)blockIdx.x * 4 + wave) * R * 1024 + lane * 16;
    const uint8_t* src = masks + (long)b * N * HW + base;
    unsigned acc = 0;
    for (int n = 0; n < N; ++n) {
        uint4 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = base + r * 1024 < HW ? *reinterpret_cast<const uint4*>(src + (long)n * HW + r * 1024) : make_uint4(0, 0, 0, 0);
        unsigned cnt = 0, sx = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const unsigned w[4] = {v[r].x, v[r].y, v[r].z, v[r].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (MODE == 0) { cnt ^= w[j]; continue; }
                const unsigned nz = ((((w[j] & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w[j]) & 0x80808080u) >> 7;
                const unsigned k = __popc(nz);
                cnt += k;
                sx += (unsigned)(4 * j + r) * k + ((nz >> 8) & 1u) + 2u * ((nz >> 16) & 1u) + 3u * ((nz >> 24) & 1u);
            }
        }
        if (MODE == 2) {
            unsigned sy = cnt * 3u;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { cnt += __shfl_down(cnt, o); sy += __shfl_down(sy, o); sx += __shfl_down(sx, o); }
            acc += sy;
        }
        acc += cnt + sx;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <class F>
static float time_ms(F f, int iters = 20) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 16, N = argc > 2 ? atoi(argv[2]) : 20;
    const long HW = 480L * 640, bytes = (long)B * N * HW;
    uint8_t* d; unsigned* out; uint8_t* flush;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&flush, 1L << 30));
    CK(hipMemset(d, 1, bytes));
    auto report = [&](const char* name, float ms) { printf("%-10s B=%d N=%d  %8.1f us  %7.0f GB/s\n", name, B, N, ms * 1e3, bytes / ms / 1e6); };
    // back to back (the Infinity Cache may keep a 98 MB stream) and with a 1 GiB write in between (HBM-cold)
    for (int cold = 0; cold < 2; ++cold) {
        printf("---- %s\n", cold ? "after a 1 GiB memset (cold)" : "back to back (warm caches)");
        auto wrap = [&](auto k) { return [=]() { if (cold) (void)hipMemsetAsync(flush, 0, 1L << 30, 0); k(); }; };
        auto base_ms = cold ? time_ms([&]() { (void)hipMemsetAsync(flush, 0, 1L << 30, 0); }) : 0.f;
        report("copy", time_ms(wrap([=]() { hipLaunchKernelGGL(copy_k, dim3(2048), dim3(256), 0, 0, (const uint4*)d, bytes / 16, out); })) - base_ms);
        report("planes", time_ms(wrap([=]() { hipLaunchKernelGGL((planes_k<1, 0>), dim3((unsigned)((HW + 4095) / 4096), B), dim3(256), 0, 0, d, N, HW, out); })) - base_ms);
        report("planes4", time_ms(wrap([=]() { hipLaunchKernelGGL((planes_k<4, 0>), dim3((unsigned)((HW + 16383) / 16384), B), dim3(256), 0, 0, d, N, HW, out); })) - base_ms);
        report("decode4", time_ms(wrap([=]() { hipLaunchKernelGGL((planes_k<4, 1>), dim3((unsigned)((HW + 16383) / 16384), B), dim3(256), 0, 0, d, N, HW, out); })) - base_ms);
        report("reduce4", time_ms(wrap([=]() { hipLaunchKernelGGL((planes_k<4, 2>), dim3((unsigned)((HW + 16383) / 16384), B), dim3(256), 0, 0, d, N, HW, out); })) - base_ms);
        report("reduce1", time_ms(wrap([=]() { hipLaunchKernelGGL((planes_k<1, 2>), dim3((unsigned)((HW + 4095) / 4096), B), dim3(256), 0, 0, d, N, HW, out); })) - base_ms);
    }
    return 0;
}
