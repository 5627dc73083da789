// What issues in the shadow of v_mfma_f32_32x32x2_f32 on gfx950 (one wave per SIMD)?  Cycles (s_memtime) of a dependent chain
// of 64 MFMAs with K filler instructions of one kind after each, against the bare chain.
//   build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_f32_shadow.hip -o /tmp/mfma_shadow ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

#define REP8(x) x x x x x x x x
#define MF "v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n"

template <int KIND, int K>
__global__ __launch_bounds__(256) void kern(unsigned long long* out, float* sink, float av, float bv) {
    __shared__ float lds[4096];
    lds[threadIdx.x] = av;
    __syncthreads();
    f32x16 acc = {};
    float f0 = av, f1 = bv, f2 = av + 1, f3 = bv + 2, f4 = av * 3, f5 = 1.f, f6 = 2.f, f7 = 3.f;
    f32x2 p0 = {av, bv}, p1 = {bv, av}, p2 = {1.f, 2.f}, p3 = {3.f, 4.f};
    unsigned i0 = threadIdx.x, i1 = 3, i2 = 5, i3 = 7;
    float l0 = 0.f;
    const float* lp = lds + (threadIdx.x & 63) * 4;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 64; ++it) {
        asm volatile(MF : "+a"(acc) : "v"(av), "v"(bv));
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (KIND == 1) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(k & 1 ? f0 : f2) : "v"(f5), "v"(f6)); }
            if (KIND == 2) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(k & 1 ? p0 : p1) : "v"(p2), "v"(p3)); }
            if (KIND == 3) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(k & 1 ? i0 : i1) : "v"(i2)); }
            if (KIND == 4) { asm volatile("s_nop 0"); }
            if (KIND == 5) { asm volatile("ds_read_b32 %0, %1" : "=v"(l0) : "v"((unsigned)(size_t)lp)); }
            if (KIND == 6) { asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(f7) : "a"(acc[15])); }
            if (KIND == 7) { asm volatile("v_mov_b32 %0, %1" : "=v"(k & 1 ? f3 : f4) : "v"(f5)); }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    asm volatile("s_nop 7\ns_nop 7\ns_nop 7");
    float r = acc[0];
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    sink[threadIdx.x] = r + f0 + f2 + p0.x + p1.y + (float)(i0 + i1) + l0 + f7 + f3 + f4;
}

template <int KIND, int K>
void run(const char* name, unsigned long long* d, float* s) {
    unsigned long long h = 0;
    for (int i = 0; i < 3; ++i) {
        hipLaunchKernelGGL((kern<KIND, K>), dim3(256), dim3(256), 0, 0, d, s, 1.f, 2.f);
        hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    }
    printf("%-28s K=%2d  %6llu cycles for 64 MFMAs = %.1f per MFMA\n", name, K, h, h / 64.0);
}

int main() {
    unsigned long long* d; float* s;
    hipMalloc(&d, 8); hipMalloc(&s, 4096);
    run<0, 0>("bare dependent chain", d, s);
    run<1, 4>("v_fma_f32", d, s); run<1, 8>("v_fma_f32", d, s); run<1, 14>("v_fma_f32", d, s); run<1, 20>("v_fma_f32", d, s);
    run<2, 2>("v_pk_fma_f32", d, s); run<2, 4>("v_pk_fma_f32", d, s); run<2, 8>("v_pk_fma_f32", d, s);
    run<3, 8>("v_add_u32", d, s); run<3, 14>("v_add_u32", d, s);
    run<4, 8>("s_nop 0", d, s); run<4, 14>("s_nop 0", d, s);
    run<5, 4>("ds_read_b32", d, s); run<5, 8>("ds_read_b32", d, s);
    run<6, 8>("v_accvgpr_read (other reg)", d, s);
    run<7, 8>("v_mov_b32", d, s); run<7, 14>("v_mov_b32", d, s);
    return 0;
}
