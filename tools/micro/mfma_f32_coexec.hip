// Does a SECOND wave's vector / LDS work execute beside v_mfma_f32_32x32x2_f32 on gfx950?  (tools/micro/mfma_f32_shadow.hip: inside ONE
// wave it does not - every vector instruction after an fp32 MFMA adds its ~8 cycles.)  Blocks of 8 waves = 2 per SIMD (waves w and
// w + 4 share a SIMD); waves 0-3 run independent MFMA chains, waves 4-7 a stream of one kind of other instruction.  Kernel time of
// {MFMA waves alone, other waves alone, both}: both == max -> they overlap, both == sum -> they share the lanes.
//   build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_f32_coexec.hip -o tools/micro/mfma_coexec ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

// MF: 0 = fp32 32x32x2 (16 passes), 1 = fp32 16x16x4 (8 passes), 2 = f16 32x32x16 (reference: known to overlap)
template <int MF, int KIND>
__global__ __launch_bounds__(512) void kern(float* sink, float av, float bv, int n_mfma, int n_other) {
    __shared__ float lds[8192];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    lds[threadIdx.x] = av;
    __syncthreads();
    float r = 0.f;
    if (wave < 4) {
        f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
        f32x4 c0 = {}, c1 = {}, c2 = {}, c3 = {};
        bf16x8 ba = {(__bf16)av, 1, 2, 3, 4, 5, 6, 7}, bb = {(__bf16)bv, 1, 2, 3, 4, 5, 6, 7};
        f16x8 ha = {(_Float16)av, 1, 2, 3, 4, 5, 6, 7}, hb = {(_Float16)bv, 1, 2, 3, 4, 5, 6, 7};
        for (int it = 0; it < n_mfma; ++it) {
            if (MF == 0) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, a3, 0, 0, 0);
            } else if (MF == 1) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, c3, 0, 0, 0);
            } else if (MF == 3) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, a3, 0, 0, 0);
            } else if (MF == 4) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c3, 0, 0, 0);
            } else {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, a3, 0, 0, 0);
            }
        }
        r = a0[0] + a1[1] + a2[2] + a3[3] + c0[0] + c1[1] + c2[2] + c3[3];
    } else {
        f32x2 p0 = {av, bv}, p1 = {bv, av}, p2 = {av, av}, p3 = {bv, bv};
        const f32x2 k0 = {1.0001f, 0.9999f}, k1 = {1e-6f, -1e-6f};
        float f0 = av, f1 = bv, f2 = av + 1.f, f3 = bv + 1.f;
        unsigned i0 = lane, i1 = 3, i2 = 5, i3 = 7;
        f32x4 l0 = {}, l1 = {};
        const unsigned lp = (unsigned)(size_t)(lds + lane * 4);
        for (int it = 0; it < n_other; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (KIND == 1) {
                    asm volatile("v_pk_fma_f32 %0, %0, %4, %5\nv_pk_fma_f32 %1, %1, %4, %5\nv_pk_fma_f32 %2, %2, %4, %5\nv_pk_fma_f32 %3, %3, %4, %5"
                                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(k0), "v"(k1));
                } else if (KIND == 2) {
                    asm volatile("v_fma_f32 %0, %0, %4, %5\nv_fma_f32 %1, %1, %4, %5\nv_fma_f32 %2, %2, %4, %5\nv_fma_f32 %3, %3, %4, %5"
                                 : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(k0.x), "v"(k1.x));
                } else if (KIND == 3) {
                    asm volatile("v_add_u32 %0, %0, %4\nv_add_u32 %1, %1, %4\nv_add_u32 %2, %2, %4\nv_add_u32 %3, %3, %4"
                                 : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(i2));
                } else if (KIND == 4) {
                    asm volatile("ds_read_b128 %0, %2\nds_read_b128 %1, %2 offset:4096\ns_waitcnt lgkmcnt(0)" : "=v"(l0), "=v"(l1) : "v"(lp));
                } else if (KIND == 5) {
                    asm volatile("ds_write_b64 %0, %1\nds_write_b64 %0, %1 offset:4096\ns_waitcnt lgkmcnt(0)" : : "v"(lp), "v"(p0));
                }
#define Q4(OP) asm volatile(OP " %0, %0, %4\n" OP " %1, %1, %4\n" OP " %2, %2, %4\n" OP " %3, %3, %4" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(k0.x))
#define Q4I(OP) asm volatile(OP " %0, %4, %0\n" OP " %1, %4, %1\n" OP " %2, %4, %2\n" OP " %3, %4, %3" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(i2 & 7))
                else if (KIND == 6) Q4("v_sub_f32");
                else if (KIND == 7) Q4("v_and_b32");
                else if (KIND == 8) Q4I("v_lshlrev_b32");
                else if (KIND == 9) Q4("v_cvt_pk_bf16_f32");
                else if (KIND == 10) asm volatile("v_perm_b32 %0, %0, %1, %4\nv_perm_b32 %1, %1, %2, %4\nv_perm_b32 %2, %2, %3, %4\nv_perm_b32 %3, %3, %0, %4" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(0x07060302u));
                else if (KIND == 11) asm volatile("v_pk_add_f32 %0, %0, %4\nv_pk_add_f32 %1, %1, %4\nv_pk_add_f32 %2, %2, %4\nv_pk_add_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(k1));
                else if (KIND == 12) asm volatile("v_pk_fma_f16 %0, %0, %4, %5\nv_pk_fma_f16 %1, %1, %4, %5\nv_pk_fma_f16 %2, %2, %4, %5\nv_pk_fma_f16 %3, %3, %4, %5" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(k0.x), "v"(k1.x));
                else if (KIND == 13) Q4("v_pk_max_f16");
                else if (KIND == 14) Q4("v_max_f32");
                else if (KIND == 15) Q4("v_cvt_pkrtz_f16_f32");
                else if (KIND == 16) asm volatile("v_dot2_f32_f16 %0, %4, %4, %0\nv_dot2_f32_f16 %1, %4, %4, %1\nv_dot2_f32_f16 %2, %4, %4, %2\nv_dot2_f32_f16 %3, %4, %4, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(k0.x));
                else if (KIND == 17) asm volatile("v_mov_b32 %0, %4\nv_mov_b32 %1, %4\nv_mov_b32 %2, %4\nv_mov_b32 %3, %4" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : "v"(k0.x));
                else if (KIND == 18) Q4("v_mul_f32");
                else if (KIND == 19) Q4("v_add_f32");
            }
        }
        r = p0.x + p1.y + p2.x + p3.y + f0 + f1 + f2 + f3 + (float)(i0 + i1 + i2 + i3) + l0.x + l1.y;
    }
    if (r == 12345.678f) sink[threadIdx.x] = r;
}

template <int MF, int KIND>
float run(float* s, int n_mfma, int n_other) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int i = 0; i < 4; ++i) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((kern<MF, KIND>), dim3(256), dim3(512), 0, 0, s, 1.f, 2.f, n_mfma, n_other);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (i && ms < best) best = ms;
    }
    return best;
}

template <int MF, int KIND>
void trio(const char* mf, const char* kind, float* s, int n_mfma, int n_other) {
    const float a = run<MF, KIND>(s, n_mfma, 0), b = run<MF, KIND>(s, 0, n_other), c = run<MF, KIND>(s, n_mfma, n_other);
    printf("%-22s + %-14s  mfma alone %.3f ms  other alone %.3f ms  both %.3f ms   (max %.3f, sum %.3f)  overlap %.2f\n", mf, kind, a, b, c,
           a > b ? a : b, a + b, (a + b - c) / (a < b ? a : b));
}

int main() {
    float* s; hipMalloc(&s, 4096);
    const int NM = 20000;     // x 4 MFMAs
    trio<0, 1>("f32 32x32x2", "v_pk_fma_f32", s, NM, 40000);
    trio<0, 2>("f32 32x32x2", "v_fma_f32", s, NM, 40000);
    trio<0, 3>("f32 32x32x2", "v_add_u32", s, NM, 40000);
    trio<0, 4>("f32 32x32x2", "ds_read_b128", s, NM, 20000);
    trio<0, 5>("f32 32x32x2", "ds_write_b64", s, NM, 20000);
    trio<1, 1>("f32 16x16x4", "v_pk_fma_f32", s, 2 * NM, 40000);
    trio<1, 4>("f32 16x16x4", "ds_read_b128", s, 2 * NM, 20000);
    trio<2, 1>("f16 32x32x16", "v_pk_fma_f32", s, NM, 40000);
    trio<2, 3>("f16 32x32x16", "v_add_u32", s, NM, 40000);
    trio<2, 4>("f16 32x32x16", "ds_read_b128", s, NM, 20000);
#define T3(K, N) trio<3, K>("bf16 32x32x16", N, s, NM, 40000)
    T3(2, "v_fma_f32"); T3(6, "v_sub_f32"); T3(18, "v_mul_f32"); T3(19, "v_add_f32"); T3(7, "v_and_b32"); T3(8, "v_lshlrev_b32"); T3(9, "v_cvt_pk_bf16_f32"); T3(10, "v_perm_b32"); T3(11, "v_pk_add_f32"); T3(14, "v_max_f32"); T3(17, "v_mov_b32");
#define T4(K, N) trio<4, K>("f16 16x16x32", N, s, 2 * NM, 40000)
    T4(2, "v_fma_f32"); T4(12, "v_pk_fma_f16"); T4(13, "v_pk_max_f16"); T4(14, "v_max_f32"); T4(15, "v_cvt_pkrtz"); T4(16, "v_dot2_f32_f16"); T4(3, "v_add_u32"); T4(17, "v_mov_b32"); T4(1, "v_pk_fma_f32");
    trio<4, 4>("f16 16x16x32", "ds_read_b128", s, 2 * NM, 20000);
    return 0;
}
