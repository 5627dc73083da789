#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
// probe: raw_buffer_load_lds 16 B per lane; out-of-range lanes: zeros or untouched?
__global__ void probe(const uint32_t* src, int bytes, uint32_t* dst) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[64 * 4 * 2];
    const int l = threadIdx.x;
    for (int i = l; i < 512; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(src), 0, bytes, 0x00020000);
    // lanes 0..31 in range (reversed 16-byte chunks), lanes 32..63 out of range
    int voff = l < 32 ? (31 - l) * 16 : (int)0x80000000;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
    // second instruction: soffset use, all in range, to lds + 1024
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + 256), 16, (l & 31) * 16, 512, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = l; i < 512; i += 64) dst[i] = lds[i];
}
int main() {
    uint32_t h[512], *d, *o, r[512];
    for (int i = 0; i < 512; ++i) h[i] = i + 1;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(h));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d, 1024, o);
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    printf("lane0 chunk: %u %u %u %u (expect 125..128)\n", r[0], r[1], r[2], r[3]);
    printf("lane31 chunk: %u %u %u %u (expect 1..4)\n", r[124], r[125], r[126], r[127]);
    printf("lane32 (OOB) chunk: %08x %08x %08x %08x\n", r[128], r[129], r[130], r[131]);
    printf("lane63 (OOB) chunk: %08x %08x\n", r[252], r[255]);
    printf("second: lane0 %u (expect 129) lane31 %u (expect 253) lane32 %u (expect 129)\n", r[256], r[256 + 124], r[256 + 128]);
    return 0;
}
