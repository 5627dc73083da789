// Evaluation support: the pairwise overlap counts behind eval/evaluation.py:57-274
// `multilabel_metrics`.  The reference loops over every (gt label, predicted label) pair and counts
// |gt_i & pred_j| with one full-frame pass each (evaluation.py:180-199); here ONE pass over the two label maps
// fills the whole contingency table, from which the host derives tp / P / R / F / IoU / union for all pairs.
//   1. presence:  which label values (0..65535) occur in each map
//   2. rank:      sorted unique labels (np.unique order) + value -> dense index LUT   (one block)
//   3. count:     table[gt index][pred index] += 1, LDS-privatised when the table fits
#include "common.h"

namespace quber {

constexpr int LAB_MAX = 65536;

__global__ void label_presence_kernel(const int* __restrict__ pred, const int* __restrict__ gt, long n,
                                      unsigned* __restrict__ flags /*[2][LAB_MAX]*/, int* __restrict__ bad) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int p = pred[i], g = gt[i];
        if ((unsigned)p >= LAB_MAX || (unsigned)g >= LAB_MAX) { *bad = 1; continue; }
        if (!flags[p]) flags[p] = 1;                 // benign race: every writer stores 1
        if (!flags[LAB_MAX + g]) flags[LAB_MAX + g] = 1;
    }
}

// one block of 1024 threads per map (blockIdx.x = 0 pred, 1 gt): ordered compaction of the set flags
__global__ __launch_bounds__(1024) void label_rank_kernel(const unsigned* __restrict__ flags, int cap,
                                                          unsigned short* __restrict__ lut /*[2][LAB_MAX]*/,
                                                          int* __restrict__ labels /*[2][cap]*/, int* __restrict__ counts) {
    __shared__ unsigned scan[1024];
    const int which = blockIdx.x, t = threadIdx.x;
    const unsigned* f = flags + (long)which * LAB_MAX;
    constexpr int SEG = LAB_MAX / 1024;
    unsigned n = 0;
    for (int i = 0; i < SEG; ++i) n += f[t * SEG + i] != 0;
    scan[t] = n;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const unsigned v = t >= o ? scan[t - o] : 0;
        __syncthreads();
        scan[t] += v;
        __syncthreads();
    }
    unsigned pos = scan[t] - n;
    for (int i = 0; i < SEG; ++i) {
        const int v = t * SEG + i;
        if (f[v]) {
            if ((int)pos < cap) {
                labels[which * cap + pos] = v;
                lut[(long)which * LAB_MAX + v] = (unsigned short)pos;
            }
            ++pos;
        }
    }
    if (t == 1023) counts[which] = (int)scan[1023];
}

__global__ __launch_bounds__(256) void contingency_kernel(const int* __restrict__ pred, const int* __restrict__ gt, long n,
                                                          const unsigned short* __restrict__ lut,
                                                          const int* __restrict__ counts, int cap,
                                                          unsigned long long* __restrict__ table /*[cap][cap]*/) {
    extern __shared__ unsigned priv[];
    const int np_ = counts[0], ng = counts[1];
    // counts[2]: label_presence_kernel met a value outside 0..65535 - the LUT cannot index it, the host raises
    if (np_ > cap || ng > cap || counts[2]) return;
    const bool use_lds = (long)np_ * ng <= 4096;
    if (use_lds) {
        for (int i = threadIdx.x; i < np_ * ng; i += 256) priv[i] = 0;
        __syncthreads();
    }
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int pv = pred[i], gv = gt[i];
        if ((unsigned)pv >= LAB_MAX || (unsigned)gv >= LAB_MAX) continue;     // unreachable once counts[2] is honoured
        const int pj = lut[pv], gi = lut[LAB_MAX + gv];
        if (use_lds) atomicAdd(&priv[gi * np_ + pj], 1u);
        else atomicAdd(&table[(long)gi * cap + pj], 1ull);
    }
    if (use_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < np_ * ng; i += 256)
            if (priv[i]) atomicAdd(&table[(long)(i / np_) * cap + (i % np_)], (unsigned long long)priv[i]);
    }
}

size_t contingency_ws_bytes(int cap) {
    return sizeof(unsigned) * 2 * LAB_MAX + sizeof(unsigned short) * 2 * LAB_MAX + sizeof(int) * (2 * cap + 4) +
           sizeof(unsigned long long) * (size_t)cap * cap;
}

// ws layout: flags u32[2][LAB_MAX] | table u64[cap][cap] | labels i32[2][cap] | counts i32[2] bad i32 pad | lut u16[2][LAB_MAX]
int launch_contingency(const int* pred, const int* gt, long n, int cap, void* ws, hipStream_t st) {
    char* w = reinterpret_cast<char*>(ws);
    unsigned* flags = reinterpret_cast<unsigned*>(w); w += sizeof(unsigned) * 2 * LAB_MAX;
    unsigned long long* table = reinterpret_cast<unsigned long long*>(w); w += sizeof(unsigned long long) * (size_t)cap * cap;
    int* labels = reinterpret_cast<int*>(w); w += sizeof(int) * 2 * cap;
    int* counts = reinterpret_cast<int*>(w); w += sizeof(int) * 4;
    unsigned short* lut = reinterpret_cast<unsigned short*>(w);
    if (int rc = launch_zero(ws, contingency_ws_bytes(cap), st)) return rc;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(label_presence_kernel, dim3(blocks), dim3(256), 0, st, pred, gt, n, flags, counts + 2);
    hipLaunchKernelGGL(label_rank_kernel, dim3(2), dim3(1024), 0, st, flags, cap, lut, labels, counts);
    hipLaunchKernelGGL(contingency_kernel, dim3(blocks), dim3(256), sizeof(unsigned) * 4096, st, pred, gt, n, lut, counts, cap,
                       table);
    QB_CHECK(hipGetLastError());
    return 0;
}

}  // namespace quber
