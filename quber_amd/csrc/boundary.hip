// Evaluation support (boundary half): the boundary precision / recall counts of
// eval/evaluation.py:21-54 `boundary_overlap` for every (ground-truth object, predicted object) pair, and the per-object
// boundary sizes of evaluation.py:165-175, in three launches on bit planes.
//
//   reference, per pair (i, j):  fg_b = seg2bmap(pred_j); gt_b = seg2bmap(gt_i);                    (utilities.py:672-697)
//                                gt_d = cv2.dilate(gt_b, disk(r)); fg_d = cv2.dilate(fg_b, disk(r))
//                                precision_tps = |fg_b & gt_d|,  recall_tps = |gt_b & fg_d|
//   seg2bmap = cv2.findContours(RETR_EXTERNAL, CHAIN_APPROX_NONE) + drawContours(thickness 1).  OpenCV is not in the image;
//   restated from the published border-following algorithm (Suzuki & Abe): with 8-connected objects the outer border of a
//   component is the set of its pixels that are 4-adjacent to the background region surrounding it, and RETR_EXTERNAL keeps
//   only the components that are not nested inside a hole of another one - i.e. those whose surrounding background is the
//   frame-connected ("outside") background.  Hence
//       outside = 4-connected flood of the background from the image frame
//       bmap    = seg & (a 4-neighbour is outside, the frame counting as outside)
//   Parity unpinned (hand-derived cases + scipy restatement in oracle/metrics_np.py).
//
//   1. pack   label map == label  ->  one bit per pixel (one __ballot per 64 pixels)
//   2. bmap   one workgroup per object: flood fill in LDS (whole-word run fills + one-row / one-word hops per sweep, until
//             a sweep changes nothing), border bits, disk dilation as row-wise shifted ORs, popcounts
//   3. pairs  one workgroup per (gt, pred): popcount(fg_b & gt_d), popcount(gt_b & fg_d)
#include "common.h"

namespace quber {

using u64 = unsigned long long;

static inline int bnd_wpr(int W) { return (W + 63) / 64; }

__global__ __launch_bounds__(256) void bnd_pack_kernel(const int* __restrict__ pred, const int* __restrict__ gt,
                                                       const int* __restrict__ labels, int n_pred, int H, int W, int wpr,
                                                       u64* __restrict__ seg) {
    const int m = blockIdx.y;
    const int* map = m < n_pred ? pred : gt;
    const int value = labels[m];
    const long word = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int y = (int)(word / wpr), wx = (int)(word - (long)y * wpr);
    const int x = wx * 64 + (threadIdx.x & 63);
    const bool on = y < H && x < W && map[(long)y * W + x] == value;
    const u64 bits = __ballot(on);
    if ((threadIdx.x & 63) == 0 && y < H) seg[((long)m * H + y) * wpr + wx] = bits;
}

// all bits reachable from `gen` by moving towards higher / lower bit positions through set bits of `pro` (within the word)
__device__ inline u64 fill_up(u64 gen, u64 pro) {
    gen |= pro & (gen << 1); pro &= pro << 1;
    gen |= pro & (gen << 2); pro &= pro << 2;
    gen |= pro & (gen << 4); pro &= pro << 4;
    gen |= pro & (gen << 8); pro &= pro << 8;
    gen |= pro & (gen << 16); pro &= pro << 16;
    gen |= pro & (gen << 32);
    return gen;
}
__device__ inline u64 fill_down(u64 gen, u64 pro) {
    gen |= pro & (gen >> 1); pro &= pro >> 1;
    gen |= pro & (gen >> 2); pro &= pro >> 2;
    gen |= pro & (gen >> 4); pro &= pro >> 4;
    gen |= pro & (gen >> 8); pro &= pro >> 8;
    gen |= pro & (gen >> 16); pro &= pro >> 16;
    gen |= pro & (gen >> 32);
    return gen;
}

// grid (n objects); LDS: reach[H][wpr]
__global__ __launch_bounds__(1024) void bnd_bmap_kernel(const u64* __restrict__ seg, int H, int W, int wpr, int radius,
                                                        u64* __restrict__ bmap, u64* __restrict__ dil,
                                                        unsigned* __restrict__ counts) {
    extern __shared__ u64 reach[];
    __shared__ int changed;
    __shared__ unsigned total;
    const int m = blockIdx.x, t = threadIdx.x, n = H * wpr;
    const u64* s = seg + (long)m * n;
    const u64 last_mask = (W & 63) ? ((1ull << (W & 63)) - 1) : ~0ull;
    auto bgw = [&](int i) { const int wx = i % wpr; return ~s[i] & (wx == wpr - 1 ? last_mask : ~0ull); };
    // seeds: background pixels of the first / last row and column
    for (int i = t; i < n; i += 1024) {
        const int y = i / wpr, wx = i - y * wpr;
        u64 seed = (y == 0 || y == H - 1) ? ~0ull : 0ull;
        if (wx == 0) seed |= 1ull;
        if (wx == wpr - 1) seed |= 1ull << ((W - 1) & 63);
        reach[i] = seed & bgw(i);
    }
    if (t == 0) total = 0;
    __syncthreads();
    for (int sweep = 0; sweep < H * W; ++sweep) {        // monotone: every sweep but the last adds a pixel
        if (t == 0) changed = 0;
        __syncthreads();
        bool any = false;
        for (int i = t; i < n; i += 1024) {
            const int y = i / wpr, wx = i - y * wpr;
            const u64 bg = bgw(i), old = reach[i];
            u64 r = old;
            if (y > 0) r |= reach[i - wpr];
            if (y + 1 < H) r |= reach[i + wpr];
            if (wx > 0) r |= reach[i - 1] >> 63;
            if (wx + 1 < wpr) r |= reach[i + 1] << 63;
            r &= bg;
            r = fill_up(r, bg);
            r = fill_down(r, bg);
            if (r != old) { reach[i] = r; any = true; }      // racing readers see old or new: both are valid reached sets
        }
        if (any) changed = 1;
        __syncthreads();
        const int c = changed;
        __syncthreads();
        if (!c) break;
    }
    // border pixels: object pixels with an outside 4-neighbour (beyond the image counts as outside)
    unsigned cnt = 0;
    u64* bm = bmap + (long)m * n;
    for (int i = t; i < n; i += 1024) {
        const int y = i / wpr, wx = i - y * wpr;
        const u64 r = reach[i];
        u64 nb = (r << 1) | (r >> 1);
        nb |= wx > 0 ? reach[i - 1] >> 63 : 1ull;
        nb |= wx + 1 < wpr ? reach[i + 1] << 63 : 0ull;
        if (wx == wpr - 1) nb |= 1ull << ((W - 1) & 63);                 // right of the last column: outside
        nb |= y > 0 ? reach[i - wpr] : ~0ull;
        nb |= y + 1 < H ? reach[i + wpr] : ~0ull;
        const u64 b = s[i] & nb & (wx == wpr - 1 ? last_mask : ~0ull);
        bm[i] = b;
        cnt += __popcll(b);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
    if ((t & 63) == 0 && cnt) atomicAdd(&total, cnt);
    __syncthreads();                                                     // bm complete (this block wrote it), total summed
    if (t == 0) counts[m] = total;
    // dilation by skimage.morphology.disk(radius): OR of the rows dy away, each spread by floor(sqrt(r^2 - dy^2)) columns
    u64* dl = dil + (long)m * n;
    for (int i = t; i < n; i += 1024) {
        const int y = i / wpr, wx = i - y * wpr;
        u64 acc = 0;
        for (int dy = -radius; dy <= radius; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= H) continue;
            int ext = 0;
            while ((ext + 1) * (ext + 1) + dy * dy <= radius * radius) ++ext;
            const u64* row = bm + (long)yy * wpr;
            const u64 c = row[wx], l = wx > 0 ? row[wx - 1] : 0ull, rr = wx + 1 < wpr ? row[wx + 1] : 0ull;
            u64 v = c;
            for (int sft = 1; sft <= ext; ++sft) v |= (c << sft) | (l >> (64 - sft)) | (c >> sft) | (rr << (64 - sft));
            acc |= v;
        }
        dl[i] = acc & (wx == wpr - 1 ? last_mask : ~0ull);
    }
}

// grid (n_pred, n_gt): out_fg[i][j] = |bmap_pred_j & dil_gt_i|, out_gt[i][j] = |bmap_gt_i & dil_pred_j|
__global__ __launch_bounds__(256) void bnd_pair_kernel(const u64* __restrict__ bmap, const u64* __restrict__ dil, int n,
                                                       int n_pred, unsigned* __restrict__ out_fg, unsigned* __restrict__ out_gt) {
    __shared__ unsigned acc[2];
    const int j = blockIdx.x, i = blockIdx.y;
    if (threadIdx.x < 2) acc[threadIdx.x] = 0;
    __syncthreads();
    const u64 *bp = bmap + (long)j * n, *dp = dil + (long)j * n;
    const u64 *bg = bmap + (long)(n_pred + i) * n, *dg = dil + (long)(n_pred + i) * n;
    unsigned a = 0, b = 0;
    for (int k = threadIdx.x; k < n; k += 256) {
        a += __popcll(bp[k] & dg[k]);
        b += __popcll(bg[k] & dp[k]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_down(a, o);
        b += __shfl_down(b, o);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&acc[0], a);
        atomicAdd(&acc[1], b);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out_fg[(long)i * n_pred + j] = acc[0];
        out_gt[(long)i * n_pred + j] = acc[1];
    }
}

size_t boundary_ws_bytes(int H, int W, int n_masks) { return (size_t)3 * n_masks * H * bnd_wpr(W) * sizeof(u64); }

// out: counts[n_pred + n_gt] | fg_match[n_gt][n_pred] | gt_match[n_gt][n_pred]   (u32)
int launch_boundary_overlap(const int* pred, const int* gt, int H, int W, const int* labels, int n_pred, int n_gt, int radius,
                            void* ws, unsigned* out, hipStream_t st) {
    const int wpr = bnd_wpr(W), nm = n_pred + n_gt;
    const size_t plane = (size_t)H * wpr * sizeof(u64);
    if (n_pred < 1 || n_gt < 1) return fail("boundary overlap: needs at least one object on each side");
    if (radius < 0 || radius > 63) return fail("boundary overlap: dilation radius must be 0..63");
    if (plane > 160 * 1024 - 64) return fail("boundary overlap: frame too large for the LDS flood fill (H * ceil(W/64) * 8 <= 160 KiB)");
    u64* seg = reinterpret_cast<u64*>(ws);
    u64* bmap = seg + (size_t)nm * H * wpr;
    u64* dil = bmap + (size_t)nm * H * wpr;
    hipLaunchKernelGGL(bnd_pack_kernel, dim3((unsigned)(((long)H * wpr + 3) / 4), nm), dim3(256), 0, st, pred, gt, labels, n_pred, H,
                       W, wpr, seg);
    QB_CHECK(hipGetLastError());
    // the attribute is per device (and this entry point may be called from several threads): set it on every call
    QB_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(bnd_bmap_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 160 * 1024 - 64));
    hipLaunchKernelGGL(bnd_bmap_kernel, dim3(nm), dim3(1024), plane, st, seg, H, W, wpr, radius, bmap, dil, out);
    QB_CHECK(hipGetLastError());
    hipLaunchKernelGGL(bnd_pair_kernel, dim3(n_pred, n_gt), dim3(256), 0, st, bmap, dil, H * wpr, n_pred, out + nm,
                       out + nm + (size_t)n_gt * n_pred);
    QB_CHECK(hipGetLastError());
    return 0;
}

}  // namespace quber
