// Explicit quadruple error maps (region + boundary TP/TN/FP/FN) of a set of initial masks against a
// set of ground-truth masks.  Replaces
//   explicit_error_estimation/util.py:62-68   masks_to_fg_mask   (uint8 wrap-around sum > 0)
//   explicit_error_estimation/util.py:72-90   mask_to_boundary   (3x3 erosion x d on the zero-padded mask)
//   explicit_error_estimation/util.py:92-99   masks_to_boundary
//   tools/ours/panoptic2eee.py:110-123        TP/TN/FP/FN
// d iterations of a 3x3 erosion with a zero border equal one (2d+1)^2 minimum filter, evaluated
// separably (row minimum, then column minimum) on an LDS tile with a d-pixel halo.
#include "common.h"

namespace quber {

constexpr int ET_H = 32, ET_W = 64;

// grid (tiles_x, tiles_y, B); writes fg[b][p] and bnd[b][p] in {0,1}
__global__ __launch_bounds__(256) void fg_boundary_kernel(const uint8_t* __restrict__ masks, int N, int H, int W, int d,
                                                          uint8_t* __restrict__ fg, uint8_t* __restrict__ bnd) {
    extern __shared__ uint8_t tile[];  // s0[(ET_H+2d)][(ET_W+2d)], s1[(ET_H+2d)][ET_W]
    const int PW = ET_W + 2 * d, PH = ET_H + 2 * d;
    uint8_t* s0 = tile;
    uint8_t* s1 = tile + PH * PW;
    const int b = blockIdx.z;
    const int ty0 = blockIdx.y * ET_H, tx0 = blockIdx.x * ET_W;
    const long HW = (long)H * W;
    constexpr int PPT = ET_H * ET_W / 256;  // pixels per thread
    unsigned accfg[PPT], accb[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) accfg[i] = accb[i] = 0;

    for (int n = 0; n < N; ++n) {
        const uint8_t* m = masks + ((long)b * N + n) * HW;
        for (int i = threadIdx.x; i < PH * PW; i += 256) {
            const int ly = i / PW, lx = i - ly * PW;
            const int gy = ty0 + ly - d, gx = tx0 + lx - d;
            s0[i] = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? m[(long)gy * W + gx] : 0;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < PH * ET_W; i += 256) {
            const int ly = i / ET_W, lx = i - ly * ET_W;
            const uint8_t* r = s0 + ly * PW + lx;
            unsigned v = 255;
            for (int k = 0; k <= 2 * d; ++k) v = min(v, (unsigned)r[k]);
            s1[i] = (uint8_t)v;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int idx = threadIdx.x + 256 * i;
            const int ly = idx / ET_W, lx = idx - ly * ET_W;
            unsigned v = 255;
            for (int k = 0; k <= 2 * d; ++k) v = min(v, (unsigned)s1[(ly + k) * ET_W + lx]);
            const unsigned mv = s0[(ly + d) * PW + lx + d];
            accfg[i] += mv;
            accb[i] += (mv - v) & 0xffu;   // uint8 `mask - eroded`
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int idx = threadIdx.x + 256 * i;
        const int ly = idx / ET_W, lx = idx - ly * ET_W;
        const int gy = ty0 + ly, gx = tx0 + lx;
        if (gy < H && gx < W) {
            const long o = (long)b * HW + (long)gy * W + gx;
            fg[o] = (accfg[i] & 0xffu) ? 1 : 0;    // the reference accumulates in uint8
            bnd[o] = (accb[i] & 0xffu) ? 1 : 0;
        }
    }
}

// ws = [gt_fg, in_fg, gt_bnd, in_bnd] each B*H*W ; out[b][2][4][H][W]
__global__ void quadruple_kernel(const uint8_t* __restrict__ ws, long BHW, long HW, uint8_t* __restrict__ out) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < BHW; i += (long)gridDim.x * blockDim.x) {
        const long b = i / HW, p = i - b * HW;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const bool g = ws[(2 * k) * BHW + i] != 0, in = ws[(2 * k + 1) * BHW + i] != 0;
            uint8_t* o = out + ((b * 2 + k) * 4) * HW + p;
            o[0] = g && in;
            o[HW] = !g && !in;
            o[2 * HW] = !g && in;
            o[3 * HW] = g && !in;
        }
    }
}

size_t errmaps_ws_bytes(int B, int H, int W) { return (size_t)4 * B * H * W; }

int launch_errmaps(const uint8_t* init, int N, const uint8_t* gt, int Ng, int B, int H, int W, int d, uint8_t* ws,
                   uint8_t* out, hipStream_t st) {
    if (d < 1 || d > 32) return fail("error maps: boundary width out of range (1..32)");
    const long BHW = (long)B * H * W;
    const size_t sm = (size_t)(ET_H + 2 * d) * (ET_W + 2 * d) + (size_t)(ET_H + 2 * d) * ET_W;
    dim3 grid((W + ET_W - 1) / ET_W, (H + ET_H - 1) / ET_H, B);
    hipLaunchKernelGGL(fg_boundary_kernel, grid, dim3(256), sm, st, gt, Ng, H, W, d, ws, ws + 2 * BHW);
    hipLaunchKernelGGL(fg_boundary_kernel, grid, dim3(256), sm, st, init, N, H, W, d, ws + BHW, ws + 3 * BHW);
    QB_CHECK(hipGetLastError());
    int blocks = (int)((BHW + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(quadruple_kernel, dim3(blocks), dim3(256), 0, st, ws, BHW, (long)H * W, out);
    QB_CHECK(hipGetLastError());
    return 0;
}

}  // namespace quber
