// A 3x3 / stride 1 convolution as ONE Winograd F(4x4,3x3) kernel: the same arithmetic as winograd.hip's three-kernel pipeline -
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A     (reference: the plain F.conv2d of detectron2's Conv2d, see winograd.hip)
// - but the transformed activations V and the products M never exist in HBM.  The three-kernel form writes V (2.25x the
// layer's input), reads it back in the GEMM, writes M (2.25x the output) and reads that back: 27 GB of the 81 GB the
// convolution family moved per 16-frame step (profiles/r03_final_conv_hbm_traffic.json), and two HBM-bound passes of 5.2 ms.
// Used for the layers of up to 160 input channels (tuning key 27): there it beats the pipeline by 1.2-2.1x and, below 128
// channels, the direct kernel by 1.2-1.7x (profiles/r05_wino_fused_layers.md); DECISIONS.md section 4 has the measurements that
// shaped it.  Two kernels:
//   wino_fused64_kernel   16 tiles x 64 output channels per block, v_mfma_f32_16x16x4_f32, rounds of 32 input channels
//                         (64 | Cout and an even number of rounds)
//   wino_fused_kernel     32 tiles x 32 output channels per block, v_mfma_f32_32x32x2_f32, rounds of 16 input channels
//                         (the 32- and 96-channel outputs, the 32- / 96- / 160-channel inputs)
// Common structure.  A block owns its tiles x channels for ALL 36 transform positions and the whole K = Cin:
//   * accumulators: nine positions per wave, 144 registers per set - a block is one wave per SIMD with the whole 512-register
//     file; there is no second wave to hide latency behind, so everything is software-pipelined inside the wave, and since
//     vector instructions never execute beside the fp32 MFMA (SQ_VALU_MFMA_COEXEC_CYCLES = 0; ~8 cycles each,
//     tools/micro/mfma_f32_shadow.hip) the loop carries as few of them as possible: hand-packed transforms, scalar address
//     arithmetic, no second-level addition (two accumulator sets, one per K-slice parity, added once in the epilogue);
//   * thread (tile, channel pair) loads its 6x6 input patch straight from the NHWC tensor (8-byte buffer loads, range-checked:
//     the zero padding and the tiles past the end cost no predicates), transforms it in registers (B^T d B on channel pairs)
//     and stores the 36 values into the LDS image of the NEXT round while the matrix pipe multiplies the current one: 2 x 72 KB
//     of LDS, one barrier per round; the producer's GroupNorm + ReLU can be applied on the way (WinoNorm);
//   * the B operand (transformed filters) is packed in the order the MFMAs consume it (winograd_fused_pack_host) and goes
//     global -> registers, each wave loading only its nine positions;
//   * epilogue: the sums go through LDS once (the 144 KB the two images occupied), thread (tile, 4 channels) applies A^T . A,
//     the affine, the ReLU, accumulates the GroupNorm sums and stores 16 pixels x 16 bytes.
// HBM traffic of a layer: its input (re-read per channel chunk through L2: the chunks of a tile block are neighbours on one
// XCD), its output, the filters.
#include <algorithm>

#include "common.h"
#include "winograd_xf.h"

namespace quber {


using namespace wxf;

// element i of the packed filter array [Cout/32][Cin/8][36][64 lanes][4] <- index into U [36][Cout][Cin]:
// lane l of the wave multiplying position p supplies output channel l % 32 and input channels 4 * (l / 32) + e of the slice
__host__ __device__ inline long fused_src(long i, int Cout, int Cin) {
    const int e = (int)(i & 3), l = (int)((i >> 2) & 63);
    long r = i >> 8;
    const int p = (int)(r % 36);
    r /= 36;
    const int ks = (int)(r % (Cin / 8)), cc = (int)(r / (Cin / 8));
    return ((long)p * Cout + cc * 32 + (l & 31)) * Cin + ks * 8 + 4 * (l >> 5) + e;
}

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

#ifndef WF_SPLIT
#define WF_SPLIT 7                              // positions (of a wave's 9) that get an accumulator set per slice; see `SPLIT` in the kernels
#endif
#ifndef WF_SKIP
#define WF_SKIP 0                               // diagnostic builds of the 32 x 32 kernel only (tools/wino_fused_ablate.sh): bit 0 no patch loads, 1 no filter loads, 2 no transforms, 4 no MFMA, 5 no epilogue, 6 no K loop
#endif
#ifndef WF_OPT
#define WF_OPT 7                                // A/B switches of the round-4 diet (profiles/r09d_wino_fused_diet.md): bit 0 = no transform work for rounds that do not
#endif                                          // exist (the last two rounds of the K loop; 16 x 64 kernel only: in the 32 x 32 kernel the branches cost 20 %), bit 1 = MFMA operands
                                                // swapped (filters as A): a lane's result registers are 4 consecutive CHANNELS of one tile, the epilogue stores them to LDS 16 bytes
                                                // at a time (36 instead of 144 stores), bit 2 = GroupNorm sums of a thread's 64 outputs in packed fp32, converted to double once
#ifdef WF_STAMPS
// diagnostic build (make WFX=-DWF_STAMPS, tools/wf_stamps.py): s_memtime at the phase boundaries of the first blocks
constexpr int WF_STAMP_BLOCKS = 512, WF_STAMP_N = 16;
__device__ unsigned long long g_wf_stamps[WF_STAMP_BLOCKS * 4 * WF_STAMP_N];
#define WF_STAMP(i) do { if (lane == 0 && blockIdx.x < WF_STAMP_BLOCKS && blockIdx.z == 0) g_wf_stamps[((int)blockIdx.x * 4 + wave) * WF_STAMP_N + (i)] = (i) == 15 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WF_STAMP(i) do {} while (0)
#endif
constexpr int FT = 32;                          // tiles per block
constexpr int FC = 32;                          // output channels per block
constexpr int FK = 16;                          // input channels per round: two 8-channel MFMA slices
constexpr int FP = 36;                          // transform positions
constexpr int SLICE = FP * FT * 8;              // floats of one 8-channel slice image [position][tile][8]
constexpr int STAGE = 2 * SLICE;                // ... of one round
constexpr int M_FLOATS = FP * FT * FC;          // epilogue image [position][tile][32]  (== 2 * STAGE)
constexpr int SMEM_BYTES = 2 * STAGE * 4 + 2 * 32 * 2 * 8;
static_assert(M_FLOATS == 2 * STAGE, "the epilogue image overlays the two round images");
constexpr unsigned OOBH = 0x40000000u;          // added once per invalid row and once per invalid column: past any view (host check)
constexpr int RSRC_FLAGS = 0x00020000;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* ptr, unsigned bytes) {
    const unsigned long a = (unsigned long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long)hi << 32) | lo), 0,
                                             (int)__builtin_amdgcn_readfirstlane(bytes), RSRC_FLAGS);
}
__device__ __forceinline__ f32x2 buf_load2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_store4(f32x4 v, __amdgpu_buffer_rsrc_t r, unsigned voff) {
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, 0, 0);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// y = B^T x on a channel pair (points 0, +-3/4, +-3/2, inf; the constants are dyadic), 12 fused multiply-adds on packed fp32.
// Written as v_pk_fma_f32 by hand: left to instruction selection, half of these operations come out as two scalar ones -
// and beside the fp32 MFMA every vector instruction costs its full issue time (tools/micro/mfma_f32_shadow.hip).
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 c, f32x2 b) {        // a * c + b, c = a uniform constant pair in scalar registers
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(c), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pk_fnma(f32x2 a, f32x2 c, f32x2 b) {       // b - a * c
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,1,0] neg_hi:[0,1,0]" : "=v"(d) : "v"(a), "s"(c), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pk_fma_v(f32x2 a, f32x2 c, f32x2 b) {      // a * c + b, all in vector registers
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(c), "v"(b));
    return d;
}
__device__ __forceinline__ void bt4(const f32x2* x, f32x2* y) {
    const f32x2 k225 = {2.25f, 2.25f}, k05625 = {0.5625f, 0.5625f}, k075 = {0.75f, 0.75f}, k15 = {1.5f, 1.5f};
    const f32x2 k1265625 = {1.265625f, 1.265625f}, k28125 = {2.8125f, 2.8125f};
    const f32x2 t1 = pk_fnma(x[1], k225, x[3]), e1 = pk_fnma(x[2], k225, x[4]);
    const f32x2 t2 = pk_fnma(x[1], k05625, x[3]), e2 = pk_fnma(x[2], k05625, x[4]);
    y[0] = pk_fma(x[0], k1265625, pk_fnma(x[2], k28125, x[4]));
    y[1] = pk_fma(t1, k075, e1);
    y[2] = pk_fnma(t1, k075, e1);
    y[3] = pk_fma(t2, k15, e2);
    y[4] = pk_fnma(t2, k15, e2);
    y[5] = pk_fma(x[1], k1265625, pk_fnma(x[3], k28125, x[5]));
}

// y = A^T x on four channels (two packed pairs), 12 packed operations per pair: the epilogue's output transform
struct PV { f32x2 lo, hi; };
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 c) {                 // c = a uniform constant pair in scalar registers
    f32x2 d;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "s"(c));
    return d;
}
__device__ __forceinline__ f32x2 at4_y(int o, f32x2 x0, f32x2 a, f32x2 b, f32x2 c, f32x2 d, f32x2 x5) {
    const f32x2 k075 = {0.75f, 0.75f}, k15 = {1.5f, 1.5f}, k05625 = {0.5625f, 0.5625f}, k225 = {2.25f, 2.25f};
    const f32x2 k0421875 = {0.421875f, 0.421875f}, k3375 = {3.375f, 3.375f};
    if (o == 0) return pk_add(pk_add(x0, a), c);
    if (o == 1) return pk_fma(d, k15, pk_mul(b, k075));
    if (o == 2) return pk_fma(c, k225, pk_mul(a, k05625));
    return pk_fma(b, k0421875, pk_fma(d, k3375, x5));
}
__device__ __forceinline__ void at4(const PV* x, PV* y) {
    const f32x2 al = pk_add(x[1].lo, x[2].lo), bl = pk_sub(x[1].lo, x[2].lo), cl = pk_add(x[3].lo, x[4].lo), dl = pk_sub(x[3].lo, x[4].lo);
    const f32x2 ah = pk_add(x[1].hi, x[2].hi), bh = pk_sub(x[1].hi, x[2].hi), ch = pk_add(x[3].hi, x[4].hi), dh = pk_sub(x[3].hi, x[4].hi);
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        y[o].lo = at4_y(o, x[0].lo, al, bl, cl, dl, x[5].lo);
        y[o].hi = at4_y(o, x[0].hi, ah, bh, ch, dh, x[5].hi);
    }
}
__device__ __forceinline__ PV ldpv(const float* p) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
    return PV{f32x2{t.x, t.y}, f32x2{t.z, t.w}};
}

struct FusedArgs {
    const float* in; long in_gs; int in_cs, H, W, Cin; unsigned in_bytes;
    const float* uf; long uf_gs;                 // packed filters [G][Cout/32][Cin/8][36][64 lanes][4]
    const float* coef; long coef_gs;             // fused GroupNorm of the input: [G][Ball][Cin][scale, bias], or null
    int relu_in;
    int Ball, boff, B;
    const float* scale; const float* shift; int ss_gs, relu;
    float* out; int out_cs; long out_gs; int Cout; unsigned out_bytes;
    int TH, TW, d; long tiles, per_img; int NC, NB;     // 32-channel chunks, blocks = tile blocks x NC
    double* gn_sum; int gn_groups, gn_cpg;
};

template <bool NORM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wino_fused_kernel(const FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    double* const gacc = reinterpret_cast<double*>(smem + 2 * STAGE);     // [image b0 / b0 + 1][norm group][sum, sum of squares]
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);        // wave-uniform: its offsets belong in scalar registers
    const int g = blockIdx.z;
    // XCD-aware order: blocks b and b + 8 share an XCD.  Each XCD takes one contiguous run of (tile block, channel
    // chunk) pairs, chunk fastest: the NC blocks that read the same input tiles run side by side on one L2.
    const int xcd = blockIdx.x & 7, bq_ = a.NB >> 3, br = a.NB & 7;
    const int vb = (xcd < br ? xcd * (bq_ + 1) : br * (bq_ + 1) + (xcd - br) * bq_) + (blockIdx.x >> 3);
    const int tb = vb / a.NC, cc = vb - tb * a.NC;
    if (t < 128) gacc[t] = 0.0;                  // published by the barriers of the K loop
    WF_STAMP(0);
    WF_STAMP(15);

    // ---- loader role: tile lt of the block, channel pair q of the round ----
    const int lt = t >> 3, q = t & 7;
    const long tile = (long)tb * FT + lt;
    const bool tvalid = tile < a.tiles;
    const TileAt ta = locate32(tvalid ? (unsigned)tile : 0u, (unsigned)a.TH, (unsigned)a.TW, (unsigned)a.d);
    const int y0 = a.d * (4 * ta.ty - 1) + ta.py, x0 = a.d * (4 * ta.tx - 1) + ta.px;
    const unsigned pix = (unsigned)a.in_cs * 4u;
    const unsigned base = (unsigned)((ta.b * a.H + y0) * a.W + x0) * pix + (unsigned)q * 8u;      // may be "negative": wraps, fixed by the valid offsets
    unsigned rowbase[6], colterm[6];
    bool edge = false;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const bool ok = tvalid && (unsigned)(y0 + i * a.d) < (unsigned)a.H;
        rowbase[i] = ok ? base + (unsigned)(i * a.d * a.W) * pix : OOBH;
        edge |= !ok;
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const bool ok = (unsigned)(x0 + j * a.d) < (unsigned)a.W;
        colterm[j] = ok ? (unsigned)(j * a.d) * pix : OOBH;
        edge |= !ok;
    }
    // some tile of this wave touches the zero padding (wave-uniform: the interior waves skip the masking of the fused GroupNorm)
    const bool border = NORM && __any(edge);
    const float relu_lo = a.relu_in ? 0.f : -__builtin_inff();      // ReLU of the fused GroupNorm as max(v, lo)
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(a.in + (long)g * a.in_gs, a.in_bytes);
    const int nslice = a.Cin / 8;
    const __amdgpu_buffer_rsrc_t rs_u = make_rsrc(a.uf + (long)g * a.uf_gs + (long)cc * nslice * (FP * 256), (unsigned)nslice * (FP * 1024u));
    const float* coef = NORM ? a.coef + (long)g * a.coef_gs + ((long)(a.boff + ta.b) * a.Cin + 2 * q) * 2 : nullptr;

    // The fp32 MFMA runs on the SIMD's fp32 lanes: vector instructions of the same wave do NOT execute beside it
    // (SQ_VALU_MFMA_COEXEC_CYCLES = 0, profiles/r05h_fused_pmc.txt), so every vector instruction of the loop adds its issue
    // time to the 64 cycles per MFMA.  Hence: the transforms on packed fp32 (channel pairs: half the instructions), and a
    // register budget that leaves the compiler no reason to shuffle values between the two register files.
    f32x2 dd[6][6];
    f32x4 cf = {1.f, 0.f, 1.f, 0.f};
    const int R = a.Cin / FK;
    auto gload_row = [&](int i, int r) __attribute__((always_inline)) {
        const unsigned so = (unsigned)r * (FK * 4u);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
#if (WF_SKIP & 1)
            asm volatile("" : "+v"(dd[i][j]) : "s"(so));
#else
            dd[i][j] = buf_load2(rs_in, rowbase[i] + colterm[j], so);
#endif
        }
    };
    auto gload_coef = [&](int r) __attribute__((always_inline)) {
        if constexpr (NORM) cf = *reinterpret_cast<const f32x4*>(coef + (long)r * (FK * 2));
    };
    // column j: the producer's GroupNorm (+ ReLU) on the in-image pixels (the padding stays zero), then t = B^T d in place
    auto col_pass = [&](int j) __attribute__((always_inline)) {
        f32x2 col[6], tc[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            f32x2 v = dd[i][j];
            if constexpr (NORM) {
                v = pk_fma_v(v, f32x2{cf.x, cf.z}, f32x2{cf.y, cf.w});
                v.x = fmaxf(v.x, relu_lo); v.y = fmaxf(v.y, relu_lo);
            }
            col[i] = v;
        }
        if constexpr (NORM) {
            if (border) {                        // a real (wave-uniform) branch: the interior waves pay nothing for the padding
                asm volatile("");
#pragma unroll
                for (int i = 0; i < 6; ++i)
                    if (rowbase[i] == OOBH || colterm[j] == OOBH) col[i] = f32x2{0.f, 0.f};     // the padding stays zero
            }
        }
        bt4(col, tc);
#pragma unroll
        for (int i = 0; i < 6; ++i) dd[i][j] = tc[i];
    };
    // row i: (B^T d) B, stored into round image `vs`
    auto row_pass = [&](int i, float* vs) __attribute__((always_inline)) {
        float* dst = vs + (q >> 2) * SLICE + lt * 8 + 2 * (q & 3);
        f32x2 row[6];
        bt4(dd[i], row);
#pragma unroll
        for (int j = 0; j < 6; ++j) *reinterpret_cast<f32x2*>(dst + (i * 6 + j) * (FT * 8)) = row[j];
    };

    // ---- MFMA role: positions wave * 9 .. + 8, all 32 tiles x 32 channels ----
    // Two accumulator sets, one per 8-channel slice of a round: each is a chain over HALF of K (at most 80 of the 160 input channels
    // this kernel is used up to, option key 27 - float64-anchor ratios 0.64-1.11, profiles/r05_fused_anchor.md),
    // and the two are added once, in the epilogue.  No second-level addition runs inside the loop, where every vector
    // instruction would add its issue time to the MFMAs'.
    constexpr int SPLIT = WF_SPLIT;              // positions (of the wave's 9) with a chain per slice; the rest keep one chain
    f32x16 acc[2][9];            // never zeroed: the first MFMA of every set (round 0) takes SrcC = 0
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int a_off = (wave * 9) * (FT * 8) + (lane & 31) * 8 + (lane >> 5) * 4;     // floats inside a slice image
    const unsigned b_voff = (unsigned)lane * 16u;
    const unsigned b_pos = (unsigned)(wave * 9) * 1024u;
    // B operand: a ring of BR registers sets, the operand of global step g (= 18 r + k) requested BR steps ahead
    constexpr int BR = 6;
    f32x4 bq[BR];
    // operand of step kk (0 .. 17 + BR; 18 and up: the next round) of round r; past the last slice the load is out of range (zeros)
    auto bload = [&](int slot, int r, int kk) __attribute__((always_inline)) {
        const int sl = 2 * r + kk / 9, pi = kk % 9;
#if (WF_SKIP & 2)
        asm volatile("" : "+v"(bq[slot]) : "s"(sl));
#else
        bq[slot] = buf_load4(rs_u, sl < nslice ? b_voff : 0x80000000u, (unsigned)sl * (FP * 1024u) + b_pos + (unsigned)pi * 1024u);
#endif
    };
    // One round (16 input channels) = 18 steps: 2 slices x 9 positions, a step = the 4 dependent MFMAs of one position plus
    // a share of the other work; sched_barrier keeps every share in its step.  At the start of round r the thread's registers
    // hold its patch of round r + 1 after the column pass (t = B^T d):
    //   steps 0-5    row pass i of that patch -> 6 stores into image r + 1, then row i of the patch of round r + 2 is requested
    //   steps 12-17  column pass j of the patch of round r + 2 (requested >= 7 steps earlier)
    auto round = [&](const float* vs, float* vn, const int r, const int kbase, const bool first) __attribute__((always_inline)) {
        const int r2 = r + 2 < R ? r + 2 : R - 1;                  // past the end: the last round again (never multiplied)
        // (the last two rounds redo the last patch for nobody: skipping that work behind block-uniform branches, as the 16 x 64 kernel
        // does, splits the 4-MFMA steps of this kernel into basic blocks and costs 20-27 %: profiles/r09d_wino_fused_diet.md)
        f32x4 av_next = *reinterpret_cast<const f32x4*>(vs + a_off);
#pragma unroll
        for (int k = 0; k < 18; ++k) {
            const int s = k / 9, pi = k % 9;
            const f32x4 av = av_next;
            if (k + 1 < 18) av_next = *reinterpret_cast<const f32x4*>(vs + ((k + 1) / 9) * SLICE + a_off + ((k + 1) % 9) * (FT * 8));
            const f32x4 bv = bq[(kbase + k) % BR];
#if (WF_SKIP & 16)
            acc[pi < SPLIT ? s : 0][pi][0] += av.x * bv.x + av.y * bv.y + av.z * bv.z + av.w * bv.w;
#else
            f32x16& ac = acc[pi < SPLIT ? s : 0][pi];
#if (WF_OPT & 2)
            ac = __builtin_amdgcn_mfma_f32_32x32x2f32(bv.x, av.x, (first && (s == 0 || pi < SPLIT)) ? zero : ac, 0, 0, 0);      // rows = channels, columns = tiles
            ac = __builtin_amdgcn_mfma_f32_32x32x2f32(bv.y, av.y, ac, 0, 0, 0);
            ac = __builtin_amdgcn_mfma_f32_32x32x2f32(bv.z, av.z, ac, 0, 0, 0);
            ac = __builtin_amdgcn_mfma_f32_32x32x2f32(bv.w, av.w, ac, 0, 0, 0);
#else
            ac = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, (first && (s == 0 || pi < SPLIT)) ? zero : ac, 0, 0, 0);
            ac = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, ac, 0, 0, 0);
            ac = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, ac, 0, 0, 0);
            ac = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, ac, 0, 0, 0);
#endif
#endif
            bload((kbase + k) % BR, r, k + BR);
#if !(WF_SKIP & 4)
            if (k < 6) { row_pass(k, vn); gload_row(k, r2); }
            if (k == 6) gload_coef(r2);
            if (k >= 12) col_pass(k - 12);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    float* const st0 = smem;
    float* const st1 = smem + STAGE;
    // prologue: the patches of rounds 0 and 1 are requested together (one exposure to the memory latency, not two; the
    // accumulators' registers are still free to hold the second patch)
    {
        f32x2 d1[6][6];
        const unsigned so1 = (unsigned)(R > 1 ? 1 : 0) * (FK * 4u);
#pragma unroll
        for (int i = 0; i < 6; ++i) gload_row(i, 0);
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) d1[i][j] = buf_load2(rs_in, rowbase[i] + colterm[j], so1);
        gload_coef(0);
#pragma unroll
        for (int g0 = 0; g0 < BR; ++g0) bload(g0, 0, g0);
#pragma unroll
        for (int j = 0; j < 6; ++j) col_pass(j);
#pragma unroll
        for (int i = 0; i < 6; ++i) row_pass(i, st0);
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) dd[i][j] = d1[i][j];
    }
    gload_coef(R > 1 ? 1 : 0);
#pragma unroll
    for (int j = 0; j < 6; ++j) col_pass(j);
    __syncthreads();
    static_assert(36 % BR == 0, "the ring index is static over a pair of rounds");
    WF_STAMP(1);
    round(st0, st1, 0, 0, true);                 // Cin % 32 == 0: rounds come in pairs; the first pair starts the accumulation chains
    __syncthreads();
    WF_STAMP(2);
    round(st1, st0, 1, 18, false);
    __syncthreads();
    WF_STAMP(3);
    for (int r = 2; r < ((WF_SKIP & 64) ? 0 : R); r += 2) {
        round(st0, st1, r, 0, false);
        __syncthreads();
        round(st1, st0, r + 1, 18, false);
        __syncthreads();
    }
    WF_STAMP(6);

    // ---- epilogue: sums -> LDS [position][tile][32 channels] -> A^T . A per (tile, 4 channels) ----
#if (WF_SKIP & 32)
    if (acc[0][0][0] != 12345.f) return;
#endif
    const int h = lane >> 5, rr = lane & 31;
#if (WF_OPT & 2)
    // filters were the A operand: lane (rr, h) holds, per register quad g4, channels 8 g4 + 4 h .. + 3 of tile rr: one 16-byte store
    // each.  The eight 16-byte chunks of a tile's 128-byte row are XOR-swizzled by tile / 2 (rows alternate between the two halves
    // of the 64 banks): 16 lanes of a store, and the 2 tiles x 8 chunk readers below, cover all banks once.
#pragma unroll
    for (int pi = 0; pi < 9; ++pi)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = pi < SPLIT ? acc[0][pi][4 * g4 + e] + acc[1][pi][4 * g4 + e] : acc[0][pi][4 * g4 + e];
            *reinterpret_cast<f32x4*>(smem + ((wave * 9 + pi) * FT + rr) * FC + (((2 * g4 + h) ^ ((rr >> 1) & 7)) << 2)) = v;
        }
#else
#pragma unroll
    for (int pi = 0; pi < 9; ++pi)
#pragma unroll
        for (int e = 0; e < 16; ++e)
            smem[((wave * 9 + pi) * FT + 8 * (e >> 2) + 4 * h + (e & 3)) * FC + rr] = pi < SPLIT ? acc[0][pi][e] + acc[1][pi][e] : acc[0][pi][e];
#endif
    __syncthreads();
    WF_STAMP(7);
    const int c = cc * FC + 4 * q;
#if (WF_OPT & 2)
    const int csw = (q ^ ((lt >> 1) & 7)) << 2;
#else
    const int csw = 4 * q;
#endif
    PV s[4][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        PV col[6], sj[4];
#pragma unroll
        for (int i = 0; i < 6; ++i) col[i] = ldpv(smem + ((i * 6 + j) * FT + lt) * FC + csw);
        at4(col, sj);
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i][j] = sj[i];
    }
    PV sc = {f32x2{1.f, 1.f}, f32x2{1.f, 1.f}}, sh = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
    if (a.scale) {
        sc = ldpv(a.scale + g * a.ss_gs + c);
        sh = ldpv(a.shift + g * a.ss_gs + c);
    }
    const float lo = a.relu ? 0.f : -__builtin_inff();
    const int b0 = (int)(((long)tb * FT) / a.per_img);
    double sa = 0.0, sq = 0.0;
    f32x2 psa = {0.f, 0.f}, psq = {0.f, 0.f};          // WF_OPT & 4: the thread's 64 outputs summed in packed fp32 first
    // the 16 pixels of the tile through a buffer descriptor, as the loader's patch: row + column offsets, and the pixels outside
    // the map (ragged last tiles, short phases, tiles past the end) are dropped by the range check instead of branches
    const __amdgpu_buffer_rsrc_t rs_out = make_rsrc(a.out + (long)g * a.out_gs, a.out_bytes);
    const unsigned opix = (unsigned)a.out_cs * 4u;
    const int oy0 = a.d * 4 * ta.ty + ta.py, ox0 = a.d * 4 * ta.tx + ta.px;
    const unsigned obase = (unsigned)((ta.b * a.H + oy0) * a.W + ox0) * opix + (unsigned)c * 4u;
    unsigned orow[4], ocol[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        orow[i] = (tvalid && oy0 + i * a.d < a.H) ? obase + (unsigned)(i * a.d * a.W) * opix : OOBH;
        ocol[i] = (ox0 + i * a.d < a.W) ? (unsigned)(i * a.d) * opix : OOBH;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        PV row[4];
        at4(s[i], row);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x2 yl = pk_fma_v(row[j].lo, sc.lo, sh.lo), yh = pk_fma_v(row[j].hi, sc.hi, sh.hi);
            const f32x4 y = {fmaxf(yl.x, lo), fmaxf(yl.y, lo), fmaxf(yh.x, lo), fmaxf(yh.y, lo)};
            const unsigned vo = orow[i] + ocol[j];
            if (a.gn_sum) {
                if (vo < OOBH) {
#if (WF_OPT & 4)
                    const f32x2 y01 = {y.x, y.y}, y23 = {y.z, y.w};
                    psa = pk_add(pk_add(psa, y01), y23);
                    psq = pk_fma_v(y23, y23, pk_fma_v(y01, y01, psq));
#else
#pragma unroll
                    for (int e = 0; e < 4; ++e) { sa += (double)y[e]; sq += (double)y[e] * y[e]; }
#endif
                }
            }
            buf_store4(y, rs_out, vo);
        }
    }
    WF_STAMP(8);
    if (a.gn_sum) {
#if (WF_OPT & 4)
        sa = (double)psa.x + (double)psa.y;
        sq = (double)psq.x + (double)psq.y;
#endif
        if (tvalid && (sa != 0.0 || sq != 0.0)) {
            const int grp = c / a.gn_cpg, o = ta.b == b0 ? 0 : 64;
            atomicAdd(&gacc[o + grp * 2], sa);
            atomicAdd(&gacc[o + grp * 2 + 1], sq);
        }
        __syncthreads();
        if (t < 128) {
            const double v = gacc[t];
            const int b = b0 + (t >> 6);
            if (v != 0.0 && b < a.B) atomicAdd(&a.gn_sum[(((long)g * a.Ball + a.boff + b) * a.gn_groups) * 2 + (t & 63)], v);
        }
    }
    WF_STAMP(9);
#ifdef WF_STAMPS
    if (lane == 0 && blockIdx.x < WF_STAMP_BLOCKS && blockIdx.z == 0) {
        g_wf_stamps[((int)blockIdx.x * 4 + wave) * WF_STAMP_N + 14] = __builtin_amdgcn_s_memrealtime();
        g_wf_stamps[((int)blockIdx.x * 4 + wave) * WF_STAMP_N + 13] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492);
    }
#endif
}


// ------------------------------------------------------------------------------------------------------------------------
// The same layer on 16 tiles x 64 output channels per block (64 | Cout), v_mfma_f32_16x16x4_f32, rounds of 32 input channels.
// Against the 32 x 32 form above: a transformed patch value now meets 64 output channels instead of 32 - the input
// transform is recomputed by every block of a tile group, and beside the fp32 MFMA its vector instructions are pure
// overhead (~8 cycles each, nothing co-executes: tools/micro/mfma_f32_shadow.hip) - and the 16 lanes of a tile read the
// whole 128-byte line of a pixel's 32 channels (the 16-channel rounds fetched every line twice).
namespace w64 {
constexpr int FT = 16, FC = 64, FK = 32;
constexpr int SLICE = FP * FT * 16;             // floats of one 16-channel slice image [position][tile][16]
constexpr int STAGE = 2 * SLICE;
static_assert(FP * FT * FC == 2 * STAGE, "the epilogue image overlays the two round images");
constexpr int SMEM_BYTES = 2 * STAGE * 4 + 2 * 32 * 2 * 8;
}  // namespace w64

// element i of [Cout/64][Cin/16][36][4 column blocks][64 lanes][4] <- index into U [36][Cout][Cin]: lane l of the MFMA on
// column block nt supplies output channel nt * 16 + l % 16 and input channels 4 * (l / 16) + e of the 16-channel slice
// which kernel a layer takes: 16 tiles x 64 channels needs 64 | Cout and an even number of 32-channel rounds (an odd count would
// run one round on zero filters; the 32 x 32 kernel's rounds are 16 channels) - or exactly one round (the ONE instance: stem.conv3)
__host__ __device__ inline bool fused_wide(int Cout, int Cin) { return Cout % 64 == 0 && ((Cin / 32) % 2 == 0 || (Cin == 32 && (WF_OPT & 1))); }
__host__ __device__ inline long fused64_src(long i, int Cout, int Cin) {
    const int e = (int)(i & 3), l = (int)((i >> 2) & 63), nt = (int)((i >> 8) & 3);
    long r = i >> 10;
    const int p = (int)(r % 36);
    r /= 36;
    const int ks = (int)(r % (Cin / 16)), cc = (int)(r / (Cin / 16));
    return ((long)p * Cout + cc * 64 + nt * 16 + (l & 15)) * Cin + ks * 16 + 4 * (l >> 4) + e;
}

// ONE: the layer has a single 32-channel round (32 input channels: stem.conv3): one patch, one round, no second image - a template
// instance, because a branch around the second round of the pair costs the register allocation its balance (2.5x slower)
template <bool NORM, bool ONE = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wino_fused64_kernel(const FusedArgs a) {
    constexpr int FT = w64::FT, FC = w64::FC, FK = w64::FK, SLICE = w64::SLICE, STAGE = w64::STAGE;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    double* const gacc = reinterpret_cast<double*>(smem + 2 * STAGE);
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int g = blockIdx.z;
    const int xcd = blockIdx.x & 7, bq_ = a.NB >> 3, br = a.NB & 7;
    const int vb = (xcd < br ? xcd * (bq_ + 1) : br * (bq_ + 1) + (xcd - br) * bq_) + (blockIdx.x >> 3);
    const int tb = vb / a.NC, cc = vb - tb * a.NC;
    if (t < 128) gacc[t] = 0.0;
    WF_STAMP(0);
    WF_STAMP(15);

    // ---- loader role: tile lt of the block, channel pair q of the round's 32 channels ----
    const int lt = t >> 4, q = t & 15;
    const long tile = (long)tb * FT + lt;
    const bool tvalid = tile < a.tiles;
    const TileAt ta = locate32(tvalid ? (unsigned)tile : 0u, (unsigned)a.TH, (unsigned)a.TW, (unsigned)a.d);
    const int y0 = a.d * (4 * ta.ty - 1) + ta.py, x0 = a.d * (4 * ta.tx - 1) + ta.px;
    const unsigned pix = (unsigned)a.in_cs * 4u;
    const unsigned base = (unsigned)((ta.b * a.H + y0) * a.W + x0) * pix + (unsigned)q * 8u;
    unsigned rowbase[6], colterm[6];
    bool edge = false;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const bool ok = tvalid && (unsigned)(y0 + i * a.d) < (unsigned)a.H;
        rowbase[i] = ok ? base + (unsigned)(i * a.d * a.W) * pix : OOBH;
        edge |= !ok;
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const bool ok = (unsigned)(x0 + j * a.d) < (unsigned)a.W;
        colterm[j] = ok ? (unsigned)(j * a.d) * pix : OOBH;
        edge |= !ok;
    }
    // some tile of this wave touches the zero padding (wave-uniform: the interior waves skip the masking of the fused GroupNorm)
    const bool border = NORM && __any(edge);
    const float relu_lo = a.relu_in ? 0.f : -__builtin_inff();      // ReLU of the fused GroupNorm as max(v, lo)
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(a.in + (long)g * a.in_gs, a.in_bytes);
    const int nslice = a.Cin / 16;
    const __amdgpu_buffer_rsrc_t rs_u = make_rsrc(a.uf + (long)g * a.uf_gs + (long)cc * nslice * (FP * 1024), (unsigned)nslice * (FP * 4096u));
    const float* coef = NORM ? a.coef + (long)g * a.coef_gs + ((long)(a.boff + ta.b) * a.Cin + 2 * q) * 2 : nullptr;

    f32x2 dd[6][6];
    f32x4 cf = {1.f, 0.f, 1.f, 0.f};
    const int R = a.Cin / FK;                    // rounds with data; an odd count is followed by one round of zero filters
    auto gload_row = [&](int i, int r) __attribute__((always_inline)) {
        const unsigned so = (unsigned)r * (FK * 4u);
#pragma unroll
        for (int j = 0; j < 6; ++j) dd[i][j] = buf_load2(rs_in, rowbase[i] + colterm[j], so);
    };
    auto gload_coef = [&](int r) __attribute__((always_inline)) {
        if constexpr (NORM) cf = *reinterpret_cast<const f32x4*>(coef + (long)r * (FK * 2));
    };
    auto col_pass = [&](int j) __attribute__((always_inline)) {
        f32x2 col[6], tc[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            f32x2 v = dd[i][j];
            if constexpr (NORM) {
                v = pk_fma_v(v, f32x2{cf.x, cf.z}, f32x2{cf.y, cf.w});
                v.x = fmaxf(v.x, relu_lo); v.y = fmaxf(v.y, relu_lo);
            }
            col[i] = v;
        }
        if constexpr (NORM) {
            if (border) {                        // a real (wave-uniform) branch: the interior waves pay nothing for the padding
                asm volatile("");
#pragma unroll
                for (int i = 0; i < 6; ++i)
                    if (rowbase[i] == OOBH || colterm[j] == OOBH) col[i] = f32x2{0.f, 0.f};     // the padding stays zero
            }
        }
        bt4(col, tc);
#pragma unroll
        for (int i = 0; i < 6; ++i) dd[i][j] = tc[i];
    };
    auto row_pass = [&](int i, float* vs) __attribute__((always_inline)) {
        float* dst = vs + (q >> 3) * SLICE + lt * 16 + 2 * (q & 7);
        f32x2 row[6];
        bt4(dd[i], row);
#pragma unroll
        for (int j = 0; j < 6; ++j) *reinterpret_cast<f32x2*>(dst + (i * 6 + j) * (FT * 16)) = row[j];
    };

    // ---- MFMA role: positions wave * 9 .. + 8; per position 16 tiles x 4 column blocks of 16 channels ----
    // two accumulator sets, one per 16-channel slice of a round (see the 32 x 32 kernel): chains over half of K, added in the epilogue
    constexpr int SPLIT = WF_SPLIT;
    f32x4 acc[2][9][4];          // never zeroed: the first MFMA of every set (round 0) takes SrcC = 0
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const int a_off = (wave * 9) * (FT * 16) + (lane & 15) * 16 + (lane >> 4) * 4;     // floats inside a slice image
    const unsigned b_voff = (unsigned)lane * 16u;
    const unsigned b_pos = (unsigned)(wave * 9) * 4096u;
    constexpr int BR = 3;                        // the B operand (4 column blocks = 4 KB per step) is requested BR steps ahead
    f32x4 bq[BR][4];
    auto bload = [&](int slot, int r, int kk) __attribute__((always_inline)) {
        const int sl = 2 * r + kk / 9, pi = kk % 9;
        const unsigned vo = sl < nslice ? b_voff : 0x80000000u;         // past the last slice: out of range, zeros
        const unsigned so = (unsigned)sl * (FP * 4096u) + b_pos + (unsigned)pi * 4096u;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bq[slot][nt] = buf_load4(rs_u, vo + nt * 1024u, so);
    };
    // One round (32 input channels) = 18 steps: 2 slices x 9 positions, a step = 16 MFMAs (4 k-steps x 4 column blocks);
    // the shares of the other work as in the 32 x 32 kernel
    auto round = [&](const float* vs, float* vn, const int r, const int kbase, const bool first) __attribute__((always_inline)) {
        const int r2 = r + 2 < R ? r + 2 : R - 1;
        // what this round prepares for the rounds after it: the row pass of the next round's patch, and the request + column pass of
        // the patch after that - nothing for rounds that do not exist (WF_OPT & 1: the last two rounds used to redo the last patch for
        // nobody: 147 vector, 36 LDS and 36 memory instructions per round beside an MFMA pipe that shares its lanes with them)
        const bool do_rows = !(WF_OPT & 1) || r + 1 < R, do_full = !(WF_OPT & 1) || r + 2 < R;
        f32x4 av_next = *reinterpret_cast<const f32x4*>(vs + a_off);
#pragma unroll
        for (int k = 0; k < 18; ++k) {
            const int s2 = k / 9, pi = k % 9;
            const f32x4 av = av_next;
            if (k + 1 < 18) av_next = *reinterpret_cast<const f32x4*>(vs + ((k + 1) / 9) * SLICE + a_off + ((k + 1) % 9) * (FT * 16));
            f32x4 bv[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) bv[nt] = bq[(kbase + k) % BR][nt];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    f32x4& ac = acc[pi < SPLIT ? s2 : 0][pi][nt];
                    const f32x4 c0 = (first && e == 0 && (s2 == 0 || pi < SPLIT)) ? zero : ac;
#if (WF_OPT & 2)
                    ac = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[nt][e], av[e], c0, 0, 0, 0);      // rows = channels, columns = tiles
#else
                    ac = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[nt][e], c0, 0, 0, 0);
#endif
                }
            bload((kbase + k) % BR, r, k + BR);
            // (block-uniform branches: the accumulators are not touched inside them)
            if (k < 6 && do_rows) row_pass(k, vn);
            if (do_full) {
                if (k < 6) gload_row(k, r2);
                if (k == 6) gload_coef(r2);
                if (k >= 12) col_pass(k - 12);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    float* const st0 = smem;
    float* const st1 = smem + STAGE;
    // prologue: the patches of rounds 0 and 1 are requested together (one exposure to the memory latency, not two)
    const int r1 = R > 1 ? 1 : 0;
    static_assert(36 % BR == 0, "the ring index is static over a pair of rounds");
    if constexpr (ONE) {
#pragma unroll
        for (int i = 0; i < 6; ++i) gload_row(i, 0);
        gload_coef(0);
#pragma unroll
        for (int g0 = 0; g0 < BR; ++g0) bload(g0, 0, g0);
#pragma unroll
        for (int j = 0; j < 6; ++j) col_pass(j);
#pragma unroll
        for (int i = 0; i < 6; ++i) row_pass(i, st0);
        __syncthreads();
        WF_STAMP(1);
        round(st0, st1, 0, 0, true);             // R == 1: nothing to prepare (do_rows, do_full are false), both slices start their chains
        __syncthreads();
    } else {
    {
        f32x2 d1[6][6];
        const unsigned so1 = (unsigned)r1 * (FK * 4u);
#pragma unroll
        for (int i = 0; i < 6; ++i) gload_row(i, 0);
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) d1[i][j] = buf_load2(rs_in, rowbase[i] + colterm[j], so1);
        gload_coef(0);
#pragma unroll
        for (int g0 = 0; g0 < BR; ++g0) bload(g0, 0, g0);
#pragma unroll
        for (int j = 0; j < 6; ++j) col_pass(j);
#pragma unroll
        for (int i = 0; i < 6; ++i) row_pass(i, st0);
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) dd[i][j] = d1[i][j];
    }
    gload_coef(r1);
#pragma unroll
    for (int j = 0; j < 6; ++j) col_pass(j);
    __syncthreads();
    WF_STAMP(1);
    round(st0, st1, 0, 0, true);                 // the first pair of rounds starts the accumulation chains
    __syncthreads();
    WF_STAMP(2);
    round(st1, st0, 1, 18, false);
    __syncthreads();
    WF_STAMP(3);
    for (int r = 2; r < R; r += 2) {
        round(st0, st1, r, 0, false);
        __syncthreads();
        round(st1, st0, r + 1, 18, false);
        __syncthreads();
    }
    }
    WF_STAMP(6);

    // ---- epilogue: sums -> LDS [position][tile][64 channels] -> A^T . A per (tile, 4 channels) ----
    const int hq = lane >> 4, rr = lane & 15;
#if (WF_OPT & 2)
    // filters were the A operand: lane (rr, hq) holds, per column block nt, channels nt * 16 + 4 hq .. + 3 of tile rr - one 16-byte
    // store per (position, column block).  The sixteen 16-byte chunks of a tile's row are XOR-swizzled by the tile: the 16 lanes of
    // a store (and, below, the 16 chunk readers of a tile) cover all 64 banks once.
#pragma unroll
    for (int pi = 0; pi < 9; ++pi)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const f32x4 v = pi < SPLIT ? acc[0][pi][nt] + acc[1][pi][nt] : acc[0][pi][nt];
            *reinterpret_cast<f32x4*>(smem + ((wave * 9 + pi) * FT + rr) * FC + (((nt * 4 + hq) ^ rr) << 2)) = v;
        }
#else
    // (the 16-channel column blocks of a row are stored XOR-swizzled by row / 4: the four lane groups of an MFMA result hold
    // rows 4 apart, which would otherwise meet in the same banks)
#pragma unroll
    for (int pi = 0; pi < 9; ++pi)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                smem[((wave * 9 + pi) * FT + 4 * hq + e) * FC + ((nt ^ hq) << 4) + rr] = pi < SPLIT ? acc[0][pi][nt][e] + acc[1][pi][nt][e] : acc[0][pi][nt][e];
#endif
    __syncthreads();
    WF_STAMP(7);
    const int c = cc * FC + 4 * q;
#if (WF_OPT & 2)
    const int csw = ((q ^ lt) & 15) << 2;
#else
    const int csw = (((q >> 2) ^ ((lt >> 2) & 3)) << 4) + (q & 3) * 4;
#endif
    PV s[4][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        PV col[6], sj[4];
#pragma unroll
        for (int i = 0; i < 6; ++i) col[i] = ldpv(smem + ((i * 6 + j) * FT + lt) * FC + csw);
        at4(col, sj);
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i][j] = sj[i];
    }
    PV sc = {f32x2{1.f, 1.f}, f32x2{1.f, 1.f}}, sh = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
    if (a.scale) {
        sc = ldpv(a.scale + g * a.ss_gs + c);
        sh = ldpv(a.shift + g * a.ss_gs + c);
    }
    const float lo = a.relu ? 0.f : -__builtin_inff();
    const int b0 = (int)(((long)tb * FT) / a.per_img);
    double sa = 0.0, sq = 0.0;
    f32x2 psa = {0.f, 0.f}, psq = {0.f, 0.f};          // WF_OPT & 4: the thread's 64 outputs summed in packed fp32 first
    // the 16 pixels of the tile through a buffer descriptor, as the loader's patch: row + column offsets, and the pixels outside
    // the map (ragged last tiles, short phases, tiles past the end) are dropped by the range check instead of branches
    const __amdgpu_buffer_rsrc_t rs_out = make_rsrc(a.out + (long)g * a.out_gs, a.out_bytes);
    const unsigned opix = (unsigned)a.out_cs * 4u;
    const int oy0 = a.d * 4 * ta.ty + ta.py, ox0 = a.d * 4 * ta.tx + ta.px;
    const unsigned obase = (unsigned)((ta.b * a.H + oy0) * a.W + ox0) * opix + (unsigned)c * 4u;
    unsigned orow[4], ocol[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        orow[i] = (tvalid && oy0 + i * a.d < a.H) ? obase + (unsigned)(i * a.d * a.W) * opix : OOBH;
        ocol[i] = (ox0 + i * a.d < a.W) ? (unsigned)(i * a.d) * opix : OOBH;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        PV row[4];
        at4(s[i], row);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x2 yl = pk_fma_v(row[j].lo, sc.lo, sh.lo), yh = pk_fma_v(row[j].hi, sc.hi, sh.hi);
            const f32x4 y = {fmaxf(yl.x, lo), fmaxf(yl.y, lo), fmaxf(yh.x, lo), fmaxf(yh.y, lo)};
            const unsigned vo = orow[i] + ocol[j];
            if (a.gn_sum) {
                if (vo < OOBH) {
#if (WF_OPT & 4)
                    const f32x2 y01 = {y.x, y.y}, y23 = {y.z, y.w};
                    psa = pk_add(pk_add(psa, y01), y23);
                    psq = pk_fma_v(y23, y23, pk_fma_v(y01, y01, psq));
#else
#pragma unroll
                    for (int e = 0; e < 4; ++e) { sa += (double)y[e]; sq += (double)y[e] * y[e]; }
#endif
                }
            }
            buf_store4(y, rs_out, vo);
        }
    }
    WF_STAMP(8);
    if (a.gn_sum) {
#if (WF_OPT & 4)
        sa = (double)psa.x + (double)psa.y;
        sq = (double)psq.x + (double)psq.y;
#endif
        if (tvalid && (sa != 0.0 || sq != 0.0)) {
            const int grp = c / a.gn_cpg, o = ta.b == b0 ? 0 : 64;
            atomicAdd(&gacc[o + grp * 2], sa);
            atomicAdd(&gacc[o + grp * 2 + 1], sq);
        }
        __syncthreads();
        if (t < 128) {
            const double v = gacc[t];
            const int b = b0 + (t >> 6);
            if (v != 0.0 && b < a.B) atomicAdd(&a.gn_sum[(((long)g * a.Ball + a.boff + b) * a.gn_groups) * 2 + (t & 63)], v);
        }
    }
    WF_STAMP(9);
#ifdef WF_STAMPS
    if (lane == 0 && blockIdx.x < WF_STAMP_BLOCKS && blockIdx.z == 0) {
        g_wf_stamps[((int)blockIdx.x * 4 + wave) * WF_STAMP_N + 14] = __builtin_amdgcn_s_memrealtime();
        g_wf_stamps[((int)blockIdx.x * 4 + wave) * WF_STAMP_N + 13] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492);
    }
#endif
}

// per-(image, channel) scale and bias of a GroupNorm whose sums are in `stats` (the arithmetic of wino_input_kernel)
__global__ void wino_norm_coef_kernel(const WinoNorm np, int G, int Ball, int Cin, float* __restrict__ coef) {
    const long n = (long)G * Ball * Cin;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cin);
        const long gb = i / Cin;
        const int g = (int)(gb / Ball);
        const double* sb = np.stats + (gb * np.groups + c / np.cpg) * 2;
        const double mean = sb[0] / np.n;
        double var = sb[1] / np.n - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)np.eps));
        const float sc = rstd * np.gamma[g * np.param_gs + c];
        coef[i * 2] = sc;
        coef[i * 2 + 1] = np.beta[g * np.param_gs + c] - (float)mean * sc;
    }
}

// U [36][Cout][Cin] (winograd.hip) -> the per-lane MFMA operand order of this kernel
__global__ void wino_pack_fused_kernel(const float* __restrict__ u, int Cout, int Cin, float* __restrict__ uf) {
    const long n = (long)FP * Cout * Cin;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        uf[i] = u[fused_wide(Cout, Cin) ? fused64_src(i, Cout, Cin) : fused_src(i, Cout, Cin)];
}

}  // namespace

void winograd_fused_pack_host(const float* u, int Cout, int Cin, float* uf) {
    const long n = (long)FP * Cout * Cin;
    for (long i = 0; i < n; ++i) uf[i] = u[fused_wide(Cout, Cin) ? fused64_src(i, Cout, Cin) : fused_src(i, Cout, Cin)];
}

int launch_winograd_fused_pack(const float* u, int Cout, int Cin, float* uf, hipStream_t st) {
    hipLaunchKernelGGL(wino_pack_fused_kernel, dim3(512), dim3(256), 0, st, u, Cout, Cin, uf);
    QB_CHECK(hipGetLastError());
    return 0;
}

// what the single-kernel form covers: F(4x4), fp32 tensors, the exact fp32 and the bf16x3 mode, 32 | Cin <= key 27, 32 | Cout, views below 1 GiB
bool winograd_fused_ok(const WinoP& q, int Ball, int G) {
    const View& in = q.in;
    const View& out = q.out;
    if (!tune().wino_fused || q.m != 4 || (q.dtype != 0 && q.dtype != 3) || !q.uf || in.es != 4 || out.es != 4) return false;
    // bf16x3 mode (fp32-equivalent): this exact fp32 kernel where it beats the pipeline's bf16x3 GEMMs + transforms - not the wide
    // heads that share one input transform between their groups (profiles/r03_final_conv_layers_dtype3.md against r05t_layers.md)
    if (q.dtype == 3 && G > 1 && in.gs == 0 && out.C >= 64) return false;
    if (in.C % 32 || out.C % FC) return false;
    if (in.p == out.p) return false;              // in place: blocks read input halos that other blocks are overwriting
    if (in.C > tune().wino_fused_max_cin) return false;   // the two accumulation chains are Cin / 2 long: at most 80 channels at the default of 160 (profiles/r05_fused_anchor.md)
    const double in_bytes = 4.0 * (((double)Ball * in.H * in.W - 1) * in.cs + in.C);
    const double out_bytes = 4.0 * (((double)Ball * in.H * in.W - 1) * out.cs + out.C);
    if (in_bytes > (double)0x3F000000u || out_bytes > (double)0x3F000000u) return false;
    if ((long)q.dil * in.W * in.cs * 4 * 6 > (1L << 26)) return false;
    return true;
}

size_t winograd_fused_ws_floats(int B, int Cin, int G) { return (size_t)2 * G * B * Cin; }

// The kernels use 145 KB of dynamic LDS: raise their limit on the current device.  Called when a plan is built and by the
// stand-alone op - not from the launcher, which may run inside a hipGraph capture.
int winograd_fused_prepare() {
    QB_CHECK(hipFuncSetAttribute((const void*)wino_fused_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
    QB_CHECK(hipFuncSetAttribute((const void*)wino_fused_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
    QB_CHECK(hipFuncSetAttribute((const void*)wino_fused64_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, w64::SMEM_BYTES));
    QB_CHECK(hipFuncSetAttribute((const void*)wino_fused64_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, w64::SMEM_BYTES));
    QB_CHECK(hipFuncSetAttribute((const void*)wino_fused64_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, w64::SMEM_BYTES));
    QB_CHECK(hipFuncSetAttribute((const void*)wino_fused64_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, w64::SMEM_BYTES));
    return 0;
}

#ifdef WF_STAMPS
int wf_read_stamps(unsigned long long* dst, int n) {
    if (n > WF_STAMP_BLOCKS * 4 * WF_STAMP_N) n = WF_STAMP_BLOCKS * 4 * WF_STAMP_N;
    QB_CHECK(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wf_stamps), sizeof(unsigned long long) * n));
    return 0;
}
#endif

int launch_conv_winograd_fused(const WinoP& q, int Ball, int G, hipStream_t st) {
    const View& in = q.in;
    const View& out = q.out;
    const int H = in.H, W = in.W, d = q.dil;
    FusedArgs a{};
    a.in = in.p; a.in_gs = in.gs; a.in_cs = in.cs; a.H = H; a.W = W; a.Cin = in.C;
    a.in_bytes = (unsigned)(4 * (((long)Ball * H * W - 1) * in.cs + in.C));
    a.uf = q.uf; a.uf_gs = (long)FP * out.C * in.C;
    a.Ball = Ball; a.boff = 0; a.B = Ball;
    a.scale = q.scale; a.shift = q.shift; a.ss_gs = q.ss_gs; a.relu = q.relu;
    a.out = out.p; a.out_cs = out.cs; a.out_gs = out.gs; a.Cout = out.C;
    a.out_bytes = (unsigned)(4 * (((long)Ball * H * W - 1) * out.cs + out.C));
    a.TH = tiles_1d(H, d, 4); a.TW = tiles_1d(W, d, 4); a.d = d;
    a.per_img = wino_tiles(H, W, d, 4);
    a.tiles = (long)Ball * a.per_img;
    const bool wide = fused_wide(out.C, in.C);       // 16 tiles x 64 channels per block (wino_fused64_kernel), else 32 x 32
    const int ft = wide ? w64::FT : FT;
    a.NC = out.C / (wide ? w64::FC : FC);
    const long tblocks = (a.tiles + ft - 1) / ft;
    if (tblocks * a.NC >= (1L << 31) || a.tiles + 64 >= (1L << 31)) return fail("winograd (fused): too many blocks");
    a.NB = (int)(tblocks * a.NC);
    const bool gn_here = q.gn_sum && q.gn_groups > 0 && q.gn_groups <= 32 && (out.C / q.gn_groups) % 4 == 0 && a.per_img >= ft;
    a.gn_sum = gn_here ? q.gn_sum : nullptr; a.gn_groups = q.gn_groups; a.gn_cpg = q.gn_groups ? out.C / q.gn_groups : 1;
    const bool norm = q.norm.stats != nullptr;
    if (norm) {
        WinoNorm np = q.norm;
        if (np.groups <= 0 || in.C % np.groups) return fail("winograd (fused): GroupNorm groups do not divide the channels");
        np.cpg = in.C / np.groups;
        np.n = (double)H * W * np.cpg;
        if (winograd_fused_ws_floats(Ball, in.C, G) > q.ws_floats) return fail("winograd (fused): workspace too small");
        const long n = (long)G * Ball * in.C;
        hipLaunchKernelGGL(wino_norm_coef_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, np, G, Ball, in.C, q.ws);
        QB_CHECK(hipGetLastError());
        a.coef = q.ws; a.coef_gs = (long)Ball * in.C * 2; a.relu_in = np.relu;
    }
    {
        ProfScope prof(q.dtype == 3 ? "conv_gemm_f32pipe" : "wino_fused", 4.0 * G * ((double)Ball * H * W * (in.C + out.C) + (double)FP * in.C * out.C),
                       2.0 * G * FP * (double)a.tiles * in.C * out.C, st);
        const void* fn32 = norm ? (const void*)wino_fused_kernel<true> : (const void*)wino_fused_kernel<false>;
        const void* fn64 = in.C == 32 ? (norm ? (const void*)wino_fused64_kernel<true, true> : (const void*)wino_fused64_kernel<false, true>)
                                      : (norm ? (const void*)wino_fused64_kernel<true> : (const void*)wino_fused64_kernel<false>);
        const void* fn = wide ? fn64 : fn32;
        const int smem_bytes = wide ? w64::SMEM_BYTES : SMEM_BYTES;
        void* args[] = {(void*)&a};
        QB_CHECK(hipLaunchKernel(fn, dim3(a.NB, 1, G), dim3(256), args, smem_bytes, st));
    }
    QB_CHECK(hipGetLastError());
    if (q.gn_sum && !gn_here) return launch_gn_stats(out, Ball, G, q.gn_groups, q.gn_sum, st, false);
    return 0;
}

}  // namespace quber
