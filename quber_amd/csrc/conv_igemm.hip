// Implicit-GEMM convolution on the CDNA4 matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32).
//
// Covers every convolution of the QuBER refiner: 1x1 / 3x3, stride 1-2,
// dilation 1-18, NHWC activations, with the per-channel affine (FrozenBN / folded BN / bias),
// residual add and ReLU of the reference's Conv2d wrapper fused into the epilogue
// (reference: detectron2 Conv2d + BottleneckBlock as used by maskrefiner/modeling/backbone/resnet.py:37-63,
//  441-447 and maskrefiner/modeling/mask_refiner/model.py:386-403, 431-451).
//
// GEMM view: M = B*OH*OW output pixels, N = Cout, K = kh*kw*Cin.  A rows are gathered on the fly
// (im2col never materialised), B rows are the packed weights [Cout][Kpad].
// Block = 256 threads = 4 waves; each wave owns a (TM x TN) grid of 32x32 MFMA tiles.
// K is consumed 32 at a time through LDS ([row][k], pitch 36 floats so that the 16-byte fragment
// reads of a wave hit 64 distinct banks); the next K-slice is prefetched into registers while
// the current one feeds the matrix pipe.
//
// MFMA operand mapping: lane (r = lane&31, h = lane>>5) supplies A[row r][kk = h] and
// B[kk = h][col r].  The order of k inside a K-slice is free as long as A and B agree, so each lane
// fetches 4 consecutive k with one ds_read_b128 and feeds them to 4 successive MFMAs:
// MFMA step s of group ks multiplies k = ks*8 + 4*h + s.
#include "common.h"

namespace quber {

using f32x16 = __attribute__((ext_vector_type(16))) float;
// native vector type for register staging: HIP's float4 is a struct whose copies can lower to memcpy and
// keep the staging arrays in scratch
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int BK = 32;
constexpr int PITCH = 36;
// DT = 1 / 2 (quber_config.compute_dtype: BASELINE.json configs[4] stand-in): the same implicit GEMM with bf16 / fp16
// operands and fp32 accumulation on v_mfma_f32_32x32x16_{bf16,f16} (16x the fp32 matrix rate).  Activations and weights
// stay fp32 in HBM and are rounded to the 16-bit type (round-to-nearest-even) as the K-slice is written to LDS; the LDS
// image is [row][k] with an 80-byte pitch, which makes the 16-byte fragment reads of a wave conflict-free; the
// accumulators, and with them the whole epilogue, are those of the fp32 kernel.
// DT = 3 (compute_dtype 3, "bf16x3"): fp32-equivalent arithmetic on the bf16 matrix pipe.  Every fp32 operand is split as
// it enters LDS into three bf16 terms, x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)
// (round-to-nearest: |x2| <= 2^-9 |x|, |x3| <= 2^-18 |x|, and the three terms carry all 24 significand bits), and a
// product a b is evaluated as the six partial products a1 b1 + a1 b2 + a2 b1 + a2 b2 + a1 b3 + a3 b1 - each exact in
// fp32 - accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The dropped terms (a2 b3, a3 b2, a3 b3) are below 2^-26 |a b|,
// a quarter of an fp32 ulp, so the result differs from the exact-fp32 MFMA chain only by accumulation order.  Six bf16
// MFMAs take 6/16 of the time of the fp32 MFMAs they replace.
template <int DT> struct Half16 { using T = __bf16; };
template <> struct Half16<2> { using T = _Float16; };
template <> struct Half16<4> { using T = _Float16; };     // DT 4: fp16 activations and weights in HBM (the fp16 data path)
constexpr int PITCH_H = 40;   // 16-bit elements per LDS row
constexpr int NPL(int dt) { return dt == 3 ? 3 : 1; }     // bf16 planes per operand

// Main-loop variants that were measured and rejected (profiles/r01c_conv_variants.md): double-buffered LDS with one
// barrier per K-slice, a per-block s_setprio stagger, and 256x128 / 128x256 tiles (8 waves, or 4 waves of 128x64) were
// all 1-25 % slower than this single-buffer, register-prefetch loop at 3 blocks per CU; a 256x64 tile (2 blocks per
// CU) lost to 64x64 on every layer with 33-64 output channels (profiles/r01i_conv_sweep_*.md).
// MODE 3 / 4: the direct / split-K launch of a dilated layer in tap-major K order whose blocks skip the filter rows that
// are zero padding for all their output rows (ASPP d = 18 on a 30-row map: 58 % of the multiplies are with padding).
// MODE 1 (split-K): blockIdx.y owns p.kchunk consecutive K-slices and writes a raw partial tile - for launches that
// cannot fill the chip, or whose tile count is an awkward multiple of the resident slots.
// MODE 2 (split tail): the first p.nfull tiles are computed whole; the remaining tiles - the ragged last round of a
// large launch - are cut into 2^p.tail_shift K-pieces each, appended to the grid, so that the hardware dispatcher hands
// out short pieces while the last whole tiles drain.
// The K range is derived without a division (host-computed kchunk): a 64-bit division here costs hipcc ~50 VGPRs over
// the whole kernel and a wave per SIMD; as written all instantiations allocate the same registers.
#ifndef QB_H16_NBUF
#define QB_H16_NBUF 1      // K-slice images of the fp16 kernel on 128x128 tiles (build-time A/B: one image and three resident
                           // blocks measured 18.2 ms of convolutions at 1024x1024 batch 8, two images and two blocks 18.8)
#endif
#ifndef QB_H16_OCC
#define QB_H16_OCC 3
#endif
// waves per SIMD the kernel is compiled for: two accumulator sets (the MFMA chain and the chunk sums, below) cost the
// 128-row tiles their third resident block; the 64x64 tiles keep five
#ifndef QB_F32_LEAN64_OCC
#define QB_F32_LEAN64_OCC 7  // LEAN loader on 64x64 fp32 tiles (the HBM-bound residual 1x1 layers): 72 registers, no spills (the per-thread
                           // tap loader needs 80 + 6-14 spilled at six waves); seven resident blocks: res2 / res3 / res4 conv3 0.326 / 0.231 /
                           // 0.189 -> 0.315 / 0.220 / 0.183 ms; eight would spill 8.  (128x64 tiles at four blocks instead of three: no change)
#endif
#ifndef QB_H16_LEAN64_OCC
#define QB_H16_LEAN64_OCC 8  // fp16 64x64 tiles with the LEAN loader: 58 registers, eight resident blocks (res3 conv3 0.170 -> 0.160 ms at 1024x1024 x 8)
#endif
#ifndef QB_H16_LEAN_OCC
#define QB_H16_LEAN_OCC 4  // LEAN loader (no 64-bit addresses, no bounds state): the 128x128 fp16 kernel fits 128 registers - four resident blocks
#endif
constexpr int igemm_occupancy(int BM, int BN, int DT = 0, bool LEAN = false) {
    if (DT == 4 && LEAN && BM == 128 && BN == 128) return QB_H16_LEAN_OCC;
    if (DT == 0 && LEAN && BM == 64 && BN == 64) return QB_F32_LEAN64_OCC;
    if (DT == 4 && BM * BN >= 128 * 128) return QB_H16_OCC;
    if (DT == 4 && LEAN && BM == 64 && BN == 64) return QB_H16_LEAN64_OCC;
    if (DT == 4 && BM == 64 && BN == 64) return 7;        // HBM-bound residual layers: blocks in flight are what they live on
    if (BM * BN >= 128 * 128 || (BM == 256 && DT == 3)) return 2;
    return BM == 64 ? (DT == 0 ? 6 : 5) : 3;
}

// LEAN (host: Cin a multiple of the K-slice, tensors below 2 GiB, at most 32 filter taps): every K-slice lies inside one filter
// tap, so the tap position is block-uniform and lives in scalar registers; a thread keeps one 32-bit byte offset and one
// tap-validity bit mask per A row, and the loads are buffer loads whose out-of-range offset (a padding tap) returns zeros -
// 4 vector instructions per A row and K-slice instead of ~20 (64-bit address arithmetic, four bounds compares, four selects
// when the slice is written to LDS).  The fp16 kernel issues 16 MFMAs of 32 cycles per K-slice and the matrix pipe does not
// run beside vector instructions of the same SIMD: the loader's ~95 vector instructions cost it as much time as the MFMAs
// (SQ_INSTS_VALU / SQ_INSTS_MFMA = 8.3, MFMA busy 0.39: profiles/r10b_h16_loader.md).
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

// ConvP::zones: pixel (oy, ox) of GEMM row `rem` of image b.  The rows of an image run through the column zones A = [0, zx1), B = [zx1, zx2),
// C = [zx2, W) as a serpentine: zone A top to bottom, zone B bottom to top, zone C top to bottom, and every odd image backwards - a tile
// that straddles two zones or two images then holds neighbouring map rows on both sides of the seam and needs few filter rows, where the
// plain order (A, B, C top to bottom) gave it the bottom of one zone and the top of the next: all nine taps (52 of 300 tiles at 16
// frames of 30 x 40, against 8 this way)
__device__ __forceinline__ void zone_pixel_of(const ConvP& p, int b, int rem, int& oy, int& ox) {
    if (b & 1) rem = p.ohw - 1 - rem;
    const int nA = p.zx1 * p.OH, nAB = p.zx2 * p.OH;
    const int zb = rem < nA ? 0 : rem < nAB ? nA : nAB;                       // first row of the zone
    const int c0 = rem < nA ? 0 : rem < nAB ? p.zx1 : p.zx2;                   // its first column ...
    const int zw = rem < nA ? p.zx1 : rem < nAB ? p.zx2 - p.zx1 : p.OW - p.zx2;      // ... and width (> 0: the zone holds `rem`)
    oy = (rem - zb) / zw;
    ox = c0 + (rem - zb) - oy * zw;
    if (rem >= nA && rem < nAB) oy = p.OH - 1 - oy;
}
// ... and the tensor row (pixel index) of GEMM row m
__device__ __forceinline__ long zone_row_of(const ConvP& p, long m) {
    const int b = (int)(m / p.ohw), rem = (int)(m - (long)b * p.ohw);
    int oy, ox;
    zone_pixel_of(p, b, rem, oy, ox);
    return (long)b * p.ohw + oy * p.OW + ox;
}

template <int BM, int BN, int WM, int WN, int MODE, int DT = 0, bool LEAN = false>
__global__ __launch_bounds__(WM * WN * 64) __attribute__((amdgpu_waves_per_eu(igemm_occupancy(BM, BN, DT, LEAN)))) void conv_igemm_f32(const ConvP p) {
    constexpr int NTH = WM * WN * 64;
    constexpr int RPP = NTH / 8;      // tile rows covered by one pass of the loader
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int AL = BM / RPP;      // float4 loads per thread per K-slice (A)
    constexpr int BL = BN / RPP;      // (B)
    // two K-slice images (one barrier per slice) for the exact fp32 kernel on 128x128 tiles, as conv_igemm_pk: 2 x 72 KiB per CU
    constexpr int NBUF = ((DT == 0 || (DT == 4 && QB_H16_NBUF == 2)) && BM == 128 && BN == 128) ? 2 : 1;
    // DT 4 (fp16 in HBM): the K-slice image holds the operands' bytes as they are - a row of 32 floats is a row of 64 halfs,
    // so the loader, the image and its pitch are those of the fp32 kernel and a slice carries twice the K
    constexpr bool RAW = DT == 0 || DT == 4;
    static_assert(BM % RPP == 0 && BN % RPP == 0, "loader geometry");
    constexpr int SW = WN * 32;        // columns staged per epilogue pass
    constexpr int SP = SW + 4;         // staging pitch (floats)
    constexpr int KSLICE_FLOATS = !RAW ? NPL(DT) * NBUF * (BM + BN) * PITCH_H / 2 : NBUF * (BM + BN) * PITCH;
    constexpr int SMEM_FLOATS = KSLICE_FLOATS > BM * SP ? KSLICE_FLOATS : BM * SP;
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
    float* const As = smem;
    float* const Bs = smem + NBUF * BM * PITCH;
    using H16 = typename Half16<DT>::T;
    using h16x8 = __attribute__((ext_vector_type(8))) H16;
    using h16x4 = __attribute__((ext_vector_type(4))) H16;
    H16* const Ah = reinterpret_cast<H16*>(smem);                    // DT = 3: planes [3][BM][PITCH_H], then [3][BN][PITCH_H]
    H16* const Bh = Ah + NPL(DT) * NBUF * BM * PITCH_H;

    const int t = threadIdx.x;
    const int g = blockIdx.z;

    // XCD-aware tile order: blocks b and b+8 share an XCD (and its L2); give every XCD one
    // contiguous run of tiles, n-tile fastest, so neighbouring tiles reuse the same input rows.
    int tile, part = 0;                    // part: which K partition this block computes
    bool raw = MODE == 1 || MODE == 4;     // raw partial tile into the workspace instead of the fused epilogue
    {
        const int bid = blockIdx.x, nblk = MODE == 2 ? p.nfull : gridDim.x;
        const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        if constexpr (MODE == 1 || MODE == 4) part = blockIdx.y;
        if constexpr (MODE == 2) {
            if (bid >= nblk) {
                const int u = bid - nblk;
                tile = nblk + (u >> p.tail_shift);
                part = u & ((1 << p.tail_shift) - 1);
                raw = true;
            }
        }
    }
    const int nt = tile % p.ntiles;
    const int mt = tile / p.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;

    int k_begin = 0, nk = p.Kpad / BK;     // first K-slice and number of K-slices of this block
    if (MODE == 1 || (MODE == 2 && raw)) {
        k_begin = part * p.kchunk;
        nk = min(nk - k_begin, p.kchunk);
    }
    // ZON (host: ConvP::zones, 3x3, stride 1, pad = dil): the rows of an image in (column zone, y, x) order, so that a tile inside one zone
    // skips the filter COLUMNS of the padding too (dilation 18 on a 30 x 40 map: per pixel 42 % of the taps meet the image; rows alone
    // leave ~62 % to execute).  The valid K-slices are then (valid filter rows) x (one range of filter columns): kx_lo..kx_hi, and
    // row_skip K-slices are jumped at the end of each filter row.
    constexpr bool ZON = (MODE == 3 || MODE == 4) && LEAN && DT == 0;
    int kx_lo = 0, kx_hi = 0, row_skip = 0, z_nA = 0, z_nAB = 0;
    bool zon = false;
    if constexpr (ZON) {
        zon = p.zones != 0;
        kx_hi = p.kw - 1;
        z_nA = p.zx1 * p.OH;
        z_nAB = p.zx2 * p.OH;
    }
    auto zone_pixel = [&](int b, int rem, int& oy, int& ox) __attribute__((always_inline)) { zone_pixel_of(p, b, rem, oy, ox); };
    if constexpr (MODE == 3 || MODE == 4) {
        // Skip the filter rows that fall in the zero padding for every output row of this tile (dilated layers: with
        // dilation 18 on a 30-row map 58 % of the multiplies are with padding).  Tap-major K order (host: kmode 0,
        // Cin % 32 == 0): the K-slices of filter row ky are contiguous, so the valid rows are one K range.
        const int rem0 = m0 % p.ohw;
        const int last = min(m0 + BM, p.M) - 1;
        const int rem1 = last % p.ohw;
        const bool one_image = m0 / p.ohw == last / p.ohw;
        int oy_min = one_image ? rem0 / p.OW : 0;
        int oy_hi = one_image ? rem1 / p.OW : p.OH - 1;
        if (ZON && zon) {
            // the tile piece by piece - a piece = consecutive rows inside one zone of one image, where the map row moves monotonically -:
            // bounding box of the map rows, union of the zones' filter columns (A = {1, 2}, B = {0, 1, 2} or {1}, C = {0, 1})
            oy_min = p.OH - 1; oy_hi = 0;
            kx_lo = 2; kx_hi = 0;
            for (int m = m0; m <= last;) {
                const int b = m / p.ohw, rem = m - b * p.ohw;
                const int rf = (b & 1) ? p.ohw - 1 - rem : rem;                     // forward position inside the image
                const int z = rf < z_nA ? 0 : rf < z_nAB ? 1 : 2;
                const int zs = z == 0 ? 0 : z == 1 ? z_nA : z_nAB, ze = z == 0 ? z_nA : z == 1 ? z_nAB : p.ohw;      // the zone's rows [zs, ze)
                // rows of the image left in this zone, walking forwards (even image) or backwards (odd image)
                const int left = (b & 1) ? rf - zs + 1 : ze - rf;
                const int len = min(left, last - m + 1);
                int ya, yb, x;
                zone_pixel(b, rem, ya, x);
                zone_pixel(b, rem + len - 1, yb, x);
                oy_min = min(oy_min, min(ya, yb)); oy_hi = max(oy_hi, max(ya, yb));
                const int lo = z == 0 ? 1 : z == 2 ? 0 : (p.zones == 1 ? 0 : 1);
                const int hi = z == 0 ? 2 : z == 2 ? 1 : (p.zones == 1 ? 2 : 1);
                kx_lo = min(kx_lo, lo); kx_hi = max(kx_hi, hi);
                m += len;
            }
        }
        int ky_lo = 0, ky_hi = p.kh - 1;
        while (ky_lo < ky_hi && oy_hi * p.stride - p.pad + ky_lo * p.dil < 0) ++ky_lo;
        while (ky_hi > ky_lo && oy_min * p.stride - p.pad + ky_hi * p.dil >= p.H) --ky_hi;
        const int cpt = p.Cin / BK;               // K-slices per filter tap
        const int per_row = p.kw * cpt;
        k_begin = ky_lo * per_row;
        nk = (ky_hi - ky_lo + 1) * per_row;
        if (ZON && zon) {
            k_begin += kx_lo * cpt;
            nk = (ky_hi - ky_lo + 1) * (kx_hi - kx_lo + 1) * cpt;
            row_skip = (p.kw - (kx_hi - kx_lo + 1)) * cpt;
        }
        if constexpr (MODE == 4) {               // the valid range again in gridDim.y partitions (a trailing one may be empty)
            const int chunk = (nk + (int)gridDim.y - 1) / (int)gridDim.y;
            const int done = min(part * chunk, nk);
            if (ZON && zon) {                    // slice `done` of the valid ones: filter row, column, channel slice -> its K-slice
                const int rowlen = (kx_hi - kx_lo + 1) * cpt;
                const int r = done / rowlen, w = done - r * rowlen;
                k_begin += r * per_row + w;      // (w < rowlen: inside the row's column range, which starts at kx_lo - already in k_begin)
            } else {
                k_begin += done;
            }
            nk = min(nk - done, chunk);
            if (nk == 0) k_begin = ky_lo * per_row;   // keep the (unused) prologue loads inside the weight rows
        }
    }
    const float* __restrict__ in = p.in + (long)g * p.in_gs;
    const float* __restrict__ wt = p.w + (long)g * p.w_gs + (long)k_begin * BK;

    // ---- loader state ----
    // Each thread owns one 16-byte column (4 consecutive k) of AL A-rows and BL B-rows.  Its position inside the
    // filter window (ky, kx, channel c) advances by BK per K-slice and is tracked incrementally (no divisions in
    // the loop).  Loads are unconditional: an out-of-image tap reads a valid dummy address and is zeroed when
    // the slice is written to LDS, so the compiler can issue all loads of a slice back to back.
    const int kq = (t & 7) * 4;
    const int lrow = t >> 3;
    int iy0[AL], ix0[AL];
    const float* rowp[AL];
#pragma unroll
    for (int i = 0; i < AL; ++i) {
        const int m = m0 + lrow + RPP * i;
        if (m < p.M && p.kh == 1 && p.stride == 1 && p.pad == 0) {
            // 1x1, stride 1: output pixel m reads input pixel m - no divisions (the short-K residual layers are all prologue)
            iy0[i] = 0;
            ix0[i] = 0;
            rowp[i] = in + (long)m * p.in_cs;
        } else if (m < p.M) {
            const int ohw = p.OH * p.OW;
            const int b = m / ohw;
            const int rem = m - b * ohw;
            const int oy = rem / p.OW;
            const int ox = rem - oy * p.OW;
            iy0[i] = oy * p.stride - p.pad;
            ix0[i] = ox * p.stride - p.pad;
            rowp[i] = in + ((long)b * p.H * p.W + (long)iy0[i] * p.W + ix0[i]) * p.in_cs;
        } else {
            iy0[i] = -(1 << 28);
            ix0[i] = 0;
            rowp[i] = in;
        }
    }
    const float* wrow[BL];
#pragma unroll
    for (int i = 0; i < BL; ++i) {
        const int n = n0 + lrow + RPP * i;
        wrow[i] = wt + (long)(n < p.Cout ? n : 0) * p.Kpad + kq;   // columns >= Cout are never stored
    }
    int kc, kx, ky;
    if (p.kmode) {
        kc = kq; kx = 0; ky = 0;
        if (MODE != 0) {
            const int taps = p.kh * p.kw;
            const int cb = k_begin / taps, tap = k_begin - cb * taps;
            kc = cb * BK + kq;
            ky = tap / p.kw;
            kx = tap - ky * p.kw;
        }
    } else {
        const int k = k_begin * BK + kq;
        const int tap = k / p.Cin;
        kc = k - tap * p.Cin;
        ky = tap / p.kw;
        kx = tap - ky * p.kw;
    }

    // LEAN loader state: byte offset of (pixel of row i at tap (0, 0), channel kq) and the taps of row i that fall inside the image
    int aoff[AL], boff[BL];
    unsigned amask[AL];
    int skc = 0, skx = 0, sky = 0;        // block-uniform tap position of the next K-slice
    int swk = 0;                          // ZON: its K-slice index relative to k_begin
    __amdgpu_buffer_rsrc_t rsa, rsb;
    if constexpr (LEAN) {
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            const int m = m0 + lrow + RPP * i;
            aoff[i] = 0;
            amask[i] = 0;
            if (m < p.M && p.kh == 1 && p.stride == 1 && p.pad == 0) {
                aoff[i] = (m * p.in_cs + kq) * 4;
                amask[i] = 1;
            } else if (m < p.M) {
                const int ohw = p.OH * p.OW;
                const int b = m / ohw;
                const int rem = m - b * ohw;
                int oy = rem / p.OW;
                int ox = rem - oy * p.OW;
                if (ZON && zon) zone_pixel(b, rem, oy, ox);
                const int y0 = oy * p.stride - p.pad, x0 = ox * p.stride - p.pad;
                aoff[i] = (((b * p.H + y0) * p.W + x0) * p.in_cs + kq) * 4;
                // a tap is inside the image when its row and its column are: kh + kw tests and kh shifts instead of kh x kw double
                // tests (this prologue runs per tile: on the short-K 3x3 layers it is a tenth of the tile's time)
                unsigned xbits = 0, mk = 0;
                for (int tx = 0; tx < p.kw; ++tx)
                    if ((unsigned)(x0 + tx * p.dil) < (unsigned)p.W) xbits |= 1u << tx;
                for (int ty = 0; ty < p.kh; ++ty)
                    if ((unsigned)(y0 + ty * p.dil) < (unsigned)p.H) mk |= xbits << (ty * p.kw);
                amask[i] = mk;
            }
        }
#pragma unroll
        for (int i = 0; i < BL; ++i) {
            const int n = n0 + lrow + RPP * i;
            boff[i] = ((n < p.Cout ? n : 0) * p.Kpad + kq) * 4;
        }
        {
            const int kb = __builtin_amdgcn_readfirstlane(k_begin);
            if (p.kmode) {
                const int taps = p.kh * p.kw;
                const int cb = kb / taps, tap = kb - cb * taps;
                skc = cb * BK;
                sky = tap / p.kw;
                skx = tap - sky * p.kw;
            } else {
                const int k = kb * BK;
                const int tap = k / p.Cin;
                skc = k - tap * p.Cin;
                sky = tap / p.kw;
                skx = tap - sky * p.kw;
            }
            skc = __builtin_amdgcn_readfirstlane(skc); skx = __builtin_amdgcn_readfirstlane(skx); sky = __builtin_amdgcn_readfirstlane(sky);
        }
        rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, p.lean_in_bytes, 0x00020000);
        rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wt), 0, 0x7ffffff0, 0x00020000);
    }

    f32x4 ra[AL], rb[BL];
    bool aok[AL];
    auto gload = [&](int kt) __attribute__((always_inline)) {
        if constexpr (LEAN) {
            const int tap = sky * p.kw + skx;
            const int soff = (((sky * p.dil) * p.W + skx * p.dil) * p.in_cs + skc) * 4;
#pragma unroll
            for (int i = 0; i < AL; ++i) {
                const bool ok = (amask[i] >> tap) & 1u;
                aok[i] = true;
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsa, ok ? aoff[i] + soff : (int)0x80000000, 0, 0);
                ra[i] = __builtin_bit_cast(f32x4, v);
            }
            const int wso = (ZON ? swk : kt) * BK * 4;           // ZON: the K-slices of a tile are not consecutive (row_skip)
#pragma unroll
            for (int i = 0; i < BL; ++i) rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsb, boff[i], wso, 0));
            if (p.kmode) {
                if (++skx == p.kw) {
                    skx = 0;
                    if (++sky == p.kh) { sky = 0; skc += BK; }
                }
            } else {
                skc += BK;
                if constexpr (ZON) ++swk;
                if (skc >= p.Cin) {
                    skc = 0;
                    if constexpr (ZON) {
                        if (++skx > kx_hi) { skx = kx_lo; ++sky; swk += row_skip; }
                    } else {
                        if (++skx == p.kw) { skx = 0; ++sky; }
                    }
                }
            }
            return;
        }
        const bool kok = p.kmode || ky < p.kh;
        const int dy = ky * p.dil, dx = kx * p.dil;
        const long off = ((long)dy * p.W + dx) * p.in_cs + kc;
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            const bool ok = kok && (unsigned)(iy0[i] + dy) < (unsigned)p.H && (unsigned)(ix0[i] + dx) < (unsigned)p.W;
            aok[i] = ok;
            ra[i] = *reinterpret_cast<const f32x4*>(ok ? rowp[i] + off : in);
        }
#pragma unroll
        for (int i = 0; i < BL; ++i) rb[i] = *reinterpret_cast<const f32x4*>(wrow[i] + kt * BK);
        // advance to the next K-slice
        if (p.kmode) {
            // slice-major K order (k = (c/32, tap, c%32)): the taps of one 32-channel slice are consecutive K-slices,
            // so the 9 shifted reads of a 3x3 window hit the same cache lines back to back instead of 8+ slices apart
            if (++kx == p.kw) {
                kx = 0;
                if (++ky == p.kh) { ky = 0; kc += BK; }
            }
        } else {
            kc += BK;
#pragma unroll
            for (int it = 0; it < BK / 8; ++it) {      // Cin >= 8: at most BK/8 filter taps per K-slice
                if (kc >= p.Cin) {
                    kc -= p.Cin;
                    if (++kx == p.kw) { kx = 0; ++ky; }
                }
            }
        }
    };
    auto lstore = [&](int buf) __attribute__((always_inline)) {
        if constexpr (DT == 3) {
            auto split = [&](const f32x4 v, H16* dst, int plane_stride) __attribute__((always_inline)) {
                h16x4 p1, p2, p3;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = v[e];
                    const H16 x1 = (H16)x;
                    const float r1 = x - (float)x1;
                    const H16 x2 = (H16)r1;
                    const float r2 = r1 - (float)x2;
                    p1[e] = x1; p2[e] = x2; p3[e] = (H16)r2;
                }
                *reinterpret_cast<h16x4*>(dst) = p1;
                *reinterpret_cast<h16x4*>(dst + plane_stride) = p2;
                *reinterpret_cast<h16x4*>(dst + 2 * plane_stride) = p3;
            };
#pragma unroll
            for (int i = 0; i < AL; ++i)
                split((LEAN || aok[i]) ? ra[i] : f32x4{0.f, 0.f, 0.f, 0.f}, &Ah[(lrow + RPP * i) * PITCH_H + kq], BM * PITCH_H);
#pragma unroll
            for (int i = 0; i < BL; ++i) split(rb[i], &Bh[(lrow + RPP * i) * PITCH_H + kq], BN * PITCH_H);
            return;
        }
        if constexpr (!RAW) {
#pragma unroll
            for (int i = 0; i < AL; ++i) {
                const f32x4 v = (LEAN || aok[i]) ? ra[i] : f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<h16x4*>(&Ah[buf * BM * PITCH_H + (lrow + RPP * i) * PITCH_H + kq]) =
                    h16x4{(H16)v.x, (H16)v.y, (H16)v.z, (H16)v.w};
            }
#pragma unroll
            for (int i = 0; i < BL; ++i)
                *reinterpret_cast<h16x4*>(&Bh[buf * BN * PITCH_H + (lrow + RPP * i) * PITCH_H + kq]) =
                    h16x4{(H16)rb[i].x, (H16)rb[i].y, (H16)rb[i].z, (H16)rb[i].w};
            return;
        }
#pragma unroll
        for (int i = 0; i < AL; ++i)
            *reinterpret_cast<f32x4*>(&As[buf * BM * PITCH + (lrow + RPP * i) * PITCH + kq]) =
                (LEAN || aok[i]) ? ra[i] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < BL; ++i)
            *reinterpret_cast<f32x4*>(&Bs[buf * BN * PITCH + (lrow + RPP * i) * PITCH + kq]) = rb[i];
    };

    const int wave = t >> 6, lane = t & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // 16-bit operands: MFMA step ks of a slice multiplies k = 16 ks + 8 h + (0..7): one 16-byte fragment read per operand tile
    auto mma_h = [&](int buf, int ks) __attribute__((always_inline)) {
        const H16* ap = &Ah[buf * BM * PITCH_H + (wm * TM * 32 + r) * PITCH_H + 8 * h + ks * 16];
        const H16* bp = &Bh[buf * BN * PITCH_H + (wn * 32 + r) * PITCH_H + 8 * h + ks * 16];
        h16x8 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const h16x8*>(ap + i * 32 * PITCH_H);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const h16x8*>(bp + j * WN * 32 * PITCH_H);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (DT == 2) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
                else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
    };
    // fp16 in HBM (DT 4): the image rows are PITCH floats = 2 * PITCH halfs; MFMA step ks multiplies k = 16 ks + 8 h + (0..7) of the
    // slice's 64 halfs: one 16-byte fragment read per operand tile, at the byte offsets the fp32 kernel reads its f32x4 from
    auto mma_raw16 = [&](int buf, int ks) __attribute__((always_inline)) {
        const H16* ap = reinterpret_cast<const H16*>(&As[buf * BM * PITCH + (wm * TM * 32 + r) * PITCH]) + 8 * h + ks * 16;
        const H16* bp = reinterpret_cast<const H16*>(&Bs[buf * BN * PITCH + (wn * 32 + r) * PITCH]) + 8 * h + ks * 16;
        h16x8 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const h16x8*>(ap + i * 32 * 2 * PITCH);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const h16x8*>(bp + j * WN * 32 * 2 * PITCH);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    };
    // bf16x3: three fragments per operand tile, six MFMAs per output tile and k-step, smallest partial products first
    auto mma_x3 = [&](int ks) __attribute__((always_inline)) {
        const H16* ap = &Ah[(wm * TM * 32 + r) * PITCH_H + 8 * h + ks * 16];
        const H16* bp = &Bh[(wn * 32 + r) * PITCH_H + 8 * h + ks * 16];
        h16x8 a[3][TM], b[3][TN];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[q][i] = *reinterpret_cast<const h16x8*>(ap + q * BM * PITCH_H + i * 32 * PITCH_H);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[q][j] = *reinterpret_cast<const h16x8*>(bp + q * BN * PITCH_H + j * WN * 32 * PITCH_H);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                f32x16 c = acc[i][j];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2][j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], c, 0, 0, 0);
                acc[i][j] = c;
            }
    };
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto mma = [&](int buf, int ks) __attribute__((always_inline)) {
        const float* ap = &As[buf * BM * PITCH + (wm * TM * 32 + r) * PITCH + 4 * h];
        const float* bp = &Bs[buf * BN * PITCH + (wn * 32 + r) * PITCH + 4 * h];
        f32x4 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(ap + i * 32 * PITCH + ks * 8);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(bp + j * WN * 32 * PITCH + ks * 8);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                // exact fp32: the chain of a K-slice starts from zero (SrcC = 0) and is added to `top` when the slice is done
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, ks == 0 ? zero16 : acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
            }
    };
    // Two-level accumulation: the chain of fp32 additions behind an output is cut into chunks - one K-slice (32 k) in the
    // exact fp32 mode, p.acc_chunk slices in the bf16x3 mode - whose sums are added in a second register set.  The rounding
    // error of a sum grows with the length of its chain: one chain over K = 128 ... 4608 left the network 2-4x further from
    // a float64 evaluation than the CPU reference's blocked reduction is; chunked, the two are level, Winograd layers
    // included (tests/fp64_anchor.py, profiles/r03f_anchor_chunk.md).
    constexpr bool TWO = DT == 0 || DT == 3;       // the 16-bit operand modes keep one chain (their tolerance is 100x wider)
    f32x16 top[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) top[i][j][e] = 0.f;
    int fold_in = p.acc_chunk;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) gload(kt + 1);
        if constexpr (DT == 3) {
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) mma_x3(ks);
        } else if constexpr (DT == 4) {
#pragma unroll
            for (int ks = 0; ks < BK / 8; ++ks) mma_raw16(NBUF == 2 ? (kt & 1) : 0, ks);
        } else if constexpr (DT != 0) {
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) mma_h(0, ks);
        } else {
#pragma unroll
            for (int ks = 0; ks < BK / 8; ++ks) mma(NBUF == 2 ? (kt & 1) : 0, ks);
        }
        if constexpr (DT == 0) {                   // every slice is a chunk
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) top[i][j] += acc[i][j];
        } else if (TWO && (kt + 1 == nk || --fold_in == 0)) {           // bf16x3: chunks of p.acc_chunk slices
            fold_in = p.acc_chunk;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    top[i][j] += acc[i][j];
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
                }
        }
        if constexpr (NBUF == 2) {
            if (kt + 1 < nk) lstore((kt + 1) & 1);      // the image nobody reads: its readers passed the barrier of slice kt - 1
            __syncthreads();
        } else {
            __syncthreads();
            if (kt + 1 < nk) {
                lstore(0);
                __syncthreads();
            }
        }
    }

    // ---- epilogue: y = acc*scale + shift (+ residual) (ReLU) ----
    // The accumulators are transposed through LDS one 32-column tile per wave at a time, so that global
    // stores (and residual loads) are 16 bytes per lane over WN*32 consecutive channels of a pixel.
    // raw: partial tile into slab (partition s, group g) = ws[(s*G + g) * ws_rows * Cout ...], rows counted from
    // ws_row0; the affine, residual and ReLU are applied by splitk_reduce_kernel after the slabs have been summed
    // (DT 4: the tensors are fp16 - out / res are element pointers of 2 bytes, the split-K workspace stays fp32)
    constexpr bool HOUT = DT == 4;
    float* __restrict__ out = raw ? p.ws + (((long)part * gridDim.z + g) * p.ws_rows - p.ws_row0) * (long)p.Cout
                                  : HOUT ? reinterpret_cast<float*>(reinterpret_cast<H16*>(p.out) + (long)g * p.out_gs)
                                         : p.out + (long)g * p.out_gs;
    const int out_cs = raw ? p.Cout : p.out_cs;
    const float* __restrict__ res = (p.res && !raw) ? (HOUT ? reinterpret_cast<const float*>(reinterpret_cast<const H16*>(p.res) + (long)g * p.res_gs)
                                                             : p.res + (long)g * p.res_gs)
                                                    : nullptr;
    const bool hstore = HOUT && !raw;
    const float* __restrict__ scale = (p.scale && !raw) ? p.scale + g * p.ss_gs : nullptr;
    const float* __restrict__ shift = (p.shift && !raw) ? p.shift + g * p.ss_gs : nullptr;
    const bool relu = p.relu && !raw;
    const float* __restrict__ prelu = (p.prelu && !raw) ? p.prelu + g * p.ss_gs : nullptr;   // per-channel PReLU slopes
    // GroupNorm sums of the stored values (the layer's consumer is a GroupNorm): fp64 sum and sum of squares per
    // (image, norm group), gathered per block in LDS.  A tile spans at most two images (host: OH*OW >= BM).
    __shared__ double gacc[2 * 32 * 2];        // [image b0 / b0+1][group][sum, sum of squares]
    const bool gn = p.gn_sum != nullptr && !raw;
    int b0 = 0, m_next = 0;
    if (gn) {
        if (t < 128) gacc[t] = 0.0;
        b0 = m0 / p.ohw;
        m_next = (b0 + 1) * p.ohw;
    }
    __syncthreads();
    // The 16-byte path of the fp32 tensors: thread t owns channel column 4 * (t % CPR) of the pass and the rows t / CPR + i * RSTEP, i < NCH.
    // Everything it reads from memory - the affine parameters, the PReLU slopes, the residual of its first GRP rows - is requested BEFORE the
    // accumulators go through LDS, and the residual of the next GRP rows before the stores of the rows in hand (vmcnt counts loads and
    // stores in issue order).  Written as a loop of load-use-store per chunk, the compiler waits for every load where it is issued: eight
    // memory latencies in a row per 64 x 64 tile, 12 of the ~21 us a block of the K = 64 residual layers lives (profiles/r17_epilogue.md).
    constexpr int CPR = SW / 4;         // float4 chunks per row
    static_assert(NTH % CPR == 0, "a thread keeps its channel column across the rows of a pass");
    constexpr int RSTEP = NTH / CPR, NCH = BM / RSTEP, GRP = NCH < 4 ? NCH : 4;
    static_assert(BM % RSTEP == 0 && NCH % GRP == 0, "whole groups of chunks per thread");
    const bool vec4 = !HOUT && p.vec_out;     // block-uniform (the fp16 kernels are compiled for 64 / 128 registers: their loops stay as they were)
    const int eq = (t % CPR) * 4, erow0 = t / CPR;
    // tensor row (pixel index) of GEMM row m: m itself, or - ZON - the pixel the zone order puts there
    auto pixel_row = [&](int m) __attribute__((always_inline)) -> long {
        if constexpr (ZON) {
            if (zon && !raw) return zone_row_of(p, m);      // (a partial tile goes to the workspace in GEMM row order: the reduce kernel maps it)
        }
        return m;
    };
    auto res_load = [&](int i, int n) __attribute__((always_inline)) -> float4 {
        const int m = min(m0 + erow0 + i * RSTEP, p.M - 1);       // rows past the end: a valid address, the value is never used
        return *reinterpret_cast<const float4*>(res + pixel_row(m) * p.res_cs + n);
    };
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nb = n0 + j * SW;    // first channel of this pass
        const int en = nb + eq;
        const bool nok = vec4 && en < p.Cout;
        float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = make_float4(0.f, 0.f, 0.f, 0.f), sl4 = sh4;
        float4 rv[GRP];
#pragma unroll
        for (int i = 0; i < GRP; ++i) rv[i] = sh4;
        auto request = [&]() __attribute__((always_inline)) {
            if (nok) {
                if (scale) {
                    sc4 = *reinterpret_cast<const float4*>(scale + en);
                    sh4 = *reinterpret_cast<const float4*>(shift + en);
                }
                if (prelu) sl4 = *reinterpret_cast<const float4*>(prelu + en);
                if (res) {
#pragma unroll
                    for (int i = 0; i < GRP; ++i) rv[i] = res_load(i, en);
                }
            }
        };
        if constexpr (!HOUT) request();
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                smem[row * SP + wn * 32 + r] = TWO ? top[i][j][e] : acc[i][j][e];
            }
        __syncthreads();
        if (HOUT && hstore && p.vec_out == 2) {
            // fp16 tensors: 8 channels = 16 bytes per lane (stores, residual loads); the two float4 halves may sit in different norm groups
            constexpr int CPR8 = SW / 8;
            static_assert(NTH % CPR8 == 0, "a thread keeps its channel column across the rows of a pass");
            double sa0 = 0.0, qa0 = 0.0, sa1 = 0.0, qa1 = 0.0, sb0 = 0.0, qb0 = 0.0, sb1 = 0.0, qb1 = 0.0;
#pragma unroll
            for (int c = t; c < BM * CPR8; c += NTH) {
                const int row = c / CPR8, q = (c - row * CPR8) * 8;
                const int m = m0 + row, n = nb + q;
                if (m < p.M && n < p.Cout) {
                    float v[8];
                    *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(&smem[row * SP + q]);
                    *reinterpret_cast<float4*>(v + 4) = *reinterpret_cast<const float4*>(&smem[row * SP + q + 4]);
                    if (scale) {
                        float sc[8], sh[8];
                        *reinterpret_cast<float4*>(sc) = *reinterpret_cast<const float4*>(scale + n);
                        *reinterpret_cast<float4*>(sc + 4) = *reinterpret_cast<const float4*>(scale + n + 4);
                        *reinterpret_cast<float4*>(sh) = *reinterpret_cast<const float4*>(shift + n);
                        *reinterpret_cast<float4*>(sh + 4) = *reinterpret_cast<const float4*>(shift + n + 4);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], sc[e], sh[e]);
                    }
                    if (res) {
                        const h16x8 rh = *reinterpret_cast<const h16x8*>(reinterpret_cast<const H16*>(res) + (long)m * p.res_cs + n);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += (float)rh[e];
                    }
                    h16x8 hv;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if (relu) v[e] = fmaxf(v[e], 0.f);
                        hv[e] = (H16)v[e];
                        v[e] = (float)hv[e];
                    }
                    *reinterpret_cast<h16x8*>(reinterpret_cast<H16*>(out) + (long)m * out_cs + n) = hv;
                    if (gn) {
                        const double a = (double)v[0] + (double)v[1] + (double)v[2] + (double)v[3];
                        const double b = (double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2] + (double)v[3] * v[3];
                        const double a2 = (double)v[4] + (double)v[5] + (double)v[6] + (double)v[7];
                        const double b2 = (double)v[4] * v[4] + (double)v[5] * v[5] + (double)v[6] * v[6] + (double)v[7] * v[7];
                        if (m < m_next) { sa0 += a; qa0 += b; sb0 += a2; qb0 += b2; } else { sa1 += a; qa1 += b; sb1 += a2; qb1 += b2; }
                    }
                }
            }
            if (gn) {
                const int n = nb + (t % CPR8) * 8;
                if (n < p.Cout) {
                    const int g0 = n / p.gn_cpg, g1 = (n + 4) / p.gn_cpg;
                    atomicAdd(&gacc[g0 * 2], sa0); atomicAdd(&gacc[g0 * 2 + 1], qa0);
                    atomicAdd(&gacc[g1 * 2], sb0); atomicAdd(&gacc[g1 * 2 + 1], qb0);
                    if (sa1 != 0.0 || qa1 != 0.0 || sb1 != 0.0 || qb1 != 0.0) {
                        atomicAdd(&gacc[64 + g0 * 2], sa1); atomicAdd(&gacc[64 + g0 * 2 + 1], qa1);
                        atomicAdd(&gacc[64 + g1 * 2], sb1); atomicAdd(&gacc[64 + g1 * 2 + 1], qb1);
                    }
                }
            }
        } else if (p.vec_out) {
            double s0 = 0.0, q0 = 0.0, s1 = 0.0, q1 = 0.0;
            if constexpr (HOUT) {
#pragma unroll
                for (int c = t; c < BM * CPR; c += NTH) {
                    const int row = c / CPR, q = (c - row * CPR) * 4;
                    const int m = m0 + row, n = nb + q;
                    if (m < p.M && n < p.Cout) {
                        float4 v = *reinterpret_cast<const float4*>(&smem[row * SP + q]);
                        if (scale) {
                            const float4 sc = *reinterpret_cast<const float4*>(scale + n);
                            const float4 sh = *reinterpret_cast<const float4*>(shift + n);
                            v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y);
                            v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
                        }
                        if (res) {
                            float4 rv;
                            if constexpr (HOUT) {
                                const h16x4 rh = *reinterpret_cast<const h16x4*>(reinterpret_cast<const H16*>(res) + (long)m * p.res_cs + n);
                                rv = make_float4((float)rh.x, (float)rh.y, (float)rh.z, (float)rh.w);
                            } else {
                                rv = *reinterpret_cast<const float4*>(res + (long)m * p.res_cs + n);
                            }
                            v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                        }
                        if (relu) {
                            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                        }
                        if (prelu) {
                            const float4 sl = *reinterpret_cast<const float4*>(prelu + n);
                            v.x = v.x > 0.f ? v.x : v.x * sl.x; v.y = v.y > 0.f ? v.y : v.y * sl.y;
                            v.z = v.z > 0.f ? v.z : v.z * sl.z; v.w = v.w > 0.f ? v.w : v.w * sl.w;
                        }
                        if (hstore) {           // rounded once to fp16; the GroupNorm sums are of the stored values
                            const h16x4 hv = {(H16)v.x, (H16)v.y, (H16)v.z, (H16)v.w};
                            *reinterpret_cast<h16x4*>(reinterpret_cast<H16*>(out) + (long)m * out_cs + n) = hv;
                            v = make_float4((float)hv.x, (float)hv.y, (float)hv.z, (float)hv.w);
                        } else {
                            *reinterpret_cast<float4*>(out + (long)m * out_cs + n) = v;
                        }
                        if (gn) {
                            const double a = (double)v.x + (double)v.y + (double)v.z + (double)v.w;
                            const double b = (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
                            if (m < m_next) { s0 += a; q0 += b; } else { s1 += a; q1 += b; }
                        }
                    }
                }
            } else {
                if (nok) {
#pragma unroll
                    for (int i0 = 0; i0 < NCH; i0 += GRP) {
                        // everything requested so far has arrived from here on: said once, outside the per-row branches - left to the compiler, the
                        // wait sits inside the first row's branch and every later row waits again, then for the stores before it as well
                        __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
                        float4 rn[GRP];
#pragma unroll
                        for (int i = 0; i < GRP; ++i) rn[i] = rv[i];
                        if (res && i0 + GRP < NCH) {
#pragma unroll
                            for (int i = 0; i < GRP; ++i) rn[i] = res_load(i0 + GRP + i, en);
                        }
#pragma unroll
                        for (int i = 0; i < GRP; ++i) {
                            const int row = erow0 + (i0 + i) * RSTEP;
                            const int m = m0 + row, n = en;
                            if (m < p.M) {
                                float4 v = *reinterpret_cast<const float4*>(&smem[row * SP + eq]);
                                // (real block-uniform branches: with the operands in registers either way, the compiler turns `if (scale)` / `if (res)`
                                // into compute-and-select - 12 vector instructions per chunk that the launches without affine or residual, the
                                // Winograd position GEMMs, do not need: +1.5 % on those, profiles/r17_epilogue.md)
                                if (scale) {
                                    asm volatile("");
                                    v.x = fmaf(v.x, sc4.x, sh4.x); v.y = fmaf(v.y, sc4.y, sh4.y);
                                    v.z = fmaf(v.z, sc4.z, sh4.z); v.w = fmaf(v.w, sc4.w, sh4.w);
                                }
                                if (res) {
                                    asm volatile("");
                                    v.x += rv[i].x; v.y += rv[i].y; v.z += rv[i].z; v.w += rv[i].w;
                                }
                                if (relu) {
                                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                                }
                                if (prelu) {
                                    v.x = v.x > 0.f ? v.x : v.x * sl4.x; v.y = v.y > 0.f ? v.y : v.y * sl4.y;
                                    v.z = v.z > 0.f ? v.z : v.z * sl4.z; v.w = v.w > 0.f ? v.w : v.w * sl4.w;
                                }
                                *reinterpret_cast<float4*>(out + pixel_row(m) * out_cs + n) = v;
                                if (gn) {
                                    const double a = (double)v.x + (double)v.y + (double)v.z + (double)v.w;
                                    const double b = (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
                                    if (m < m_next) { s0 += a; q0 += b; } else { s1 += a; q1 += b; }
                                }
                            }
                        }
#pragma unroll
                        for (int i = 0; i < GRP; ++i) rv[i] = rn[i];
                    }
                }
            }
            if (gn) {
                const int n = nb + (t % CPR) * 4;
                if (n < p.Cout) {
                    const int grp = n / p.gn_cpg;
                    atomicAdd(&gacc[grp * 2], s0);
                    atomicAdd(&gacc[grp * 2 + 1], q0);
                    if (s1 != 0.0 || q1 != 0.0) {
                        atomicAdd(&gacc[64 + grp * 2], s1);
                        atomicAdd(&gacc[64 + grp * 2 + 1], q1);
                    }
                }
            }
        } else {
            for (int c = t; c < BM * SW; c += NTH) {
                const int row = c / SW, q = c - row * SW;
                const int m = m0 + row, n = nb + q;
                if (m < p.M && n < p.Cout) {
                    float v = smem[row * SP + q];
                    if (scale) v = fmaf(v, scale[n], shift[n]);
                    const long pm = pixel_row(m);
                    if (res) v += HOUT ? (float)reinterpret_cast<const H16*>(res)[pm * p.res_cs + n] : res[pm * p.res_cs + n];
                    if (relu) v = fmaxf(v, 0.f);
                    if (prelu) v = v > 0.f ? v : v * prelu[n];
                    if (hstore) reinterpret_cast<H16*>(out)[pm * out_cs + n] = (H16)v;
                    else out[pm * out_cs + n] = v;
                }
            }
        }
        if (j + 1 < TN) __syncthreads();
    }
    if (gn) {
        __syncthreads();
        if (t < 128) {
            const double v = gacc[t];
            const int b = b0 + (t >> 6);
            if (v != 0.0 && b < p.B)
                atomicAdd(&p.gn_sum[(((long)g * p.B + b) * p.gn_groups) * 2 + (t & 63)], v);
        }
    }
}

// sums the split-K partial tiles in a fixed order (deterministic) and applies the fused epilogue; block x owns the
// contiguous element range [x*chunk, (x+1)*chunk) so that, with GroupNorm sums requested, it meets at most two images
template <int V>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ConvP p, int S, int G, long chunk) {
    using vec = __attribute__((ext_vector_type(V))) float;
    const int g = blockIdx.y;
    const long MN = (long)p.ws_rows * p.Cout;          // the rows [ws_row0, ws_row0 + ws_rows) were computed in pieces
    const float* __restrict__ scale = p.scale ? p.scale + g * p.ss_gs : nullptr;
    const float* __restrict__ shift = p.shift ? p.shift + g * p.ss_gs : nullptr;
    const float* __restrict__ prelu = p.prelu ? p.prelu + g * p.ss_gs : nullptr;
    const float* __restrict__ res = p.res ? p.res + (long)g * p.res_gs : nullptr;
    float* __restrict__ out = p.out + (long)g * p.out_gs;
    const long i0 = blockIdx.x * chunk, i1 = min(MN, i0 + chunk);
    __shared__ double gacc[2 * 32 * 2];
    const bool gn = p.gn_sum != nullptr;
    int b0 = 0;
    long m_next = 0;
    if (gn) {
        if (threadIdx.x < 128) gacc[threadIdx.x] = 0.0;
        b0 = (int)((i0 / p.Cout + p.ws_row0) / p.ohw);
        m_next = (long)(b0 + 1) * p.ohw;
        __syncthreads();
    }
    for (long i = i0 + (long)threadIdx.x * V; i < i1; i += 256 * V) {
        const long mr = i / p.Cout;
        const int n = (int)(i - mr * p.Cout);
        const long m = mr + p.ws_row0;
        const long pm = p.zones ? zone_row_of(p, m) : m;          // tensor row of GEMM row m (ConvP::zones: the rows of an image in zone order)
        // everything this element needs is requested before the first value is used (V == 4: n, the strides and the bases are multiples of
        // 4 floats - host), the slabs four at a time; the slabs are added one by one in slab order, as they always were
        vec sc, sh, sl, rr;
        if (scale) { sc = *reinterpret_cast<const vec*>(scale + n); sh = *reinterpret_cast<const vec*>(shift + n); }
        if (prelu) sl = *reinterpret_cast<const vec*>(prelu + n);
        if (res) rr = *reinterpret_cast<const vec*>(res + pm * p.res_cs + n);
        const float* const w0 = p.ws + (long)g * MN + i;
        const long sstep = (long)G * MN;
        vec v = *reinterpret_cast<const vec*>(w0);
        int s = 1;
        for (; s + 3 < S; s += 4) {
            const vec a = *reinterpret_cast<const vec*>(w0 + s * sstep), b = *reinterpret_cast<const vec*>(w0 + (s + 1) * sstep);
            const vec c = *reinterpret_cast<const vec*>(w0 + (s + 2) * sstep), d = *reinterpret_cast<const vec*>(w0 + (s + 3) * sstep);
            v += a; v += b; v += c; v += d;
        }
        for (; s < S; ++s) v += *reinterpret_cast<const vec*>(w0 + s * sstep);
        float o[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float x = V == 1 ? v[0] : v[e];
            if (scale) x = fmaf(x, V == 1 ? sc[0] : sc[e], V == 1 ? sh[0] : sh[e]);
            if (res) x += V == 1 ? rr[0] : rr[e];
            if (p.relu) x = fmaxf(x, 0.f);
            if (prelu) x = x > 0.f ? x : x * (V == 1 ? sl[0] : sl[e]);
            o[e] = x;
        }
        if constexpr (V == 4) {
            *reinterpret_cast<vec*>(out + pm * p.out_cs + n) = vec{o[0], o[1], o[2], o[3]};
            if (gn) {   // host: V == 4 and 4 | channels per group whenever sums are requested
                const double a = (double)o[0] + (double)o[1] + (double)o[2] + (double)o[3];
                const double b = (double)o[0] * o[0] + (double)o[1] * o[1] + (double)o[2] * o[2] + (double)o[3] * o[3];
                const int slot = (m < m_next ? 0 : 64) + (n / p.gn_cpg) * 2;
                atomicAdd(&gacc[slot], a);
                atomicAdd(&gacc[slot + 1], b);
            }
        } else {
            out[pm * p.out_cs + n] = o[0];
        }
    }
    if (gn) {
        __syncthreads();
        if (threadIdx.x < 128) {
            const double v = gacc[threadIdx.x];
            const int b = b0 + (threadIdx.x >> 6);
            if (v != 0.0 && b < p.B)
                atomicAdd(&p.gn_sum[(((long)g * p.B + b) * p.gn_groups) * 2 + (threadIdx.x & 63)], v);
        }
    }
}


// ---- work distribution -------------------------------------------------------------------------------------------
// A launch is (tile shape, S = number of K partitions).  `bpc` blocks of a tile shape are resident per CU (registers /
// LDS), so 256*bpc blocks run at once and a launch whose block count is an awkward multiple of that leaves CUs idle in
// its last round; with few blocks (small batches) most of the chip idles throughout.  Splitting K trades that for a
// second pass over the output (partial tiles -> splitk_reduce_kernel, fixed summation order: deterministic).
// The choice is a small cost model fitted to a sweep of every convolution shape of the refiner at 1-16 frames
// (tools/conv_sweep.py, profiles/r01i_conv_sweep_*.md): cost in K-slice units of one CU-round,
//     (nk/S + c) * W(blocks per CU)  [+ lat + (2S+1) * outputs * per_float   if S > 1]
// where W counts rounds, a partly filled last round being cheaper than a full one (a lone block runs faster).
struct SplitModel { double c, thr1, thr2, per_float, lat; };
// (fitted in round 1 with 3 / 7 resident blocks per CU; since the two-level accumulation of round 3 the 128x128 tiles run
// 2 and the 64x64 tiles 5 - igemm_occupancy() - and the constants are used as they are)
constexpr SplitModel kModel3 = {3.9, 0.74, 0.995, 1.09e-7, 13.7};   // 2-3 resident blocks per CU (128x128, 256x32, 128x64)
constexpr SplitModel kModel7 = {7.5, 0.0, 0.0, 6.05e-7, 40.0};      // 5 resident blocks per CU (64x64)
constexpr int kMinSlicesPerSplit = 6;

static double launch_cost(long blocks, int nk, double outputs, int S, int bpc, const SplitModel& m) {
    const long per_cu = (blocks * S + 255) / 256;
    const long q = per_cu / bpc, rem = per_cu % bpc;
    double W = (double)bpc * q;
    if (rem) W += bpc <= 3 ? (rem == 1 ? 1.0 / m.thr1 : 2.0 / m.thr2) : (rem > 0.55 * bpc ? (double)rem : 0.55 * bpc);
    double t = ((double)nk / S + m.c) * W;
    if (S > 1) t += m.lat + (2.0 * S + 1.0) * outputs * m.per_float;
    return t;
}

static int choose_split(const ConvP& p, int G, int BM, int BN, int bpc) {
    const int nk = p.Kpad / BK;
    if (!p.ws) return 1;
    const long blocks = (long)((p.M + BM - 1) / BM) * ((p.Cout + BN - 1) / BN) * G;
    const double outputs = (double)G * p.M * p.Cout;
    const SplitModel& m = bpc <= 3 ? kModel3 : kModel7;
    int best = 1;
    double best_t = launch_cost(blocks, nk, outputs, 1, bpc, m);
    for (int S = 2; S <= 16 && nk / S >= kMinSlicesPerSplit; ++S) {
        if ((double)S * outputs > (double)p.ws_floats) break;
        const double t = launch_cost(blocks, nk, outputs, S, bpc, m);
        if (t < 0.97 * best_t) { best_t = t; best = S; }
    }
    return best;
}

// one launch site for both loaders of an instantiation
template <int BM, int BN, int WM, int WN, int MODE, int DT>
static void launch_igemm(bool lean, dim3 grid, dim3 block, hipStream_t st, const ConvP& p) {
    if (lean) hipLaunchKernelGGL((conv_igemm_f32<BM, BN, WM, WN, MODE, DT, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((conv_igemm_f32<BM, BN, WM, WN, MODE, DT, false>), grid, block, 0, st, p);
}

// the LEAN loader of conv_igemm_f32: block-uniform filter taps (every K-slice inside one tap), 32-bit byte offsets, <= 32 taps;
// fills p.lean_in_bytes (the extent of one group's input view, the buffer descriptor's range)
static bool lean_loader(ConvP& p) {
    if (!tune().lean_loader || p.in2) return false;
    if (p.Cin % BK || p.K != p.Kpad || p.kh * p.kw > 32) return false;
    const long in_bytes = ((long)p.B * p.H * p.W * p.in_cs) * 4, w_bytes = (long)p.Cout * p.Kpad * 4;
    if (in_bytes >= 0x7fffff00L || w_bytes >= 0x7fffff00L) return false;
    // a padding tap of the first pixel reaches at most (pad rows + pad columns) before the view: offsets stay inside int32
    p.lean_in_bytes = (int)in_bytes;
    return true;
}

template <int BM, int BN, int WM, int WN>
static int run(ConvP p, int G, int S, hipStream_t st) {
    p.mtiles = (p.M + BM - 1) / BM;
    p.ntiles = (p.Cout + BN - 1) / BN;
    p.vec_out = (p.Cout % 4 == 0) && (p.out_cs % 4 == 0) && (((uintptr_t)p.out & 15) == 0) && (p.out_gs % 4 == 0) &&
                (!p.res || ((p.res_cs % 4 == 0) && (((uintptr_t)p.res & 15) == 0) && (p.res_gs % 4 == 0))) &&
                (!p.scale || ((((uintptr_t)p.scale & 15) == 0) && (p.ss_gs % 4 == 0))) &&
                (!p.prelu || ((((uintptr_t)p.prelu & 15) == 0) && (p.ss_gs % 4 == 0)));
    if (p.es == 2 && p.vec_out && p.Cout % 8 == 0 && p.out_cs % 8 == 0 && p.out_gs % 8 == 0 && !p.prelu &&
        (!p.res || (p.res_cs % 8 == 0 && p.res_gs % 8 == 0)) && (!p.scale || (p.ss_gs % 8 == 0 && (((uintptr_t)p.scale & 31) == 0))))
        p.vec_out = 2;        // fp16 tensors: 16-byte accesses of 8 channels
    // GroupNorm sums in the epilogue: 16-byte stores, whole float4s inside one norm group, at most 32 groups, and images
    // of at least one tile of rows (a tile then meets at most two images); otherwise a separate pass over the output
    const bool gn_sep = p.gn_sum && !(p.vec_out && p.gn_cpg % 4 == 0 && p.gn_groups <= 32 && p.ohw >= BM && p.ohw >= 8);
    double* const gn_sum = p.gn_sum;
    if (gn_sep) p.gn_sum = nullptr;
    auto gn_separate = [&]() {
        if (!gn_sep) return 0;
        View o;
        o.p = p.out; o.B = p.B; o.H = p.OH; o.W = p.OW; o.C = p.Cout; o.cs = p.out_cs; o.gs = p.out_gs; o.es = p.es;
        return launch_gn_stats(o, p.B, G, p.gn_groups, gn_sum, st, false);
    };
    const int nk = p.Kpad / BK;
    if (p.ws && tune().force_split > 0) S = tune().force_split;
    if (!p.ws || S < 1 || p.es == 2) S = 1;
    if (S > nk) S = nk;
    while (S > 1 && (size_t)S * G * p.M * p.Cout > p.ws_floats) --S;
    p.kchunk = (nk + S - 1) / S;
    S = (nk + p.kchunk - 1) / p.kchunk;        // no empty partitions
    p.ksplit = S;
    p.ws_rows = p.M;
    p.ws_row0 = 0;
    p.nfull = p.mtiles * p.ntiles;
    p.tail_shift = 0;
    const dim3 block(WM * WN * 64);
    // stage profile: algorithmic traffic = the input tensor, the weights and the output (+ residual) once each
    // (fp16 data path: Cin / K are in 4-byte units here - two halfs each - so the input and weight bytes come out right)
    const double out_bytes = (double)p.es * G * (double)p.M * p.Cout;
    const double conv_bytes = 4.0 * G * ((double)p.B * p.H * p.W * p.Cin + (double)p.Cout * p.K) + out_bytes * (p.res ? 2.0 : 1.0);
    const double conv_flops = 2.0 * G * (double)p.M * p.K * p.Cout * (p.es == 2 ? 2.0 : 1.0);
    const char* tag = p.tag ? p.tag : "conv_gemm";
    // a launch of the bf16x3 mode that keeps the exact fp32 MFMA kernel is profiled under its own stage, so that the
    // bench prices each matrix pipe with the work it actually executed
    auto exact_fallback = [&]() { p.bf16 = 0; tag = "conv_gemm_f32pipe"; };
    auto reduce = [&](int parts) {
        const long MN = (long)p.ws_rows * p.Cout;
        ProfScope prof("splitk_reduce", 4.0 * G * (double)MN * (parts + 1.0), 0.0, st);
        const int V = p.vec_out ? 4 : 1;
        long chunk = (MN + 2047) / 2048;                       // at most 2048 blocks ...
        if (chunk < 256L * V * 4) chunk = 256L * V * 4;        // ... of at least 4 elements-vectors per thread
        chunk = (chunk + V - 1) / V * V;
        if (p.gn_sum && chunk / p.Cout + 2 > p.ohw) chunk = (long)(p.ohw - 2) * p.Cout / V * V;   // at most two images per block
        const int blocks = (int)((MN + chunk - 1) / chunk);
        if (p.vec_out)
            hipLaunchKernelGGL(splitk_reduce_kernel<4>, dim3(blocks, G), dim3(256), 0, st, p, parts, G, chunk);
        else
            hipLaunchKernelGGL(splitk_reduce_kernel<1>, dim3(blocks, G), dim3(256), 0, st, p, parts, G, chunk);
    };
    const bool skip = BM != 256 && p.skip_rows && p.kmode == 0 && p.kh > 1 && p.Cin % BK == 0 && p.K == p.Kpad && p.ohw > 0;
    if (p.es == 2) {        // fp16 data path: one K pass (the loop is 16x shorter than the fp32 one), padded filter rows skipped; never
                            // persistent (below), split or on the fp32 pipe
                            // (128x256 tiles - wave tiles of 64x128, 25 % fewer LDS fragment reads per MFMA - measured level with
                            // 128x128: 15.84 against 15.79 ms of convolutions, profiles/r10b_h16_loader.md)
        ProfScope prof(tag, conv_bytes, conv_flops, st);
        const dim3 grid(p.mtiles * p.ntiles, 1, G);
        const bool lean = lean_loader(p);
        if (skip) launch_igemm<BM, BN, WM, WN, 3, 4>(lean, grid, block, st, p);
        else launch_igemm<BM, BN, WM, WN, 0, 4>(lean, grid, block, st, p);
        QB_CHECK(hipGetLastError());
        return gn_separate();
    }
    const bool lean = lean_loader(p);
    // Persistent launch (conv_persist.hip): one block per resident slot walks whole tiles and an equal share of the
    // K-slices of the remainder
    // (launches of a few dozen tiles - small batches - keep the one-tile-per-block kernel and its fitted split-K model:
    // sharing every tile's K between all resident blocks writes more partial tiles than that model's 2-8 partitions)
    // (fp16 data path: only the fused projection shortcuts - launch_conv_dual - go persistent: with 16x the matrix rate the
    // persistent kernel's per-slice tap arithmetic costs the 3x3 layers 30 %, profiles/r04a_f16_persistent_all_rejected.md)
    if (tune().persist && p.es != 2 && ((BM == 128 && BN == 128) || (tune().persist == 2 && (BM == BN || BM == 256))) && BN <= 128 && !skip && p.ws && conv_persistent_ok(p) &&
        (long)p.mtiles * p.ntiles * G >= tune().persist_min_tiles &&
        ((long)p.mtiles * p.ntiles * G >= 256L * (p.es == 2 ? 3 : BM == 64 ? 5 : 2) || nk >= tune().persist_min_nk)) {      // several tiles per block, or K worth sharing
        if (p.bf16 == 3 && (BM == 256 || (BM == 64 && nk <= 8))) exact_fallback();
        const int bpc = p.es == 2 ? 3 : BM == 64 ? 5 : 2;          // conv_persist.hip: pk_occupancy()
        if (p.ws_floats >= conv_persistent_ws_floats(BM, BN, bpc)) {
            {
                ProfScope prof(tag, conv_bytes, conv_flops, st);
                int rc = 0;
                if constexpr (BM == BN || BM == 256) rc = launch_conv_persistent<BM, BN, WM, WN>(p, G, bpc, st);
                if (rc) return rc;
            }
            return gn_separate();
        }
    }
    // Split tail: with more than one round of tiles, cut the tiles of the ragged last round into 2^shift K-pieces that
    // together fill about one more (short) round; taken when the model prices it below the launch chosen so far.
    constexpr int BPC = igemm_occupancy(BM, BN);
    const long slots = 256L * BPC, tiles = (long)p.mtiles * p.ntiles, blocks_all = tiles * G;
    if (p.ws && !p.bf16 && tune().tail_split && tune().force_split == 0 && BM == 128 && blocks_all > slots && nk >= 32) {
        const long nfull = (blocks_all / slots) * slots / G / p.ntiles * p.ntiles;    // per group, whole tile rows
        const long rem = tiles - nfull;
        int shift = 0;
        while (shift < 3 && (rem * G << (shift + 1)) <= slots + slots / 8 && (nk >> (shift + 1)) >= 8) ++shift;
        const long row0 = nfull / p.ntiles * BM;
        const double rem_outputs = (double)G * (p.M - row0) * p.Cout;
        if (rem > 0 && shift > 0 && rem_outputs * (1 << shift) <= (double)p.ws_floats) {
            const double whole = launch_cost(blocks_all, nk, (double)G * p.M * p.Cout, S, BPC, kModel3);
            const double tail = (nk + kModel3.c) * BPC * ((double)nfull * G / slots) +
                                launch_cost(rem * G, nk, rem_outputs, 1 << shift, BPC, kModel3);
            if (tail < 0.995 * whole || tune().tail_split == 2) {
                p.nfull = (int)nfull;
                p.tail_shift = shift;
                p.kchunk = (nk + (1 << shift) - 1) >> shift;
                p.ws_row0 = (int)row0;
                p.ws_rows = p.M - (int)row0;
                // every piece owns at least one K-slice: (2^shift - 1) * kchunk < nk because nk >= 8 * 2^shift
                {
                    ProfScope prof(tag, conv_bytes, conv_flops, st);
                    launch_igemm<BM, BN, WM, WN, 2, 0>(lean, dim3((unsigned)(nfull + (rem << shift)), 1, G), block, st, p);
                }
                reduce(1 << shift);
                QB_CHECK(hipGetLastError());
                return gn_separate();
            }
        }
    }
    // bf16x3 pays where the matrix pipe is the limit.  The HBM-bound short-K launches on 64x64 tiles (bottleneck conv3 +
    // residual) and the 32-column head layers on 256x32 tiles lose occupancy to its three LDS planes and gain nothing
    // (profiles/r02j_conv_layers_dtype{0,3}.md): they keep the exact fp32 MFMA kernel, which is at least as accurate.
    // (so does the dilated layer that skips padded filter rows on 64x64 tiles - ASPP d = 18 on a 30-row map: 1.49 ms as bf16x3 against 1.03 ms
    //  exact at batch 16, profiles/r12_final_conv_layers_dtype{0,3}.md)
    if (p.bf16 == 3 && (BM == 256 || (BM == 64 && (nk <= 8 || skip)))) exact_fallback();
    if (p.bf16) {
        {
            ProfScope prof(tag, conv_bytes, conv_flops, st);
            const dim3 grid(p.mtiles * p.ntiles, S, G);
            if (p.bf16 == 3) {
                if (S > 1) launch_igemm<BM, BN, WM, WN, 1, 3>(lean, grid, block, st, p);
                else launch_igemm<BM, BN, WM, WN, 0, 3>(lean, grid, block, st, p);
            } else if (p.bf16 == 2) {
                if (S > 1) launch_igemm<BM, BN, WM, WN, 1, 2>(lean, grid, block, st, p);
                else launch_igemm<BM, BN, WM, WN, 0, 2>(lean, grid, block, st, p);
            } else {
                if (S > 1) launch_igemm<BM, BN, WM, WN, 1, 1>(lean, grid, block, st, p);
                else launch_igemm<BM, BN, WM, WN, 0, 1>(lean, grid, block, st, p);
            }
        }
        if (S > 1) reduce(S);
        QB_CHECK(hipGetLastError());
        return gn_separate();
    }
    p.zones = 0;
    if (skip && tune().zone_cols && lean && BM == 64 && p.kh == 3 && p.kw == 3 && p.stride == 1 && p.pad == p.dil && p.OH == p.H && p.OW == p.W) {
        // filter columns skipped as well (ConvP::zones): 3x3, stride 1, pad = dil, the LEAN kernel on 64-row tiles (a 128-row tile of an
        // 18-column zone spans 7 map rows and loses in filter rows what it gains in columns), at least a quarter of the columns without
        // one of their taps.  The split-K form too: its partial tiles go to the workspace in GEMM row order, the reduce kernel maps them.
        const int a = p.dil, b = p.W - p.dil;
        p.zx1 = std::max(0, std::min(std::min(a, b), p.W));
        p.zx2 = std::max(0, std::min(std::max(a, b), p.W));
        if (a > b || 8 * p.zx1 >= p.W) p.zones = a <= b ? 1 : 2;
        // (one frame of 30 x 40 is 19 tiles whose partitions all run at once: the launch lasts as long as its longest block, and the two tiles of
        // the 4-column middle zone - 16 map rows each, all nine taps - are longer than any tile of the row order: 119 -> 127 us; from 38 tiles on
        // - two frames, or one of 45 x 80 - the shorter average wins: 210 -> 186 us, 318 -> 283 us)
        if (S > 1 && p.mtiles < 32) p.zones = 0;
    }
    if (S > 1) {
        {
            ProfScope prof(tag, conv_bytes, conv_flops, st);
            if (skip) launch_igemm<BM, BN, WM, WN, 4, 0>(lean, dim3(p.mtiles * p.ntiles, S, G), block, st, p);
            else launch_igemm<BM, BN, WM, WN, 1, 0>(lean, dim3(p.mtiles * p.ntiles, S, G), block, st, p);
        }
        reduce(S);
    } else {
        ProfScope prof(tag, conv_bytes, conv_flops, st);
        if (skip) launch_igemm<BM, BN, WM, WN, 3, 0>(lean, dim3(p.mtiles * p.ntiles, 1, G), block, st, p);
        else launch_igemm<BM, BN, WM, WN, 0, 0>(lean, dim3(p.mtiles * p.ntiles, 1, G), block, st, p);
    }
    QB_CHECK(hipGetLastError());
    return gn_separate();
}


int launch_conv(const ConvP& p0, int G, hipStream_t st) {
    ConvP p = p0;
    p.acc_chunk = tune().acc_chunk;
    if (p.es != 2) p.es = 4;
    if (p.es == 2) {
        // fp16 data path (activations and packed weights are fp16 in HBM): the kernel moves the operands as 4-byte units, so
        // every K-side quantity is handed over in units of two halfs; Cout / M and the output strides stay in elements
        if (p.bf16 != 2) return fail("conv: fp16 tensors need compute_dtype 2");
        if (p.Cin % 8 || p.in_cs % 8 || p.Kpad % 64 || p.K % 2 || (p.in_gs & 7) || (p.w_gs & 1) || p.in2 || p.prelu)
            return fail("conv (fp16 data path): Cin / channel stride must be multiples of 8 and Kpad a multiple of 64");
        p.Cin /= 2; p.in_cs /= 2; p.K /= 2; p.Kpad /= 2; p.in_gs /= 2; p.w_gs /= 2;
    }
    if (p.Cin % 4 || p.in_cs % 4 || p.Kpad % BK || p.K > p.Kpad)
        return fail("conv: Cin / channel stride must be multiples of 4 and Kpad a multiple of 32");
    if (((uintptr_t)p.in & 15) || ((uintptr_t)p.w & 15) || (p.in_gs & 3) || (p.w_gs & 3))
        return fail("conv: operands must be 16-byte aligned");
    if (p.M <= 0 || p.Cout <= 0) return fail("conv: empty problem");
    // the loader steps its filter-tap position at most BK/8 times per K-slice (found by tools/conv_fuzz.py: Cin = 4 with a 3x3
    // filter read the wrong taps; the refiner pads its 4-channel input to 8)
    if (p.kh * p.kw > 1 && p.Cin < 8) return fail("conv: filters larger than 1x1 need at least 8 input channels (pad the input)");
    if (p.kmode && (p.Cin % BK || p.K != p.Kpad)) return fail("conv: slice-major weights need Cin % 32 == 0");
    if ((p.scale == nullptr) != (p.shift == nullptr)) return fail("conv: scale and shift go together");
    if (p.es == 2 && tune().force_tile == 0) {      // the wide layers of the fp16 data path: 256 x 256 tiles, LDS-DMA pipeline (conv_h8.hip)
        const int rc = launch_conv_h8(p, G, st);
        if (rc != 1) return rc;
    }
    if (p.n_stats) return fail("conv: a launch that normalises its input needs the fp16 patch kernel (conv_h8.hip)");
    if (p0.dil_g[0]) {
        // a grouped launch with per-group dilation that conv_h8.hip did not take: the groups one after the other, each with its own
        for (int g = 0; g < G; ++g) {
            ConvP q = p0;
            const long es = p0.es == 2 ? 2 : 4;
            q.dil = q.pad = p0.dil_g[g];
            q.dil_g[0] = 0;
            q.in = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p0.in) + (long)g * p0.in_gs * es);
            q.w = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p0.w) + (long)g * p0.w_gs * es);
            q.out = reinterpret_cast<float*>(reinterpret_cast<char*>(p0.out) + (long)g * p0.out_gs * es);
            if (p0.res) q.res = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p0.res) + (long)g * p0.res_gs * es);
            if (p0.scale) { q.scale = p0.scale + (long)g * p0.ss_gs; q.shift = p0.shift + (long)g * p0.ss_gs; }
            if (p0.gn_sum) q.gn_sum = p0.gn_sum + (long)g * p0.B * p0.gn_groups * 2;
            const int rc = launch_conv(q, 1, st);
            if (rc) return rc;
        }
        return 0;
    }
    if (p.es == 4 && p.bf16 == 3 && tune().force_tile == 0 && tune().force_split == 0) {      // bf16x3: the wide 1x1 launches, pre-split weights, LDS-DMA pipeline (conv_x8.hip)
        const int rc = launch_conv_x8(p, G, st);
        if (rc != 1) return rc;
    }
    switch (tune().force_tile) {
        case 1: return run<64, 64, 2, 2>(p, G, 1, st);
        case 2: return run<128, 128, 2, 2>(p, G, 1, st);
        case 4: return run<256, 32, 4, 1>(p, G, 1, st);
        case 3: return run<128, 64, 2, 2>(p, G, 1, st);
        default: break;
    }
    // Tile shape (sweep: profiles/r01i_conv_sweep_*.md).  <= 32 output channels: 256x32.  <= 64 channels, and the
    // memory-bound residual 1x1 layers with a short K (res2-4 conv3): 64x64, whose 7 resident blocks per CU keep more
    // loads in flight.  Otherwise 128x128, split when the model says so; 64x64 again for launches too small for that.
    const int nk = p.Kpad / BK;
    if (p.Cout <= 32) return run<256, 32, 4, 1>(p, G, choose_split(p, G, 256, 32, igemm_occupancy(256, 32)), st);
    // 33-64 output channels without a residual, at least a round of tiles: 128x64 (each wave 64x32: the weight fragments
    // are read from LDS half as often as with 64x64 tiles; +3-5 % on stem.conv3 / res2 conv1, conv2 - tools/tile_ab.py)
    if (tune().tile_128x64 && p.Cout <= 64 && p.Cout > 32 && !p.res && (long)((p.M + 127) / 128) * G >= 1024) return run<128, 64, 2, 2>(p, G, 1, st);
    // (64 x 256 tiles for the residual 1x1 layers - input rows read once, 512-byte row segments - measured 30-50 % SLOWER than
    // 64x64 in both the fp32 and the fp16 path, profiles/r03x_tile_64x256_rejected.txt: those layers live on blocks in flight)
    if (p.Cout <= 64 || (p.res && nk <= 8 && p.Cout >= 128)) return run<64, 64, 2, 2>(p, G, choose_split(p, G, 64, 64, igemm_occupancy(64, 64)), st);
    // dilated layers that skip the filter rows lying in the padding (ASPP d = 18 on a 30-row map): 64-row tiles span 2-3 map rows
    // instead of 4-5, so more of them see a filter row entirely in the padding - 1.28 -> 1.07 ms on that layer (tools/aspp_d18_ab.py)
    if (p.skip_rows && p.es != 2 && p.kmode == 0 && p.kh > 1 && p.dil * 5 >= p.H * 2)
        return run<64, 64, 2, 2>(p, G, choose_split(p, G, 64, 64, igemm_occupancy(64, 64)), st);
    const long tiles128 = (long)((p.M + 127) / 128) * ((p.Cout + 127) / 128) * G;
    const int s128 = choose_split(p, G, 128, 128, igemm_occupancy(128, 128));
    if (tiles128 < 64 || (tiles128 < 384 && s128 == 1)) return run<64, 64, 2, 2>(p, G, choose_split(p, G, 64, 64, igemm_occupancy(64, 64)), st);
    // 1x1 layers whose 128 x 128 tiling is at most ~5 tiles per CU (res4 / res5 conv1, the ASPP 1x1 and res3 conv1 at batch 16; every wide 1x1 layer
    // at batches 1-4): 64 x 64 tiles, one per block - five to seven resident blocks per CU hide what two 128 x 128 blocks (split-K or persistent) cannot.
    // tools/conv_sweep.py at 1 / 2 / 4 / 8 / 16 frames, round 6: 0.211 -> 0.176 ms on res4 conv1 and 0.357 -> 0.322 on res5.0 conv1 at batch 16,
    // 0.134 -> 0.107 on res5 conv3 at batch 2; the batch-1 step 3.71 -> 3.50 ms, 1280x720 7.40 -> 6.96 ms (profiles/r14d_tile_rule.md)
    if (tune().small_n_64 && p.es != 2 && p.kh == 1 && p.kw == 1 && !p.in2 && nk >= 16 && tiles128 <= 1280)
        return run<64, 64, 2, 2>(p, G, choose_split(p, G, 64, 64, igemm_occupancy(64, 64)), st);
    // ... and, exact fp32, every 1x1 GEMM of K <= 1024 however large the launch - the Winograd position GEMMs of the 256- / 512-channel layers
    // (fusion_res2 / res3, res4 / res5 conv2), the fusion 1x1 convolutions of res2 / res3, res5 conv3: a K loop of 8-32 slices is over before a
    // 128 x 128 block has amortised its prologue and epilogue, and only two of those fit a CU.  Per layer at batch 16 (rocprofv3, forced tiles):
    // fusion_res3 conv0 0.992 -> 0.936 ms, res4 conv2 0.170 -> 0.154, res5.1 conv3 0.685 -> 0.639; K = 2048 / 4096 (ASPP, fusion_res5) lose 1-3 %
    // on 64 x 64 tiles and keep the persistent 128 x 128 launch (profiles/r14d_tile_rule.md)
    // (128 x 64 tiles for fusion_res5.conv, K = 4096: 2.42 against 2.49 ms in the layer table, nothing in the step - not kept)
    if (tune().small_n_64 == 1 && p.es != 2 && p.bf16 == 0 && p.kh == 1 && p.kw == 1 && !p.in2 && nk <= 32)
        return run<64, 64, 2, 2>(p, G, choose_split(p, G, 64, 64, igemm_occupancy(64, 64)), st);
    return run<128, 128, 2, 2>(p, G, s128, st);
}

}  // namespace quber
