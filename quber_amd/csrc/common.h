// Internal declarations shared by the HIP translation units of libquber_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

namespace quber {

// A strided NHWC view: element (b,y,x,c) lives at p[((b*H + y)*W + x)*cs + c]; `p` already points at
// the view's first channel, so writers can target a channel slice of a wider (concatenated) buffer.
// `gs` is the element distance between the G independent operand sets of one grouped launch
// (blockIdx.z), e.g. the rgb / depth encoder streams.
struct View {
    float* p = nullptr;      // base address (typed float* whatever the element type: never indexed directly - use at())
    int B = 0, H = 0, W = 0, C = 0;
    int cs = 0;              // channel stride of a pixel, in elements
    long gs = 0;             // group stride, in elements
    int es = 4;              // element size in bytes: 4 = fp32 (default), 2 = fp16 (the fp16 data path, compute_dtype 2)
    float* at(long elems) const { return p ? reinterpret_cast<float*>(reinterpret_cast<char*>(p) + elems * es) : nullptr; }
};

struct ConvP {
    const float* in;
    const float* w;      // [G][Cout][Kpad]; k order per `kmode`
    const float* scale;  // [G][Cout] or null
    const float* shift;  // [G][Cout] or null
    const float* res;    // residual view or null
    const float* prelu;  // [G][Cout] PReLU slopes or null (LMFFNet's BN+PReLU epilogue)
    float* out;
    int B, H, W, Cin, in_cs;
    int OH, OW, Cout, out_cs;
    int res_cs;
    int K, Kpad;
    int kh, kw, stride, pad, dil;
    int relu;
    int M;               // B*OH*OW
    int mtiles, ntiles;
    int vec_out;         // 16-byte epilogue accesses are legal for this launch
    float* ws;           // split-K workspace (partial tiles [ksplit][G][M][Cout]); null = never split
    size_t ws_floats;    // its capacity
    int ksplit;          // number of K partitions (grid.y); 1 = direct epilogue
    int kchunk;          // K-slices per partition
    int nfull, tail_shift;   // split tail: tiles computed whole, log2(pieces per remaining tile)
    int ws_rows, ws_row0;    // output rows held in the workspace slabs
    double* gn_sum;          // GroupNorm sums of the output [G][B][gn_groups][sum, sum of squares] (accumulated), or null
    int gn_groups, gn_cpg;   // norm groups, channels per group
    int ohw;                 // OH * OW
    int kmode;           // 0: k = (tap, c)   1: k = (c/32, tap, c%32)  (weights packed accordingly)
    int skip_rows;       // kmode 0 only: blocks skip the filter rows that are padding for all their output rows
    long in_gs, out_gs, res_gs, w_gs;
    int ss_gs;
    const char* tag;     // stage-profile tag of the launch (null = "conv_gemm")
    int pk_T, pk_tpg, pk_min, pk_in_bytes, pk_debug;   // persistent launch (conv_persist.hip): tiles over all groups, tiles per group, shortest K share
    // second input of a dual 1x1 launch (launch_conv_dual: K = K1 channels of `in`, then the channels of `in2` at stride2), or null
    const float* in2;
    int in2_cs, H2, W2, stride2, K1, pk_in2_bytes;
    long in2_gs;
    int es;              // element size of the activations (in / in2 / res / out): 4 = fp32, 2 = fp16 in HBM (conv_igemm_f32 DT 4)
    int acc_chunk;       // K-slices per accumulation chunk (two-level fp32 accumulation: the MFMA accumulator is folded into a
                         // second register set every acc_chunk slices); 0 = one sequential chain over K
    int lean_in_bytes;   // LEAN loader (conv_igemm.hip): bytes of one group's input view (buffer descriptor range)
    unsigned dv_m[5], dv_s[5];   // conv_h8.hip: magic numbers of the divisions by ohw, OW, tiles per group, channel tiles, channels per norm group (h8_magic)
    const void* w3;      // bf16x3 mode: the weights pre-split into three bf16 planes [3][w3_plane] (launch_split_bf16x3), same [G][Cout][Kpad] order inside a plane; or null
    long w3_plane;       // elements of one plane
    int dil_g[4];        // per-group dilation = padding of a grouped 3x3 launch (dil_g[0] == 0: `dil` / `pad` for every group); conv_h8.hip reads it, any other
                         //   kernel gets the groups one launch at a time (launch_conv)
    int h8_ss_bytes;     // conv_h8.hip: bytes of the scale / shift vectors over all groups (descriptor range)
    // conv_h8.hip patch kernels: the GroupNorm (+ ReLU) of the INPUT tensor applied on the LDS patch (its producer's norm pass absorbed by this consumer):
    // the sums [G][B][n_groups][2] and affine parameters of the tensor being read (null: none), the workspace for the per-(image, channel) coefficients
    const double* n_stats; const float* n_gamma; const float* n_beta; float* n_coef;
    int n_groups, n_param_gs, n_relu, n_coef_bytes; float n_eps;       // (n_coef_bytes: filled in by the launcher)
    int bf16;            // quber_config.compute_dtype: 0 = fp32 MFMA, 1 = bf16 / 2 = fp16 operands, 3 = fp32 operands split into 3 bf16 terms
    // conv_igemm.hip, skip-rows launch of a dilated 3x3 layer: filter COLUMNS skipped as well.  The GEMM rows of an image are ordered
    // (column zone, y, x) instead of (y, x) - as a serpentine, see zone_pixel there: zone A = columns [0, zx1) - the left tap is padding,
    // B = [zx1, zx2), C = [zx2, W) - the right tap is; a tile inside one zone multiplies only the filter columns that meet the image.
    // 0 = off, 1 = zone B needs all three columns (dil <= W - dil), 2 = only the centre one
    int zones, zx1, zx2;
};

int device_cus();         // compute units of the CURRENT device (cached per device id; plan.hip), <= 0: the query failed
void set_error(const std::string& msg);
int fail(const std::string& msg);

// Stage profiler (benchmark use only).  Between quber_profile_begin and quber_profile_end every launcher brackets its
// kernels with a HIP-event pair recorded on the launch stream, tagged with a stage name and the ALGORITHMIC bytes
// (what the stage must read + write once) and FLOPs of the bracket; outside such a window a ProfScope costs one
// thread-local load.  Scopes do not nest: they sit at the leaves (one kernel, or a kernel plus its tiny helpers).
struct ProfScope {
    int rec;
    hipStream_t st;
    ProfScope(const char* tag, double bytes, double flops, hipStream_t s);
    ~ProfScope();
    ProfScope(const ProfScope&) = delete;
    ProfScope& operator=(const ProfScope&) = delete;
};

// GroupNorm (+ ReLU) folded into the Winograd input transform: sums of the input tensor, affine parameters
struct WinoNorm {
    const double* stats;     // [G][B][groups][sum, sum of squares] of the tensor being read, or null = no normalisation
    const float* gamma;
    const float* beta;
    int groups, cpg, param_gs, relu;
    double n;                // elements per (image, group); filled in by the launcher
    float eps;
};

// one 3x3 / stride 1 / pad = dilation convolution through Winograd F(m x m, 3x3) (winograd.hip)
struct WinoP {
    View in, out;            // NHWC views; groups via View::gs
    const float* u;          // transformed weights [G][16][Cout][Cin]
    const void* u3;          // bf16x3 mode: `u` pre-split into three bf16 planes (conv_x8.hip), or null
    long u3_plane;           // elements of one plane
    const float* uf;         // F(4x4) only: the same weights in the operand order of the single-kernel form (wino_fused.hip), or null
    int algo;                // 1 = the three-kernel pipeline, 2 = the single kernel: fixed when the plan is built, for max_batch, and
                             // honoured at every launch (a layer's algorithm never depends on the batch); 0 = decide at launch
                             // (stand-alone op only)
    const float* scale;      // per-channel affine of the epilogue ([G][Cout], stride ss_gs) or null
    const float* shift;
    int ss_gs, relu;
    int dil;                 // dilation (= padding)
    int m;                   // output tile edge: 2 = F(2x2,3x3), 4 = F(4x4,3x3)
    int dtype;               // arithmetic of the P GEMMs (ConvP::bf16): 0 = fp32 MFMA, 3 = fp32 operands as 3 bf16 terms
    double* gn_sum;          // GroupNorm sums of the output to accumulate ([G][B][gn_groups][2]) or null
    int gn_groups;
    WinoNorm norm;           // normalisation of the INPUT applied on load (stats == null: none)
    float* ws;               // V | M workspace (winograd_ws_floats)
    size_t ws_floats;
    float* splitk_ws;        // forwarded to the grouped GEMM launch
    size_t splitk_floats;
};

// launchers (all asynchronous on `st`, no allocation, no synchronisation)
int launch_conv(const ConvP& p, int G, hipStream_t st);
int launch_conv_winograd(const WinoP& q, int B, int G, hipStream_t st);
int launch_winograd_weights(const float* w_oihw, int Cout, int Cin, int m, float* u, hipStream_t st);
void winograd_weights_host(const float* w_oihw, int Cout, int Cin, int m, float* u);
bool winograd_eligible(int k, int stride, int pad, int dil, int Cin, int Cout);
bool winograd_m6_channels_ok(int Cin, int Cout);
size_t winograd_ws_floats(int B, int H, int W, int Cin, int Cout, int G, int dil, int m);
double winograd_mac_ratio(int H, int W, int dil, int m);
// single-kernel F(4x4,3x3) (wino_fused.hip)
bool winograd_fused_ok(const WinoP& q, int B, int G);
int launch_conv_winograd_fused(const WinoP& q, int B, int G, hipStream_t st);
void winograd_fused_pack_host(const float* u, int Cout, int Cin, float* uf);
int launch_winograd_fused_pack(const float* u, int Cout, int Cin, float* uf, hipStream_t st);
size_t winograd_fused_ws_floats(int B, int Cin, int G);
int winograd_fused_prepare();

// Tuning state.  Every knob that shapes a plan or changes the arithmetic of a launch lives in the CONTEXT (quber_ctx::tune, set
// with quber_set_option between quber_create and quber_finalize_weights for the plan-time keys, any time for the launch-time ones):
// two engines with different settings coexist in one process and on several threads.  `g_tune` holds the process defaults - what
// quber_set_tuning writes, what quber_create copies, and what the stand-alone test ops (quber_op_*), which have no context, use.
// The launchers read tune(): the tuning of the context whose plan is being built or launched on THIS thread (TuneScope).
struct Tuning {
    int winograd = 0;            // key 6 (plan): Winograd path for the eligible 3x3 layers: 0 = where it pays, 1 = never, 2 = always
    int wino_min_cin = 32;       // key 7 (plan): smallest input width routed to the Winograd path (the 32-channel stem layers take the single-kernel form)
    int wino_max_ratio = 67;     // key 8 (plan): executed / direct multiplies (%) up to which a (dilated) layer takes the Winograd path
    int wino_variant = 0;        // key 9 (plan): output tile edge m of the eligible layers: 0 = automatic (4 or 2), 2, 4, 6 (opt-in)
    int wino_min_cout = 32;      // key 10 (plan): smallest output width routed to the Winograd path
    int wino_pairs = 0;          // key 17 (launch): F(4x4) transforms on channel pairs (8-byte accesses) instead of quads
    int wino_chunk_mb = 0;       // key 20 (launch): largest V | M footprint (MiB) of one pass over a pipeline layer; 0 = the whole batch at once
    int wino_fused = 1;          // key 25 (plan): the eligible F(4x4) layers of the exact fp32 / bf16x3 modes as ONE kernel (wino_fused.hip); 0 = the three-kernel pipeline
    int wino_fused_max_cin = 160;  // key 27 (plan): widest input (channels) the single-kernel form takes.  Its two accumulation chains are Cin / 2 long: 80
                                 //   channels at the default - the float64-anchor ratios stay 0.64-1.11 (0.67-0.98 on the final plan); admitting 256 / 320
                                 //   channels (chains of 128-160) measures 1.20 / 1.21 (DECISIONS.md section 4, profiles/r05_fused_anchor.md)
    int acc_chunk = 2;           // key 21 (launch, arithmetic): K-slices per chunk of the two-level fp32 accumulation (2 = 64 k; 0 = one chain over K)
    int tile_128x64 = 1;         // key 19 (launch): 128x64 tiles for the 33-64 channel layers (0: 64x64)
    int force_tile = 0;          // key 4 (launch, test harness): force the tile shape: 1 = 64x64, 2 = 128x128, 3 = 128x64, 4 = 256x32 (0 = automatic)
    int force_split = 0;         // key 3 (launch, test harness): force the number of K partitions of every convolution with a workspace
    int tail_split = 1;          // key 5 (launch): split the ragged last round of large launches when the model favours it (1), never (0), whenever feasible (2)
    int persist = 1;             // key 13 (launch): persistent convolution launches: 0 = never, 1 = 128x128 tiles, 2 = every tile shape
    int persist_min_nk = 32;     // key 14 (launch): shortest K (in 32-wide slices) whose remainder tiles are shared between blocks
    int persist_min_tiles = 256; // key 15 (launch): fewest tiles (all groups) of a launch that goes persistent
    int persist_debug = 0;       // key 16 (diagnostics): 1 = every store of the persistent epilogue is dropped by the range check
    int stem_fused = 1;          // key 29 (plan): input normalisation + concat (a3) inside the first stem convolution's kernel (csrc/stem.hip); 0 = preprocess kernel + implicit GEMM; 2 = as 1, but the fp16 data path keeps the vector-FMA form (1: its matrix-pipe form)
    int lean_loader = 1;         // key 30 (launch): implicit GEMM with block-uniform filter taps and buffer loads where the layer allows it (conv_igemm.hip LEAN); 0 = per-thread tap arithmetic
    int fuse_shortcut = 1;       // key 18 (plan): conv3 + projection shortcut of a bottleneck as one dual-input GEMM
    int lanes = 1;               // key 24 (launch): side lanes for batches <= 16, exact fp32 / bf16x3 <= 12 (0 = everything on the caller's stream)
    int h8 = 1;                  // key 31 (plan + launch: the plan's ASPP grouping and norm absorption read it too, include/quber_hip.h): fp16 data path: 256 x 256 tiles with the LDS-DMA pipeline for the wide layers (conv_h8.hip); 0 = conv_igemm.hip everywhere
    int x8 = 1;                  // key 35 (launch): bf16x3 mode: the wide 1x1 launches and the Winograd position GEMMs on 256 x 128 tiles with the LDS-DMA pipeline, weights pre-split at plan time
                                 //   (conv_x8.hip); 0 = conv_igemm.hip / conv_persist.hip everywhere, 2 = every covered launch (tests)
    int x8_min_nk = 8;           // key 37 (launch): fewest K-slices (of 32) of a launch that key 35 = 1 takes
    int x8_min_rounds = 2;       // key 36 (launch): fewest rounds of tiles (tiles / CUs) of a launch that key 35 = 1 takes
    int zone_cols = 1;           // key 43 (launch): the dilated layers that skip padded filter rows (ASPP d = 18) skip padded filter columns as well (ConvP::zones)
    int small_n_64 = 1;          // key 42 (launch): 64 x 64 tiles, one per block, for the 1x1 GEMMs (1) with at most 1 280 tiles of 128 x 128, and - exact fp32 - of K <= 1024 whatever their
                                 //   size; 2 = only the first rule; 0 = the 128 x 128 split-K / persistent launches as before round 6 (conv_igemm.hip launch_conv)
    int aspp_lanes = 1;          // key 41 (plan): the dilated ASPP branches d = 6 / 12 on the two side lanes (idle since the fusion convolutions), d = 18 and the 1x1 branch on the caller's
                                 //   stream: 3.74 -> 3.70 ms per batch-1 step (the branches share the chip rather than fill it: each is 430 blocks of short K); 0 = one after the other
    int h8_narrow = 1;           // key 38 (plan + launch, as key 31): fp16 data path: the undilated 3x3 layers with the pixel operand as an LDS patch (conv_h8.hip conv_h8p / h8w / h8s kernels); 0 = the DMA-gather kernels /
                                 //   conv_igemm.hip there, 2 = only the layers of up to 128 output channels
    int h8_norm = 1;             // key 39 (plan): fp16 data path: a patch-kernel layer applies the GroupNorm + ReLU in front of it to its LDS patches (same arithmetic as the norm pass, no pass over
                                 //   the tensor in HBM); 0 = every norm is a pass of its own
    int h8_min_tiles = 224;      // key 32 (launch): fewest tiles (all groups) of a launch that takes it (one block per CU: a launch of fewer tiles leaves CUs idle)
};
extern Tuning g_tune;
extern thread_local const Tuning* t_tune;
inline const Tuning& tune() { return t_tune ? *t_tune : g_tune; }
struct TuneScope {
    const Tuning* prev;
    explicit TuneScope(const Tuning* t) : prev(t_tune) { t_tune = t; }
    ~TuneScope() { t_tune = prev; }
    TuneScope(const TuneScope&) = delete;
    TuneScope& operator=(const TuneScope&) = delete;
};
bool tuning_set(Tuning& t, int key, int value);   // false: not a key of Tuning
// persistent launch of the implicit GEMM (conv_persist.hip); p.mtiles / ntiles / vec_out filled in by the caller
template <int BM, int BN, int WM, int WN> int launch_conv_persistent(ConvP p, int G, int bpc, hipStream_t st);
size_t conv_persistent_ws_floats(int BM, int BN, int bpc);
bool conv_persistent_ok(const ConvP& p);
int conv_persistent_segments(int T, int P, int nk, int min_slices, int bid, int* out4, int cap);
int conv_persistent_fixup(int T, int P, int nk, int min_slices, int xcd, int j, int* tile, int* slots, int cap);
int launch_conv_dual(ConvP p, int G, hipStream_t st);   // 0 done, 1 not covered (run the two convolutions), -1 error
bool conv_h8_patch_takes(const ConvP& p, int G, bool norm);     // conv_h8.hip: this launch (ConvP as launch_conv receives it) runs on a patch kernel [that can normalise its input]
int launch_conv_x8(ConvP p, int G, hipStream_t st);     // bf16x3 1x1 / grouped GEMM, 256 x 128 tiles (conv_x8.hip): 0 done, 1 not covered, -1 error
int launch_split_bf16x3(const float* w, long n, void* planes, hipStream_t st);     // w [n] -> bf16 terms [3][n]
int launch_conv_h8(ConvP p, int G, hipStream_t st);     // fp16 data path, 256 x 256 tiles (conv_h8.hip): 0 done, 1 not covered, -1 error
#ifdef PK_STAMPS
int pk_read_stamps(unsigned long long* dst, int n);
int pk_read_span(unsigned long long* dst, int n);
#endif
int launch_preprocess(const uint8_t* rgb, const uint8_t* depth, const float* offs, const View& x, int B, int Bcap,
                      int H, int W, const float* mean6, const float* std6, int streams, hipStream_t st);
int launch_maxpool3x3s2(const View& in, const View& out, int B, int G, hipStream_t st);
int launch_stem_conv1(const uint8_t* bgr, const uint8_t* depth, const float* offs, int B, int H, int W, int streams, const float* mean6,
                      const float* std6, const float* w, const float* scale, const float* shift, float* out, long out_gs, int es, hipStream_t st,
                      const void* wf16 = nullptr);
int launch_zero(void* p, size_t bytes, hipStream_t st);
int launch_gn_stats(const View& in, int B, int G, int groups, double* stats, hipStream_t st, bool zero = true);
int launch_gn_apply(const View& in, const View& out, int B, int G, int groups, const double* stats,
                    const float* gamma, const float* beta, int param_gs, float eps, int relu, hipStream_t st);
int launch_bilinear(const View& in, const View& out, int B, hipStream_t st);
int launch_avgpool(const View& in, const View& out, int B, hipStream_t st);
// the 1x1 predictors of one hierarchy level (model.py:413-422, 752-759), one launch: per head its features, weights [cout][C], bias,
// first logit plane, optional activation destination (a channel slice of the next level's fusion input) and activation
struct PredHeads { int n; const void* in[5]; const float* w[5]; const float* bias[5]; void* sm[5]; int cout[5], q_ch0[5], act[5] /*0 none, 1 softmax, 2 sigmoid*/; };
int launch_predictors(const PredHeads& hs, int C, int in_cs, int es, int H, int W, float* q, int q_nch, int sm_cs, int B, hipStream_t st);
int launch_copy_channels(const View& in, const View& out, int B, hipStream_t st);
int launch_add_channels(const View& a, const View& b, const View& out, int B, hipStream_t st);
int launch_upsample_logits(const float* q, float* out, int B, int nch, int h, int w, int scale, int OH, int OW,
                           unsigned mul_mask, hipStream_t st);

int launch_normalize_depth(const void* depth, int is_float, long n, double lo, double hi, uint8_t* out3, uint8_t* zero,
                           hipStream_t st);

int inpaint_telea_u8_host(const uint8_t* img, const uint8_t* mask, int H, int W, int radius, uint8_t* out);
int inpaint_depth_u8_host(const uint8_t* depth3, int H, int W, int kernel, uint8_t* out3);
size_t inpaint_depth_ws_bytes(int B, int H, int W);
int launch_inpaint_depth(const uint8_t* depth3, int B, int H, int W, int kernel, void* ws, size_t ws_bytes, uint8_t* out3, hipStream_t st);   // HOST pointers
int launch_resize_u8(const uint8_t* src, int sh, int sw, int ch, uint8_t* dst, int dh, int dw, int linear, hipStream_t st);

// LMFFNet foreground network + post-filter (lmff.hip)
int launch_lmff_preprocess(const uint8_t* bgr, const uint8_t* depth, long pixels, float* x, hipStream_t st);
int launch_dwconv3x3(const View& in, const View& out, int B, int dil, const float* w9, const float* scale,
                     const float* shift, const float* slope, hipStream_t st);
int launch_pool_s2(const View& in, const View& out, int B, int mode /*0 avg 3x3/2/1, 1 max 2x2/2*/, hipStream_t st);
int launch_affine_prelu(const View& a, const View* b, const View& out, int B, const float* scale, const float* shift,
                        const float* slope, hipStream_t st);
int launch_pmca(const View& x, int B, const float* w2x2, const float* fc0, const float* alpha, const float* fc2, float* wts,
                double* sums /* [B][C][5] scratch */, hipStream_t st);
int launch_scale_channels(const View& in, const float* wts, const View& out, int B, hipStream_t st);
int launch_mad_gate(const View& o, const View& att, float* q, int B, int C, hipStream_t st);
int launch_argmax_fg(const float* logits, int B, long HW, int nc, int cls, uint8_t* fg, hipStream_t st);
int launch_mask_overlap(const uint8_t* masks, const uint8_t* fg, int B, int K, long HW, unsigned long long* counts,
                        hipStream_t st);

int launch_boundary_overlap(const int* pred, const int* gt, int H, int W, const int* labels, int n_pred, int n_gt, int radius,
                            void* ws, unsigned* out, hipStream_t st);
size_t boundary_ws_bytes(int H, int W, int n_masks);
int launch_contingency(const int* pred, const int* gt, long n, int cap, void* ws, hipStream_t st);
size_t contingency_ws_bytes(int cap);

int launch_encode(const uint8_t* masks, int B, int N, int H, int W, const float* gauss, int sigma, int legacy_f32,
                  void* ws, float* out, hipStream_t st);
int launch_encode_labels(const int* labels, int B, int N, int H, int W, const float* gauss, int sigma, int legacy_f32, void* ws,
                         float* out, int* bad, hipStream_t st);
size_t encode_ws_bytes(int B, int N, int H, int W);

int launch_errmaps(const uint8_t* init, int N, const uint8_t* gt, int Ng, int B, int Nmax, int H, int W, int d,
                   uint8_t* ws, uint8_t* out, hipStream_t st);
size_t errmaps_ws_bytes(int B, int Nmax, int H, int W);

struct PostCfg {
    float threshold;
    int nms_kernel, top_k, stuff_area, min_area, label_divisor, cap;
};
int launch_postprocess(const float* logits, int nch, int B, int H, int W, const PostCfg& c, void* ws,
                       float* pan, int* count, float* labels, float* scores, float* boxes, int* centers,
                       int* ncenters, hipStream_t st);
size_t postprocess_ws_bytes(int B, int H, int W, int cap);
int launch_extract_masks(const float* pan, const float* labels, int B, int H, int W, int cap, int max_inst,
                         uint8_t* out, hipStream_t st);
int launch_group_pixels(const float* logits, int nch, int B, int H, int W, int cap, const int* centers, const int* ncenters,
                        uint8_t* idmap, unsigned* area, hipStream_t st);

}  // namespace quber

#define QB_CHECK(expr)                                                                         \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess)                                                                 \
            return quber::fail(std::string(#expr) + ": " + hipGetErrorString(e__));            \
    } while (0)
