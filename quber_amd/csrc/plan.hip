// Context, weight ingestion, the network launch plan and the C ABI of libquber_hip.so.
//
// The plan is the MI355X-side equivalent of detectron2's build_model(cfg) for the QuBER refiner:
// it walks the same module tree as
//   maskrefiner/modeling/backbone/resnet.py:358-519   (two ResNet-DeepLab streams + concat fusion)
//   [d2] DeepLabV3PlusHead / ASPP                      (decoder)
//   maskrefiner/modeling/mask_refiner/model.py:711-764 (hierarchical boundary-error -> fg/centre/offset heads)
// but emits a flat list of kernel launches over NHWC buffers.  Design points:
//   * the rgb and depth streams run as ONE grouped launch per layer (blockIdx.z selects the stream),
//     as do the three second-level prediction heads;
//   * every torch.cat of the reference is free: producers write straight into a channel slice of
//     the concatenated buffer (stream outputs, ASPP branches, decoder skip joins, the 164-channel
//     y | feat_b | softmax(pred_b) fusion input);
//   * FrozenBN / eval-BN / bias are folded into the convolution epilogue's per-channel affine;
//   * the head-fusion stack the reference evaluates three times (model.py:760-762) is evaluated once.
#include <math.h>
#include <string.h>

#include <atomic>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/quber_hip.h"
#include "common.h"

namespace quber {

Tuning g_tune;
thread_local const Tuning* t_tune = nullptr;

static int* tuning_field(Tuning& t, int key) {
    switch (key) {
        case 3: return &t.force_split;
        case 4: return &t.force_tile;
        case 5: return &t.tail_split;
        case 6: return &t.winograd;
        case 7: return &t.wino_min_cin;
        case 8: return &t.wino_max_ratio;
        case 9: return &t.wino_variant;
        case 10: return &t.wino_min_cout;
        case 13: return &t.persist;
        case 14: return &t.persist_min_nk;
        case 15: return &t.persist_min_tiles;
        case 16: return &t.persist_debug;
        case 17: return &t.wino_pairs;
        case 18: return &t.fuse_shortcut;
        case 19: return &t.tile_128x64;
        case 20: return &t.wino_chunk_mb;
        case 21: return &t.acc_chunk;
        case 24: return &t.lanes;
        case 25: return &t.wino_fused;
        case 27: return &t.wino_fused_max_cin;
        case 29: return &t.stem_fused;
        case 38: return &t.h8_narrow;
        case 39: return &t.h8_norm;
        case 41: return &t.aspp_lanes;
        case 42: return &t.small_n_64;
        case 43: return &t.zone_cols;
        case 30: return &t.lean_loader;
        case 31: return &t.h8;
        case 32: return &t.h8_min_tiles;
        case 35: return &t.x8;
        case 36: return &t.x8_min_rounds;
        case 37: return &t.x8_min_nk;
        default: return nullptr;
    }
}
bool tuning_set(Tuning& t, int key, int value) {
    int* f = tuning_field(t, key);
    if (f) *f = value;
    return f != nullptr;
}
// keys that shape the plan: they act when quber_finalize_weights builds it and are refused afterwards
bool tuning_plan_time(int key) { return key == 6 || key == 7 || key == 8 || key == 9 || key == 10 || key == 18 || key == 25 || key == 27 || key == 29 || key == 39 || key == 41; }

// compute units of the current device, cached per device id (a process may drive several devices with different counts)
int device_cus() {
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    const bool cached = dev >= 0 && dev < 64;
    if (cached) {
        const int v = cache[dev].load(std::memory_order_relaxed);
        if (v > 0) return v;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (cached) cache[dev].store(cus, std::memory_order_relaxed);
    return cus;
}

static thread_local std::string g_err;
void set_error(const std::string& m) { g_err = m; }
int fail(const std::string& m) {
    g_err = m;
    return -1;
}

// ---- stage profiler (common.h: ProfScope) ----
struct ProfRec { int tag; hipEvent_t e0, e1; double bytes, flops; };
struct StageSum { std::string name; double ms = 0.0, bytes = 0.0, flops = 0.0; int launches = 0; };
struct Profiler {
    std::vector<std::string> tags;
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> pool;
    size_t used = 0;
    std::vector<StageSum> sums;
    hipEvent_t get() {
        if (used == pool.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            pool.push_back(e);
        }
        return pool[used++];
    }
    ~Profiler() {
        for (hipEvent_t e : pool) (void)hipEventDestroy(e);
    }
};
static thread_local Profiler* g_prof = nullptr;

ProfScope::ProfScope(const char* tag, double bytes, double flops, hipStream_t s) : rec(-1), st(s) {
    Profiler* p = g_prof;
    if (!p) return;
    int ti = -1;
    for (size_t i = 0; i < p->tags.size(); ++i)
        if (p->tags[i] == tag) { ti = (int)i; break; }
    if (ti < 0) { ti = (int)p->tags.size(); p->tags.emplace_back(tag); }
    ProfRec r{ti, p->get(), p->get(), bytes, flops};
    if (!r.e0 || !r.e1 || hipEventRecord(r.e0, s) != hipSuccess) return;
    rec = (int)p->recs.size();
    p->recs.push_back(r);
}
ProfScope::~ProfScope() {
    if (rec >= 0 && g_prof) (void)hipEventRecord(g_prof->recs[rec].e1, st);
}

}  // namespace quber

#ifdef WF_STAMPS
namespace quber { int wf_read_stamps(unsigned long long* dst, int n); }
#endif
#ifdef H8_STAMPS
namespace quber { int h8_read_stamps(unsigned long long* dst, int n); }
#endif
#ifdef X8_STAMPS
namespace quber { int x8_read_stamps(unsigned long long* dst, int n); }
#endif
using namespace quber;

constexpr int GN_SLOTS = 64;
static inline size_t gn_slot_doubles(int max_batch) { return (size_t)2 * 32 * max_batch * 4; }

enum OpKind { OP_CONV = 0, OP_NORM = 1, OP_OTHER = 2, OP_KINDS = 3 };
struct Op {
    std::function<int(int, hipStream_t)> run;
    int kind;
    std::string name;   // first weight key (convs / norms) or a short tag
    double flops;       // algorithmic FLOPs at batch 1 (convolutions only)
    int launches;       // kernel launches per run (memsets not counted)
    int lane = 0;       // 0 = the caller's stream; 1.. = a side stream of the context (small batches only, see quber_forward)
    int ctl = 0;        // 1 = fork `lane` here (it may start once the main stream has reached this point), 2 = join it
};
constexpr int LANES = 3;            // side lanes 1, 2: the fusion convolutions of res2 and of res3
constexpr int LANE_BATCH = 16;      // side lanes are used up to this batch (their workspaces are sized for it) ...
constexpr int LANE_BATCH_F32 = 12;  // ... in the fp32-class modes (exact fp32, bf16x3) up to this one.  Same-box A/B of the step with the lanes against one
                                    // stream (profiles/r20_lanes.md): exact fp32 -5 % at 1 frame, -4.3 % at 4, -3 % at 6, -1.5 ... -2.3 % at 8, -0.8 ... -1.3 % at 12,
                                    // 0 ... +0.5 % at 16 (the headline stays on one stream); bf16x3 -3.4 ... -4 % at 8, -2.4 % / +1 % at 16 on two boxes (one stream);
                                    // fp16 data path -11 % at 8, -5.8 ... -6.7 % at 16 (640x480), -1 ... -1.8 % at 1024x1024 x 8.  (Batches <= 2 until round 6's last pass.)

struct quber_ctx {
    quber_config cfg;
    quber::Tuning tune;           // this context's knobs (quber_set_option); starts as a copy of the process defaults
    std::map<std::string, std::vector<float>> hostw;
    std::vector<std::pair<std::string, int64_t>> specs;
    std::vector<void*> allocs;
    size_t alloc_bytes = 0;       // device bytes owned by the context (quber_workspace_bytes)
    int* enc_bad = nullptr;       // out-of-range flag of the label-map encoder
    std::vector<Op> ops;
    std::map<std::string, View> taps;
    float* gauss = nullptr;
    void* enc_ws = nullptr;
    uint8_t* err_ws = nullptr;
    void* post_ws = nullptr;
    double* gn_stats = nullptr;   // [GN_SLOTS][launch group <= 4][max_batch][32 groups][sum, sum of squares]
    int gn_slots = 0;
    float* splitk_ws = nullptr;
    size_t splitk_floats = 0;
    float* wino_ws = nullptr;     // V | M of the Winograd layers (sized for the largest one at max_batch)
    size_t wino_floats = 0;
    // side lanes (batch <= LANE_BATCH): independent branches of the network on streams of their own, each with its own workspaces
    hipStream_t lane_stream[LANES] = {};
    hipEvent_t lane_fork[LANES] = {}, lane_join[LANES] = {};
    float* lane_wino_ws[LANES] = {};
    size_t lane_wino_floats[LANES] = {};
    float* lane_splitk_ws[LANES] = {};
    size_t lane_splitk_floats = 0;
    bool lanes_built = false;     // the plan contains fork / join points
    bool lanes_on = false;        // ... and this forward uses them
    int lane_now = 0;             // lane of the op being launched (0 = the caller's stream): its workspaces are the ones to use
    View X;               // [2][Bmax][H][W][8] (16 channels of fp16 in the fp16 data path)
    float* q = nullptr;   // [Bmax][planes][H/4][W/4]
    const uint8_t* cur_bgr = nullptr;
    const uint8_t* cur_depth = nullptr;
    const float* cur_off = nullptr;
    float* cur_out = nullptr;
    double flops = 0.0;
    double wino_flops = 0.0;      // algorithmic FLOPs (batch 1) of the layers that take the Winograd path
    double wino_saved = 0.0;      // ... and the part of them the path does not execute
    double wino_pad = 0.0;        // executed FLOPs (batch 1) spent on the padding of ragged / short-phase Winograd tiles
    std::vector<hipEvent_t> prof_events;
    std::unique_ptr<quber::Profiler> prof;
    bool stem_fused = false;      // the plan's first op reads the u8 images and the encoding itself: quber_forward launches no preprocess kernel
    bool finalized = false;
    int device = 0;
};

namespace {

constexpr int BLOCKS50[4] = {3, 4, 6, 3}, BLOCKS101[4] = {3, 4, 23, 3}, BLOCKS152[4] = {3, 8, 36, 3};

enum Affine { AF_NONE, AF_FROZEN_BN, AF_BIAS, AF_BIAS_BN };

struct GnFuse { double* sums = nullptr; int groups = 0; };
// a GroupNorm + ReLU whose output has exactly one consumer: if that consumer takes the Winograd path it normalises
// while loading and the separate apply pass is skipped
struct DeferredNorm {
    View in, out;
    const double* stats = nullptr;
    const float *gamma = nullptr, *beta = nullptr;
    int C = 0, G = 0;
    bool absorbed = false;                 // set by the consumer at plan time: it normalises while it loads
};
struct LastConv { std::shared_ptr<GnFuse> fuse; const float* out = nullptr; int G = 0, C = 0; };

static int g_op_wino_reuse = 0;   // key 26

struct Builder {
    quber_ctx* c;
    bool dry;
    LastConv last_conv;
    std::shared_ptr<DeferredNorm> pending_norm;
    std::string err;
    int Bmax, H, W;

    int cur_lane = 0;   // lane of the ops being emitted (0 = main)
    int aes = 4;        // element size of the activation tensors: 2 in the fp16 data path (quber_config.compute_dtype 2)

    Builder(quber_ctx* ctx, bool d) : c(ctx), dry(d), Bmax(ctx->cfg.max_batch), H(ctx->cfg.height), W(ctx->cfg.width) {
        if (ctx->cfg.compute_dtype == 2 && ctx->cfg.with_network == 1) aes = 2;
    }

    // ---- host weights ----
    const float* hw(const std::string& name, int64_t numel) {
        if (dry) {
            c->specs.emplace_back(name, numel);
            return nullptr;
        }
        auto it = c->hostw.find(name);
        if (it == c->hostw.end()) {
            if (err.empty()) err = "missing weight '" + name + "'";
            return nullptr;
        }
        if ((int64_t)it->second.size() != numel) {
            if (err.empty())
                err = "weight '" + name + "' has " + std::to_string(it->second.size()) + " elements, expected " +
                      std::to_string(numel);
            return nullptr;
        }
        return it->second.data();
    }

    // ---- device memory ----
    void* dalloc_bytes(size_t bytes) {
        if (dry) return nullptr;
        void* p = nullptr;
        if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) {
            if (err.empty()) err = "hipMalloc of " + std::to_string(bytes) + " bytes failed";
            return nullptr;
        }
        c->allocs.push_back(p);
        c->alloc_bytes += bytes ? bytes : 16;
        if (hipMemset(p, 0, bytes ? bytes : 16) != hipSuccess && err.empty()) err = "hipMemset of a new buffer failed";
        return p;
    }
    float* upload16(const std::vector<_Float16>& v) {
        float* d = (float*)dalloc_bytes(v.size() * sizeof(_Float16));
        if (d && hipMemcpy(d, v.data(), v.size() * sizeof(_Float16), hipMemcpyHostToDevice) != hipSuccess && err.empty())
            err = "upload of " + std::to_string(v.size()) + " halfs failed";
        return d;
    }
    float* upload(const std::vector<float>& v) {
        float* d = (float*)dalloc_bytes(v.size() * sizeof(float));
        if (d && hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess && err.empty())
            err = "upload of " + std::to_string(v.size()) + " floats failed";
        return d;
    }
    // bf16x3 mode: an uploaded fp32 weight array as three planes of bf16 terms (conv_x8.hip); on the null stream, finalize synchronises
    const void* split3(const float* dev_w, size_t n) {
        if (dry || !dev_w || c->cfg.compute_dtype != 3) return nullptr;
        void* planes = dalloc_bytes(n * 3 * sizeof(unsigned short));
        if (planes && launch_split_bf16x3(dev_w, (long)n, planes, nullptr) && err.empty()) err = "bf16x3 weight split failed";
        return planes;
    }
    View make(int C, int h, int w, int G = 1) {
        View v;
        v.B = Bmax; v.H = h; v.W = w; v.C = C; v.cs = C;
        v.gs = (long)Bmax * h * w * C;
        v.es = aes;
        v.p = (float*)dalloc_bytes((size_t)aes * (size_t)v.gs * G);
        return v;
    }
    static View slice(View v, int coff, int C, long gs = -1) {
        v.p = v.at(coff);
        v.C = C;
        if (gs >= 0) v.gs = gs;
        return v;
    }

    // ---- ops ----
    // Emits one (grouped) convolution launch.  w[g] = OIHW host weights of group g; scale/shift/prelu are
    // [G*Cout] per-channel epilogue vectors (prelu may be empty).
    void emit_conv(const std::string& name, const std::vector<const float*>& w, const View& in, int cin_real,
                   const View& out, int k, int stride, int pad, int dil, bool affine, const std::vector<float>& scale,
                   const std::vector<float>& shift, const std::vector<float>& prelu, const View* res, bool relu,
                   const std::vector<int>& dil_g = {}) {
        const int G = (int)w.size();
        const int Cin = in.C, Cout = out.C;
        const int KS = aes == 2 ? 64 : 32;        // K-slice of the kernel in elements (32 four-byte units: 64 halfs in the fp16 data path)
        const int K = k * k * Cin, Kpad = (K + KS - 1) / KS * KS;
        const int OHp = (in.H + 2 * pad - dil * (k - 1) - 1) / stride + 1;
        // dilated 3x3 whose top / bottom filter rows are padding for >= 20 % of the (row, tap) pairs and that cannot take the
        // Winograd path: tap-major K order so that blocks can skip those rows (conv_igemm.hip MODE 3 / 4)
        const bool skip_rows = k == 3 && stride == 1 && dil > 1 && Cin % KS == 0 && cin_real == Cin && 10 * 2 * pad >= 2 * 3 * OHp &&
                               !(winograd_eligible(k, stride, pad, dil, Cin, out.C) && !res && prelu.empty() &&
                                 std::min(winograd_mac_ratio(in.H, in.W, dil, 4), winograd_mac_ratio(in.H, in.W, dil, 2)) <= tune().wino_max_ratio / 100.0 && tune().winograd != 1);
        const int kmode = (k > 1 && Cin % KS == 0 && !skip_rows) ? 1 : 0;   // slice-major K order for the 3x3 layers
        const int OH = (in.H + 2 * pad - dil * (k - 1) - 1) / stride + 1;
        const int OW = (in.W + 2 * pad - dil * (k - 1) - 1) / stride + 1;
        if (dry) return;
        c->flops += 2.0 * OH * OW * (double)cin_real * k * k * Cout * G;
        if (OH != out.H || OW != out.W) {
            if (err.empty()) err = "internal: conv output geometry mismatch at " + name;
            return;
        }
        for (const float* p : w)
            if (!p) return;   // a missing weight was already reported
        std::vector<float> packed((size_t)G * Cout * Kpad, 0.f);
        for (int g = 0; g < G; ++g)
            for (int o = 0; o < Cout; ++o) {
                float* dst = &packed[((size_t)g * Cout + o) * Kpad];
                for (int ci = 0; ci < cin_real; ++ci)
                    for (int t = 0; t < k * k; ++t) {
                        const size_t kk = kmode ? ((size_t)(ci / KS) * k * k + t) * KS + ci % KS : (size_t)t * Cin + ci;
                        dst[kk] = w[g][((size_t)o * cin_real + ci) * k * k + t];
                    }
            }
        ConvP p{};
        p.in = in.p;
        if (aes == 2) {                          // weights rounded to fp16 once, here
            std::vector<_Float16> ph(packed.size());
            for (size_t i = 0; i < packed.size(); ++i) ph[i] = (_Float16)packed[i];
            p.w = upload16(ph);
        } else {
            p.w = upload(packed);
            if (k == 1) { p.w3 = split3(p.w, packed.size()); p.w3_plane = (long)packed.size(); }
        }
        p.es = aes;
        if (in.es != aes || out.es != aes || (res && res->es != aes)) { if (err.empty()) err = "internal: element type mismatch at " + name; return; }
        // (fp16 data path, layers of >= 32 output channels without an affine - ASPP branches, decoder convolutions, heads: an identity affine, so that conv_h8.hip, whose epilogue
        //  always reads one, takes them; fma(v, 1, 0) == v)
        const bool ident = !affine && aes == 2 && Cout >= 32;
        p.scale = affine ? upload(scale) : ident ? upload(std::vector<float>((size_t)G * Cout, 1.f)) : nullptr;
        p.shift = affine ? upload(shift) : ident ? upload(std::vector<float>((size_t)G * Cout, 0.f)) : nullptr;
        for (size_t g = 0; g < dil_g.size() && g < 4; ++g) p.dil_g[g] = dil_g[g];      // per-group dilation (= padding) of a grouped launch
        p.prelu = prelu.empty() ? nullptr : upload(prelu);
        p.res = res ? res->p : nullptr;
        p.out = out.p;
        p.H = in.H; p.W = in.W; p.Cin = Cin; p.in_cs = in.cs;
        p.OH = OH; p.OW = OW; p.Cout = Cout; p.out_cs = out.cs;
        p.res_cs = res ? res->cs : 0;
        p.K = K; p.Kpad = Kpad;
        p.kh = k; p.kw = k; p.stride = stride; p.pad = pad; p.dil = dil;
        p.relu = relu;
        p.kmode = kmode;
        p.skip_rows = skip_rows;
        p.bf16 = c->cfg.compute_dtype;
        p.in_gs = in.gs; p.out_gs = out.gs; p.res_gs = res ? res->gs : 0;
        p.w_gs = (long)Cout * Kpad; p.ss_gs = Cout;
        p.ohw = OH * OW;
        quber_ctx* ctx = c;
        // Winograd F(m x m,3x3) alternatives for the wide plain 3x3 layers.  The algorithm of a layer is fixed HERE, at plan
        // time, from the layer's geometry alone (frame size, channels, dilation) - never from the batch of a launch - so
        // that a frame's logits do not change class of arithmetic with the batch it arrives in (split-K, a pure
        // re-association of the same fp32 sum, is the only per-launch choice left).
        // Ragged frames and a dilated layer's short phases are padded to whole tiles: a variant qualifies only while it
        // still executes <= tune().wino_max_ratio % of the direct multiplies.  `wq` is the best of m = 4 / 2 (F(4x4) measures the
        // direct kernel's error against float64, profiles/r02a_parity_report.txt).  `wq6`, the 6x6 variant, is OPT-IN
        // (quber_set_tuning key 9 = 6 / QUBER_WINOGRAD=f6): 2.5x the error at tap level, +4.5 % throughput at batch 16.
        WinoP wq{};
        // (the 16-bit operand modes keep every layer on the direct kernel: the Winograd transforms amplify the operands'
        // rounding error; the bf16x3 mode is fp32-equivalent and takes the same plan as the exact fp32 MFMA mode)
        bool wino = winograd_eligible(k, stride, pad, dil, Cin, Cout) && cin_real == Cin && !res && prelu.empty() && tune().winograd != 1 &&
                    (c->cfg.compute_dtype == 0 || c->cfg.compute_dtype == 3);
        if (wino) {
            const double lim = (double)tune().wino_max_ratio / 100.0;
            const double r6 = winograd_m6_channels_ok(Cin, Cout) ? winograd_mac_ratio(in.H, in.W, dil, 6) : 1e9;
            const double r4 = winograd_mac_ratio(in.H, in.W, dil, 4), r2 = winograd_mac_ratio(in.H, in.W, dil, 2);
            double best = lim;
            int wm = 0;
            if (r2 <= best && tune().wino_variant != 4 && tune().wino_variant != 6) { best = r2; wm = 2; }
            if (r4 <= best && tune().wino_variant != 2) { best = r4; wm = 4; }
            if (r2 <= lim && wm == 0) { best = r2; wm = 2; }                     // a forced larger variant does not fit: smaller tiles
            const bool has6 = wm != 0 && tune().wino_variant == 6 && r6 <= 0.9 * best;
            // Maps of a handful of tiles stay on the direct kernel.  The 64-channel layers (res2.conv2) lose to it as three
            // kernels (below 128 channels only the opt-in 6x6 variant outweighs its transforms) but not as ONE: 0.25 against
            // 0.43 ms per layer (wino_fused.hip; profiles/r05_wino_fused_layers.md).
            const bool one_kernel = wm == 4 && tune().wino_fused && (c->cfg.compute_dtype == 0 || c->cfg.compute_dtype == 3) && Cout % 32 == 0 &&
                                    Cin <= tune().wino_fused_max_cin;
            wino = wm != 0 && (tune().winograd == 2 || (Cin < 128 ? (has6 || (one_kernel && (long)in.H * in.W >= 1024)) : (long)in.H * in.W >= 1024));
            if (wino) {
                const int m = has6 ? 6 : wm;
                c->wino_flops += 2.0 * OH * OW * (double)cin_real * k * k * Cout * G;
                c->wino_saved += 2.0 * OH * OW * (double)cin_real * k * k * Cout * G * (1.0 - (has6 ? r6 : best));
                // exactly tiled, F(m x m) executes (m + 2)^2 / (9 m^2) of the direct multiplies: what it executes beyond that is tile padding
                c->wino_pad += 2.0 * OH * OW * (double)cin_real * k * k * Cout * G * ((has6 ? r6 : best) - (double)((m + 2) * (m + 2)) / (9.0 * m * m));
                const int P = (m + 2) * (m + 2);
                std::vector<float> u((size_t)G * P * Cout * Cin);
                for (int g = 0; g < G; ++g) winograd_weights_host(w[g], Cout, Cin, m, &u[(size_t)g * P * Cout * Cin]);
                wq.in = in; wq.out = out; wq.u = upload(u);
                wq.u3 = split3(wq.u, u.size()); wq.u3_plane = (long)u.size();
                if (m == 4 && (c->cfg.compute_dtype == 0 || c->cfg.compute_dtype == 3) && Cout % 32 == 0 && tune().wino_fused && Cin <= tune().wino_fused_max_cin) {     // operand order of the single-kernel form
                    std::vector<float> uf(u.size());
                    for (int g = 0; g < G; ++g) winograd_fused_pack_host(&u[(size_t)g * P * Cout * Cin], Cout, Cin, &uf[(size_t)g * P * Cout * Cin]);
                    wq.uf = upload(uf);
                }
                wq.scale = p.scale; wq.shift = p.shift; wq.ss_gs = Cout; wq.relu = relu; wq.dil = dil; wq.m = m;
                wq.dtype = c->cfg.compute_dtype;
            }
        }
        std::shared_ptr<DeferredNorm> norm, norm16;
        if (wino && pending_norm && pending_norm->out.p == in.p && pending_norm->C == Cin && pending_norm->G == G) {
            norm = pending_norm;
            norm->absorbed = true;
        }
        // fp16 data path: an undilated 3x3 layer applies the GroupNorm + ReLU in front of it to its LDS patches
        // (conv_h8.hip, key 39); whether a launch really does is decided per launch (conv_h8_patch_takes: the launch-time keys may say otherwise)
        float* coef16 = nullptr;
        if (!wino && aes == 2 && tune().h8_norm && tune().h8 && tune().h8_narrow && pending_norm && pending_norm->out.p == in.p && pending_norm->C == Cin &&
            pending_norm->G == G && k == 3 && stride == 1 && pad == 1 && dil == 1 && dil_g.empty() && Cin % 64 == 0 && Cin >= 128 && Cin <= 512 && cin_real == Cin &&
            (Cout == 128 || Cout == 64 || Cout == 32 || (Cout >= 256 && Cout % 8 == 0 && tune().h8_narrow != 2)) && !res && prelu.empty() && in.cs == pending_norm->in.cs && in.gs == pending_norm->in.gs) {
            norm16 = pending_norm;
            norm16->absorbed = true;
            coef16 = (float*)dalloc_bytes(sizeof(float) * (size_t)G * Bmax * Cin * 2);
        }
        pending_norm.reset();
        if (wino) {     // workspace: V | M of the three-kernel pipeline, or only the fused GroupNorm's coefficients of the single-kernel form
            WinoP probe = wq;
            if (norm) probe.in = norm->in;           // what the layer will read (in place -> the pipeline)
            const bool fusedk = wq.uf && winograd_fused_ok(probe, Bmax, G);
            wq.algo = fusedk ? 2 : 1;            // decided once, here, for max_batch: every smaller launch takes the same kernels
            if (wq.uf && winograd_fused_prepare() && err.empty()) err = "winograd (fused): cannot raise the kernels' LDS limit";
            const size_t need = fusedk ? winograd_fused_ws_floats(Bmax, Cin, G) : winograd_ws_floats(Bmax, in.H, in.W, Cin, Cout, G, dil, wq.m);
            if (need > c->wino_floats) c->wino_floats = need;
            if (cur_lane) {
                const int lb = std::min(Bmax, LANE_BATCH);
                const size_t ln = fusedk ? winograd_fused_ws_floats(lb, Cin, G) : winograd_ws_floats(lb, in.H, in.W, Cin, Cout, G, dil, wq.m);
                if (ln > c->lane_wino_floats[cur_lane]) c->lane_wino_floats[cur_lane] = ln;
            }
        }
        auto fuse = std::make_shared<GnFuse>();
        last_conv = {fuse, out.p, G, Cout};
        const int L = cur_lane;
        c->ops.push_back({[p, G, ctx, fuse, wq, wino, norm, norm16, coef16, L](int B, hipStream_t st) mutable {
            const bool side = L && ctx->lane_now == L;             // launched on its side lane: that lane's workspaces
            float* const sk_ws = side ? ctx->lane_splitk_ws[L] : ctx->splitk_ws;
            const size_t sk_floats = side ? ctx->lane_splitk_floats : ctx->splitk_floats;
            if (wino) {
                wq.ws = side ? ctx->lane_wino_ws[L] : ctx->wino_ws; wq.ws_floats = side ? ctx->lane_wino_floats[L] : ctx->wino_floats;
                wq.splitk_ws = sk_ws; wq.splitk_floats = sk_floats;
                wq.gn_sum = fuse->sums; wq.gn_groups = fuse->groups;
                if (norm) {                      // read the producer's pre-normalisation tensor and normalise on load
                    wq.in = norm->in;
                    wq.norm = WinoNorm{norm->stats, norm->gamma, norm->beta, 32, 0, norm->C, 1, 0.0, 1e-5f};
                }
                return launch_conv_winograd(wq, B, G, st);
            }
            p.B = B;
            p.M = B * p.OH * p.OW;
            p.ws = sk_ws;
            p.ws_floats = sk_floats;
            p.gn_sum = fuse->sums;
            p.gn_groups = fuse->groups;
            p.gn_cpg = fuse->groups ? p.Cout / fuse->groups : 0;
            if (norm16) {
                ConvP q = p;
                q.in = norm16->in.p;              // the producer's raw output: normalised on the patch
                q.n_stats = norm16->stats; q.n_gamma = norm16->gamma; q.n_beta = norm16->beta; q.n_coef = coef16;
                q.n_groups = 32; q.n_param_gs = norm16->C; q.n_relu = 1; q.n_eps = 1e-5f;
                if (conv_h8_patch_takes(q, G, true)) return launch_conv(q, G, st);
                // this launch's keys keep it off the patch kernel: the norm as the pass it was, then the convolution on its output
                const int rc = launch_gn_apply(norm16->in, norm16->out, B, G, 32, norm16->stats, norm16->gamma, norm16->beta, norm16->C, 1e-5f, 1, st);
                if (rc) return rc;
            }
            return launch_conv(p, G, st);
        }, OP_CONV, name, 2.0 * OH * OW * (double)cin_real * k * k * Cout * G, 1});
        c->ops.back().lane = cur_lane;
    }

    // refiner convolutions: `names` = one detectron2 Conv2d key prefix per group (e.g. "backbone.rgb_backbone.stem.conv1")
    void conv(const std::vector<std::string>& names, const View& in, int cin_real, const View& out, int k, int stride,
              int pad, int dil, Affine af, const View* res, bool relu, const std::vector<int>& dil_g = {}) {
        const int G = (int)names.size(), Cout = out.C;
        std::vector<float> scale((size_t)G * Cout, 1.f), shift((size_t)G * Cout, 0.f);
        std::vector<const float*> w;
        for (int g = 0; g < G; ++g) {
            const std::string& n = names[g];
            w.push_back(hw(n + ".weight", (int64_t)Cout * cin_real * k * k));
            const float *bias = nullptr, *bw = nullptr, *bb = nullptr, *bm = nullptr, *bv = nullptr;
            if (af == AF_BIAS || af == AF_BIAS_BN) bias = hw(n + ".bias", Cout);
            if (af == AF_FROZEN_BN || af == AF_BIAS_BN) {
                bw = hw(n + ".norm.weight", Cout);
                bb = hw(n + ".norm.bias", Cout);
                bm = hw(n + ".norm.running_mean", Cout);
                bv = hw(n + ".norm.running_var", Cout);
            }
            if (dry) continue;
            for (int o = 0; o < Cout; ++o) {
                float sc = 1.f, sh = 0.f;
                if (af == AF_FROZEN_BN || af == AF_BIAS_BN) {
                    if (!bw || !bb || !bm || !bv) continue;
                    sc = bw[o] * (1.0f / sqrtf(bv[o] + 1e-5f));
                    sh = bb[o] - bm[o] * sc;
                    if (af == AF_BIAS_BN && bias) sh = fmaf(bias[o], sc, sh);
                } else if (af == AF_BIAS && bias) {
                    sh = bias[o];
                }
                scale[(size_t)g * Cout + o] = sc;
                shift[(size_t)g * Cout + o] = sh;
            }
        }
        emit_conv(names[0], w, in, cin_real, out, k, stride, pad, dil, af != AF_NONE, scale, shift, {}, res, relu, dil_g);
    }

    // Projection block of a stage: the two ops emitted last - `shortcut` (1x1, stride s, FrozenBN) and `conv3` (1x1, FrozenBN,
    // + shortcut output, ReLU) - become ONE op that computes relu(bn3(conv3(y)) + bn_s(shortcut(x))) as a single 1x1 GEMM
    // over the concatenated channels of y and x (launch_conv_dual: BN scales folded into the packed weights, shifts
    // added), so that the shortcut's output never exists in HBM.  Launches the dual kernel does not cover (16-bit operand
    // modes, views past 2 GiB) run the two original ops.
    void fuse_shortcut(const std::vector<std::string>& n3, const std::vector<std::string>& ns, const View& y, int mid,
                       const View& x, int cin, int stride, const View& out) {
        if (dry || !tune().fuse_shortcut || c->ops.size() < 2) return;
        const int G = (int)n3.size(), Cout = out.C, Kd = mid + cin;
        const int KS = aes == 2 ? 64 : 32;         // K-slice in elements
        if (mid % KS || cin % KS || y.C != mid || x.C != cin) return;
        std::vector<float> packed((size_t)G * Cout * Kd), ones((size_t)G * Cout, 1.f), shift((size_t)G * Cout);
        for (int g = 0; g < G; ++g) {
            const float* w3 = hw(n3[g] + ".weight", (int64_t)Cout * mid);
            const float* wsh = hw(ns[g] + ".weight", (int64_t)Cout * cin);
            const float* bn[2][4];
            for (int q = 0; q < 2; ++q) {
                const std::string& n = q ? ns[g] : n3[g];
                bn[q][0] = hw(n + ".norm.weight", Cout); bn[q][1] = hw(n + ".norm.bias", Cout);
                bn[q][2] = hw(n + ".norm.running_mean", Cout); bn[q][3] = hw(n + ".norm.running_var", Cout);
            }
            if (!w3 || !wsh) return;
            for (int q = 0; q < 2; ++q)
                for (int e = 0; e < 4; ++e)
                    if (!bn[q][e]) return;
            for (int o = 0; o < Cout; ++o) {
                // the same per-channel affine as conv() derives for the separate launches
                const float s3 = bn[0][0][o] * (1.0f / sqrtf(bn[0][3][o] + 1e-5f)), h3 = bn[0][1][o] - bn[0][2][o] * s3;
                const float ss = bn[1][0][o] * (1.0f / sqrtf(bn[1][3][o] + 1e-5f)), hs = bn[1][1][o] - bn[1][2][o] * ss;
                float* dst = &packed[((size_t)g * Cout + o) * Kd];
                for (int ci = 0; ci < mid; ++ci) dst[ci] = s3 * w3[(size_t)o * mid + ci];
                for (int ci = 0; ci < cin; ++ci) dst[mid + ci] = ss * wsh[(size_t)o * cin + ci];
                shift[(size_t)g * Cout + o] = h3 + hs;
            }
        }
        ConvP p{};
        p.in = y.p; p.in2 = x.p; p.scale = upload(ones); p.shift = upload(shift); p.out = out.p;
        if (aes == 2) {
            std::vector<_Float16> ph(packed.size());
            for (size_t i = 0; i < packed.size(); ++i) ph[i] = (_Float16)packed[i];
            p.w = upload16(ph);
        } else {
            p.w = upload(packed);
            p.w3 = split3(p.w, packed.size()); p.w3_plane = (long)packed.size();     // bf16x3 mode: the three bf16 planes (conv_x8.hip)
        }
        p.es = aes;
        p.H = y.H; p.W = y.W; p.Cin = mid; p.in_cs = y.cs; p.in_gs = y.gs;
        p.H2 = x.H; p.W2 = x.W; p.in2_cs = x.cs; p.in2_gs = x.gs; p.stride2 = stride; p.K1 = mid;
        p.OH = out.H; p.OW = out.W; p.Cout = Cout; p.out_cs = out.cs; p.out_gs = out.gs;
        p.K = Kd; p.Kpad = Kd; p.kh = 1; p.kw = 1; p.stride = 1; p.pad = 0; p.dil = 1; p.relu = 1;
        p.bf16 = c->cfg.compute_dtype;
        p.w_gs = (long)Cout * Kd; p.ss_gs = Cout; p.ohw = out.H * out.W;
        Op op3 = c->ops.back();
        c->ops.pop_back();
        Op ops = c->ops.back();
        c->ops.pop_back();
        quber_ctx* ctx = c;
        c->ops.push_back({[p, G, ctx, op3, ops](int B, hipStream_t st) mutable {
            p.B = B;
            p.M = B * p.OH * p.OW;
            p.ws = ctx->splitk_ws;
            p.ws_floats = ctx->splitk_floats;
            const int rc = launch_conv_dual(p, G, st);
            if (rc != 1) return rc;
            const int r1 = ops.run(B, st);
            return r1 ? r1 : op3.run(B, st);
        }, OP_CONV, op3.name + " + shortcut", op3.flops + ops.flops, 1});
    }

    // GroupNorm(32) + ReLU from `in` into `out` (possibly a concat slice); names = norm key prefixes per group
    void gn_relu(const std::vector<std::string>& names, const View& in, const View& out, bool single_consumer = false) {
        const int G = (int)names.size(), C = in.C;
        std::vector<float> gamma, beta;
        for (int g = 0; g < G; ++g) {
            const float* w = hw(names[g] + ".weight", C);
            const float* b = hw(names[g] + ".bias", C);
            if (dry || !w || !b) continue;
            gamma.insert(gamma.end(), w, w + C);
            beta.insert(beta.end(), b, b + C);
        }
        if (dry) return;
        const float* dg = upload(gamma);
        const float* db = upload(beta);
        // every GroupNorm owns a slot of the sum accumulators; one launch at the start of the forward clears them all
        if (c->gn_slots >= GN_SLOTS) { if (err.empty()) err = "more GroupNorm layers than accumulator slots"; return; }
        double* stats = c->gn_stats + (size_t)c->gn_slots++ * gn_slot_doubles(c->cfg.max_batch);
        // the producer is the convolution emitted just before: it accumulates the sums while it stores its output
        const bool fused = last_conv.fuse && last_conv.out == in.p && last_conv.G == G && last_conv.C == C;
        if (fused) {
            last_conv.fuse->sums = stats;
            last_conv.fuse->groups = 32;
            last_conv.fuse.reset();
        }
        std::shared_ptr<DeferredNorm> dn;
        if (single_consumer && C % 32 == 0 && (C / 32) % 4 == 0) {
            dn = std::make_shared<DeferredNorm>();
            dn->in = in; dn->out = out; dn->stats = stats; dn->gamma = dg; dn->beta = db; dn->C = C; dn->G = G;
        }
        pending_norm = dn;
        c->ops.push_back({[=](int B, hipStream_t st) {
            if (!fused) {
                int rc = launch_gn_stats(in, B, G, 32, stats, st, false);
                if (rc) return rc;
            }
            if (dn && dn->absorbed) return 0;      // the consumer normalises while it loads
            return launch_gn_apply(in, out, B, G, 32, stats, dg, db, C, 1e-5f, 1, st);
        }, OP_NORM, names[0], 0.0, fused ? 1 : 2});
        c->ops.back().lane = cur_lane;
    }

    void op(std::function<int(int, hipStream_t)> f) {
        if (!dry) {
            c->ops.push_back({std::move(f), OP_OTHER, "elementwise", 0.0, 1});
            c->ops.back().lane = cur_lane;
        }
    }
    // Side lanes.  fork(L): the ops emitted until join(L) with cur_lane = L form a branch that depends on nothing emitted after
    // this point and whose results nothing needs before join(L): at small batches, where a launch fills a fraction of the chip,
    // quber_forward runs it on a stream of its own beside what the main stream does meanwhile (the fusion convolutions of res2
    // and res3 beside the later ResNet stages).
    void fork(int L) {
        if (dry) return;
        c->ops.push_back({nullptr, OP_OTHER, "fork", 0.0, 0});
        c->ops.back().lane = L; c->ops.back().ctl = 1;
        cur_lane = L;
        c->lanes_built = true;
    }
    void back_to_main() { cur_lane = 0; }
    void join(int L) {
        if (dry) return;
        c->ops.push_back({nullptr, OP_OTHER, "join", 0.0, 0});
        c->ops.back().lane = L; c->ops.back().ctl = 2;
    }

    // conv (no bias) -> GN -> ReLU, the [d2] Conv2d(norm=GN, activation=relu) pattern
    void conv_gn(const std::string& n, const View& in, const View& tmp, const View& out, int k, int dil,
                 bool single_consumer = false, int cin_real = -1) {
        conv({n}, in, cin_real > 0 ? cin_real : in.C, tmp, k, 1, k == 3 ? dil : 0, dil, AF_NONE, nullptr, false);
        gn_relu({n + ".norm"}, tmp, out, single_consumer);
    }

    // a3 + stem.conv1 as one kernel (csrc/stem.hip): names = the conv's key prefix per stream; out = [NS][Bmax][h2][w2][32]
    void emit_stem_fused(const std::vector<std::string>& names, const View& out) {
        const int G = (int)names.size();
        std::vector<float> packed((size_t)G * 9 * 6 * 32), scale((size_t)G * 32, 1.f), shift((size_t)G * 32, 0.f);
        bool ok = true;
        for (int g = 0; g < G; ++g) {
            const std::string& n = names[g];
            const float* w = hw(n + ".weight", (int64_t)32 * 6 * 9);
            const float* bw = hw(n + ".norm.weight", 32);
            const float* bb = hw(n + ".norm.bias", 32);
            const float* bm = hw(n + ".norm.running_mean", 32);
            const float* bv = hw(n + ".norm.running_var", 32);
            if (dry) continue;
            if (!w || !bw || !bb || !bm || !bv) { ok = false; continue; }
            for (int o = 0; o < 32; ++o) {
                for (int ci = 0; ci < 6; ++ci)
                    for (int t = 0; t < 9; ++t) {
                        const float v = w[((size_t)o * 6 + ci) * 9 + t];
                        packed[(((size_t)g * 9 + t) * 6 + ci) * 32 + o] = out.es == 2 ? (float)(_Float16)v : v;      // fp16 data path: the operand the MFMA kernel multiplies
                    }
                // the same per-channel affine as conv() derives (FrozenBN, [d2]: scale = weight * rsqrt(var + eps))
                const float sc = bw[o] * (1.0f / sqrtf(bv[o] + 1e-5f));
                scale[(size_t)g * 32 + o] = sc;
                shift[(size_t)g * 32 + o] = bb[o] - bm[o] * sc;
            }
        }
        if (dry || !ok) return;
        const int OH = (H + 1) / 2, OW = (W + 1) / 2;
        if (out.C != 32 || out.cs != 32 || out.H != OH || out.W != OW || (out.es != 4 && out.es != 2)) { if (err.empty()) err = "internal: fused stem output geometry"; return; }
        const double fl = 2.0 * OH * OW * 6.0 * 9.0 * 32.0 * G;
        c->flops += fl;
        const float *dw = upload(packed), *ds = upload(scale), *dh = upload(shift);
        // fp16 data path: the filters as fragments of v_mfma_f32_16x16x32_f16 (stem.hip): [stream][tile jj][k-step][lane][8 halfs], lane (fr, fq) =
        // output channel 8 (fr >> 2) + 4 jj + (fr & 3), tap 4 ks + fq, channels 0-5 (6, 7 and taps 9-11: zeros)
        const float* dwf = nullptr;
        if (out.es == 2 && tune().stem_fused != 2) {
            std::vector<_Float16> wf((size_t)G * 2 * 3 * 64 * 8, (_Float16)0.f);
            for (int g = 0; g < G; ++g)
                for (int jj = 0; jj < 2; ++jj)
                    for (int ks = 0; ks < 3; ++ks)
                        for (int l = 0; l < 64; ++l) {
                            const int fr = l & 15, tap = 4 * ks + (l >> 4), n = 8 * (fr >> 2) + 4 * jj + (fr & 3);
                            if (tap < 9)
                                for (int ci = 0; ci < 6; ++ci)
                                    wf[((((size_t)g * 2 + jj) * 3 + ks) * 64 + l) * 8 + ci] = (_Float16)packed[(((size_t)g * 9 + tap) * 6 + ci) * 32 + n];
                        }
            dwf = upload16(wf);
        }
        quber_ctx* ctx = c;
        const View o = out;
        c->stem_fused = true;
        c->ops.push_back({[=](int B, hipStream_t st) {
            return launch_stem_conv1(ctx->cur_bgr, ctx->cur_depth, ctx->cur_off, B, ctx->cfg.height, ctx->cfg.width, G, ctx->cfg.pixel_mean,
                                     ctx->cfg.pixel_std, dw, ds, dh, o.p, o.gs, o.es, st, dwf);
        }, OP_CONV, names[0], fl, 1});
        last_conv = {nullptr, nullptr, 0, 0};
        pending_norm.reset();
    }

    void build() {
        const quber_config& cf = c->cfg;
        const int* nb = cf.resnet_depth == 50 ? BLOCKS50 : cf.resnet_depth == 101 ? BLOCKS101 : BLOCKS152;
        // stride-2 stages round up (3x3/s2/p1 conv and pool: out = floor((in - 1) / 2) + 1; strided 1x1: the same)
        const int h2 = (H + 1) / 2, w2 = (W + 1) / 2, h4 = (h2 + 1) / 2, w4 = (w2 + 1) / 2;
        const int h8 = (h4 + 1) / 2, w8 = (w4 + 1) / 2, h16 = (h8 + 1) / 2, w16 = (w8 + 1) / 2;
        const std::string R = "backbone.rgb_backbone.", D = "backbone.depth_backbone.";
        const int NS = cf.streams;   // 2: rgb + depth streams with concat fusion; 1: a single ResNet (rgb-only / depth-only)
        auto two = [&](const std::string& tail, bool stage_prefix) -> std::vector<std::string> {
            if (NS == 1) return {"backbone." + tail};
            return {R + tail, D + (stage_prefix ? "depth_" : "") + tail};
        };
        if (!dry) {
            c->gn_stats = (double*)dalloc_bytes(sizeof(double) * GN_SLOTS * gn_slot_doubles(Bmax));
            quber_ctx* ctx = c;
            // (on side lane 2, joined where lane 1 is first forked - long before the first kernel that accumulates into the sums: at small
            // batches the stem starts at once instead of behind a 5 us fill)
            fork(2);
            op([ctx](int, hipStream_t st) {
                return launch_zero(ctx->gn_stats, sizeof(double) * ctx->gn_slots * gn_slot_doubles(ctx->cfg.max_batch), st);
            });
            back_to_main();
        }
        if (!dry) {
            c->splitk_floats = (size_t)40 << 20;   // 160 MiB of partial tiles: S x blocks stays near 1-2 rounds of 128x128 tiles at any batch
            c->splitk_ws = (float*)dalloc_bytes(sizeof(float) * c->splitk_floats);
        }

        // ---------------- input + stems (both streams as G = 2) ----------------
        // (fp16 data path: 16 channels - the loader steps through a filter tap in units of 8 four-byte words)
        // fp32 tensors (exact fp32 and bf16x3 modes): a3 runs inside the first convolution's kernel - the normalised 8-channel input
        // (315 MB per 16-frame step) is neither written nor read back (option key 29)
        // fp16 tensors: the same kernel on the fp16-rounded operands (no 16-channel fp16 input tensor, no zero channels multiplied)
        // (the fused kernel is exact fp32 arithmetic with one fold per K-slice - what the implicit GEMM does in the exact and bf16x3 modes and,
        //  on the fp16-rounded operands, in the fp16 data path; bf16 / fp16 OPERANDS on fp32 tensors - compute_dtype 1, or 2 without the
        //  fp16 tensors - keep the preprocess kernel + implicit GEMM, so that option 29 never changes a mode's arithmetic)
        const bool stem_one = tune().stem_fused != 0 && !(aes == 4 && (cf.compute_dtype == 1 || cf.compute_dtype == 2));
        View s1 = make(32, h2, w2, NS), s2 = make(32, h2, w2, NS), s3 = make(64, h2, w2, NS);
        if (stem_one) {
            emit_stem_fused(two("stem.conv1", false), s1);
        } else {
            View X = make(aes == 2 ? 16 : 8, H, W, NS);
            if (!dry) c->X = X;
            conv(two("stem.conv1", false), X, 6, s1, 3, 2, 1, 1, AF_FROZEN_BN, nullptr, true);
        }
        if (!dry) c->taps["stem1"] = s1;
        conv(two("stem.conv2", false), s1, 32, s2, 3, 1, 1, 1, AF_FROZEN_BN, nullptr, true);
        conv(two("stem.conv3", false), s2, 32, s3, 3, 1, 1, 1, AF_FROZEN_BN, nullptr, true);
        View x = make(64, h4, w4, NS);
        op([=](int B, hipStream_t st) { return launch_maxpool3x3s2(s3, x, B, NS, st); });

        // ---------------- res2..res5 ----------------
        View cat[4];  // concatenated [rgb | depth] stage outputs
        // ---------------- backbone fusion (resnet.py:472-485), emitted right after its stage ----------------
        View F[4];
        const int fch[4] = {256, 512, 1024, 2048};
        auto emit_fusion = [&](int s) {
            if (NS == 1) {   // build_resnet_deeplab_fusion_backbone: the stage outputs feed the head directly
                F[s] = cat[s];
                if (!dry) c->taps["res" + std::to_string(s + 2)] = F[s];
                return;
            }
            const std::string n = "backbone.fusion_res" + std::to_string(s + 2) + ".";
            const int C = fch[s], fh = cat[s].H, fw = cat[s].W;
            View t = make(C, fh, fw), a = make(C, fh, fw);
            if (cf.fusion_add) {   // FUSION_STRATEGY "add" (resnet.py:502-503): rgb + depth, no 1x1 reduction
                View ra = slice(cat[s], 0, C), rb = slice(cat[s], C, C);
                op([=](int B, hipStream_t st) { return launch_add_channels(ra, rb, a, B, st); });
            } else {
                conv({n + "conv"}, cat[s], 2 * C, t, 1, 1, 0, 1, AF_BIAS, nullptr, false);
                gn_relu({n + "gn"}, t, a, s != 3 && cf.backbone_fusion_layers > 0);   // read only by conv0 below
            }
            if (s != 3) {
                // a convolution that absorbs the GroupNorm before it reads that norm's INPUT (the previous convolution's raw
                // output): raw outputs alternate between two buffers so that no layer reads the tensor it writes (the
                // single-kernel Winograd layer reads input halos while other blocks store)
                View b2 = make(C, fh, fw), t2 = make(C, fh, fw);
                View cur = a, nxt = b2, traw = t2, tprev = t;
                for (int i = 0; i < cf.backbone_fusion_layers; ++i) {
                    conv({n + "conv" + std::to_string(i)}, cur, C, traw, 3, 1, 1, 1, AF_BIAS, nullptr, false);
                    gn_relu({n + "gn" + std::to_string(i)}, traw, nxt, i + 1 < cf.backbone_fusion_layers);   // read only by the next conv
                    std::swap(cur, nxt);
                    std::swap(traw, tprev);
                }
                a = cur;
            }
            F[s] = a;
            if (!dry) c->taps["res" + std::to_string(s + 2)] = a;
        };
        // decoder inputs that depend on ONE fused stage output only - the 1x1 projections of res3 / res2 (+ GroupNorm) into their slice of the
        // decoder's concatenated buffers - are emitted on that stage's side lane, right behind its fusion convolutions: at small batches
        // they are off the caller's stream altogether (2 x ~30 us per batch-1 forward)
        const std::string Hd = "ins_embed_head.";
        const int CD = cf.convs_dim, HC = cf.head_channels;      // INS_EMBED_HEAD.CONVS_DIM / HEAD_CHANNELS (128 / 32)
        View cat3 = make(64 + 256, h8, w8), t64 = make(64, h8, w8);
        // (fp16 data path: 160 -> 192 channels per pixel, the last 32 never written = zero, zero filters for them: whole 64-channel blocks for the patch kernel)
        View cat2 = make(aes == 2 ? (32 + CD + 63) / 64 * 64 : 32 + CD, h4, w4), t32 = make(32, h4, w4);
        int cin = 64, cout = 256, mid = 64, ch = h4, cw = w4;
        for (int s = 0; s < 4; ++s) {
            const int stage = s + 2;
            const int sdil = stage == 5 ? cf.res5_dilation : 1;
            const int first = (s == 0 || sdil > 1) ? 1 : 2;
            const int oh = first == 2 ? (ch + 1) / 2 : ch, ow = first == 2 ? (cw + 1) / 2 : cw;
            View t1 = make(mid, oh, ow, NS), t2 = make(mid, oh, ow, NS), sc = make(cout, oh, ow, NS);
            View oa = make(cout, oh, ow, NS), ob = make(cout, oh, ow, NS);
            const bool tapped = stage != 4;
            if (tapped) {
                View cb = make(NS * cout, oh, ow, 1);
                cat[s] = cb;
            }
            static const int mg[3] = {1, 2, 4};
            for (int i = 0; i < nb[s]; ++i) {
                const int stride = i == 0 ? first : 1;
                const int dil = stage == 5 ? sdil * mg[i % 3] : 1;
                const std::string tail = "res" + std::to_string(stage) + "." + std::to_string(i) + ".";
                const bool last = i == nb[s] - 1;
                View out = (last && tapped) ? slice(cat[s], 0, cout, cout) : ((i & 1) ? ob : oa);
                conv(two(tail + "conv1", true), x, cin, t1, 1, stride, 0, 1, AF_FROZEN_BN, nullptr, true);
                conv(two(tail + "conv2", true), t1, mid, t2, 3, 1, dil, dil, AF_FROZEN_BN, nullptr, true);
                View resv = x;
                if (cin != cout) {
                    conv(two(tail + "shortcut", true), x, cin, sc, 1, stride, 0, 1, AF_FROZEN_BN, nullptr, false);
                    resv = sc;
                }
                conv(two(tail + "conv3", true), t2, mid, out, 1, 1, 0, 1, AF_FROZEN_BN, &resv, true);
                if (cin != cout) fuse_shortcut(two(tail + "conv3", true), two(tail + "shortcut", true), t2, mid, x, cin, stride, out);
                x = out;
                cin = cout;
            }
            ch = oh; cw = ow;
            cout *= 2; mid *= 2;
            // the fusion convolutions of this stage's output: a side lane for res2 / res3 (they run beside the later stages at small
            // batches and are joined where the decoder first reads them), the main stream for res5 (the ASPP waits for it anyway)
            if (s == 0 || s == 1) {
                if (s == 0) join(2);          // the cleared GroupNorm sums: every accumulating kernel is launched behind this point
                fork(1 + s);
                emit_fusion(s);
                if (s == 1) conv_gn(Hd + "decoder.res3.project_conv", F[1], t64, slice(cat3, 0, 64), 1, 1);
                else conv_gn(Hd + "decoder.res2.project_conv", F[0], t32, slice(cat2, 0, 32), 1, 1);
                back_to_main();
            } else if (s == 3) {
                emit_fusion(3);
            }
        }


        // ---------------- decoder ([d2] DeepLabV3PlusHead.layers) ----------------
        const std::string A = Hd + "decoder.res5.project_conv.";
        View catA = make(1280, h16, w16), tA = make(256, h16, w16);
        // the image-pooling branch (global average -> 1x1 -> broadcast) on side lane 2 (idle since fusion_res3): four latency-bound launches
        // beside the other branches instead of in front of the projection
        fork(2);
        {
            View pooled = make(2048, 1, 1), pc = make(256, 1, 1);
            View f5 = F[3];
            op([=](int B, hipStream_t st) { return launch_avgpool(f5, pooled, B, st); });
            conv({A + "convs.4.1"}, pooled, 2048, pc, 1, 1, 0, 1, AF_BIAS, nullptr, true);
            View dst = slice(catA, 1024, 256);
            op([=](int B, hipStream_t st) { return launch_bilinear(pc, dst, B, st); });
        }
        back_to_main();
        conv_gn(A + "convs.0", F[3], tA, slice(catA, 0, 256), 1, 1);
        const int adil[3] = {6, 12, 18};
        // (round 2 measured the three dilated branches on lanes of their own in the bf16x3 mode: 3.92 against 3.85 ms; round 6, exact fp32, two
        // of them on the lanes that the fusion convolutions have long left: 3.74 -> 3.70 ms - key 41)
        // fp16 data path on maps large enough that no branch skips padded filter rows: the three dilated branches as ONE grouped launch
        // (they read the same tensor; per-group dilation, ConvP::dil_g) - 3 x 128 tiles at 1024x1024 batch 8 instead of three launches
        // that each leave half of conv_h8.hip's one-block-per-CU grid empty
        const int aspp_oh = F[3].H;
        const bool aspp_grouped = aes == 2 && tune().h8 && 10 * 2 * adil[2] < 2 * 3 * aspp_oh;
        if (aspp_grouped) {
            View tA3 = make(256, h16, w16, 3);
            View xin = F[3];
            xin.gs = 0;
            const std::vector<std::string> an = {A + "convs.1", A + "convs.2", A + "convs.3"};
            conv(an, xin, F[3].C, tA3, 3, 1, adil[2], adil[2], AF_NONE, nullptr, false, {adil[0], adil[1], adil[2]});
            gn_relu({an[0] + ".norm", an[1] + ".norm", an[2] + ".norm"}, tA3, slice(catA, 256, 256, 256));
        } else if (tune().aspp_lanes) {
            // key 41: the dilated branches d = 6 / 12 on the two side lanes (temporaries of their own), d = 18 on the caller's stream
            View tA1 = make(256, h16, w16), tA2 = make(256, h16, w16);
            fork(1);
            conv_gn(A + "convs.1", F[3], tA1, slice(catA, 256, 256), 3, adil[0]);
            back_to_main();
            fork(2);
            conv_gn(A + "convs.2", F[3], tA2, slice(catA, 512, 256), 3, adil[1]);
            back_to_main();
            conv_gn(A + "convs.3", F[3], tA, slice(catA, 768, 256), 3, adil[2]);
            join(1);
        } else {
            for (int i = 0; i < 3; ++i) conv_gn(A + "convs." + std::to_string(i + 1), F[3], tA, slice(catA, 256 * (i + 1), 256), 3, adil[i]);
        }
        join(2);                         // fusion_res3 + decoder.res3.project_conv + the pooling branch
        View y5 = make(256, h16, w16);
        conv_gn(A + "project", catA, tA, y5, 1, 1);

        View t128a = make(CD, F[1].H, F[1].W);
        {
            View dst = slice(cat3, 64, 256);
            op([=](int B, hipStream_t st) { return launch_bilinear(y5, dst, B, st); });
        }
        View u3 = make(CD, F[1].H, F[1].W), y3 = make(CD, F[1].H, F[1].W);
        conv_gn(Hd + "decoder.res3.fuse_conv.0", cat3, t128a, u3, 3, 1, true);    // u3 is read only by fuse_conv.1
        View t128b = make(CD, F[1].H, F[1].W);           // (not t128a: fuse_conv.1 reads it - the absorbed norm's input)
        conv_gn(Hd + "decoder.res3.fuse_conv.1", u3, t128b, y3, 3, 1);

        // ---------------- prediction heads: generic hierarchy (model.py:738-762) ----------------
        // head ids: 0 foreground, 1 center, 2 offset, 3 eee_mask, 4 eee_boundary
        static const char* HN[5] = {"foreground", "center", "offset", "eee_mask", "eee_boundary"};
        const int ncls = cf.error_classes;
        const int hch[5] = {1, 1, 2, ncls, ncls};
        const int hplane[5] = {0, 1, 2, QUBER_LOGIT_BASE + (cf.eee_boundary_on ? ncls : 0), QUBER_LOGIT_BASE};
        const bool enabled[5] = {true, true, true, cf.eee_mask_on != 0, cf.eee_boundary_on != 0};
        std::vector<std::vector<int>> levels;
        if (cf.hierarchical) {
            for (int i = 0; i < cf.n_levels; ++i) {
                std::vector<int> l;
                for (int j = 0; j < 5 && cf.level_heads[i][j] >= 0; ++j) l.push_back(cf.level_heads[i][j]);
                levels.push_back(l);
            }
        } else {
            std::vector<int> l;
            for (int k : {3, 4, 0, 1, 2})
                if (enabled[k]) l.push_back(k);
            levels.push_back(l);
        }
        const int nlev = (int)levels.size();
        // concatenated fusion inputs y | feats(prev level) | activations(prev level), one per level >= 1
        std::vector<View> YP(nlev);
        std::vector<int> ypw(nlev, 0);
        for (int i = 1; i < nlev; ++i) {
            int wd = CD;
            if (cf.fusion_feat) wd += HC * (int)levels[i - 1].size();
            if (cf.fusion_pred)
                for (int k : levels[i - 1]) wd += hch[k];
            ypw[i] = wd;
            // whole 16-byte units per pixel; fp16 data path: whole 64-channel K-tiles (164 -> 192: the zero channels meet zero filters), so that the
            // 1x1 reduction in front of the head-fusion stack runs on conv_h8.hip
            YP[i] = make(aes == 2 ? (wd + 63) / 64 * 64 : (wd + 3) / 4 * 4, h4, w4);
        }
        View t128 = make(CD, h4, w4);
        join(1);                         // fusion_res2 + decoder.res2.project_conv
        {
            View dst = slice(cat2, 32, CD);
            op([=](int B, hipStream_t st) { return launch_bilinear(y3, dst, B, st); });
        }
        View u2 = make(CD, h4, w4);
        conv_gn(Hd + "decoder.res2.fuse_conv.0", cat2, t128, u2, 3, 1, true, 32 + CD);      // u2 is read only by fuse_conv.1
        View y = nlev > 1 ? slice(YP[1], 0, CD) : make(CD, h4, w4);
        View t128c = make(CD, h4, w4);                   // (not t128: fuse_conv.1 reads it - the absorbed norm's input)
        conv_gn(Hd + "decoder.res2.fuse_conv.1", u2, t128c, y, 3, 1);
        for (int i = 2; i < nlev; ++i) {
            View dst = slice(YP[i], 0, CD);
            op([=](int B, hipStream_t st) { return launch_copy_channels(y, dst, B, st); });
        }
        if (!dry) c->taps["y"] = y;

        const int planes = QUBER_LOGIT_BASE + ncls * ((cf.eee_mask_on ? 1 : 0) + (cf.eee_boundary_on ? 1 : 0));
        if (!dry) c->q = (float*)dalloc_bytes(sizeof(float) * (size_t)Bmax * planes * h4 * w4);
        float* q = c->q;
        for (int i = 0; i < nlev; ++i) {
            const int G = (int)levels[i].size();
            View x = y;
            if (i > 0) {
                // FusionLayers_i (model.py:424-458), evaluated once (the reference re-runs it per key, model.py:760-762)
                const std::string FL = Hd + "fusion_layers_" + std::to_string(i) + ".fusion_layers.";
                View za = make(CD, h4, w4), zb = make(CD, h4, w4);
                conv({FL + "0"}, YP[i], ypw[i], za, 1, 1, 0, 1, AF_BIAS_BN, nullptr, true);
                View cur = za, nxt = zb;
                for (int j = 0; j < cf.head_fusion_layers; ++j) {
                    conv({FL + std::to_string(j + 1)}, cur, CD, nxt, 3, 1, 1, 1, AF_BIAS_BN, nullptr, true);
                    std::swap(cur, nxt);
                }
                x = cur;
                if (!dry) c->taps["z" + std::to_string(i)] = x;
            }
            std::vector<std::string> h0, h1, n0, n1;
            for (int k : levels[i]) {
                h0.push_back(Hd + HN[k] + "_pred_head.head.0");
                h1.push_back(Hd + HN[k] + "_pred_head.head.1");
                n0.push_back(h0.back() + ".norm");
                n1.push_back(h1.back() + ".norm");
            }
            View g128 = make(CD, h4, w4, G), g128n = make(CD, h4, w4, G), g32 = make(HC, h4, w4, G);
            const bool next = i + 1 < nlev;
            View feat = (next && cf.fusion_feat) ? slice(YP[i + 1], CD, HC, HC) : make(HC, h4, w4, G);
            View xin = x;
            xin.gs = 0;   // every head of the level reads the same features
            conv(h0, xin, CD, g128, 3, 1, 1, 1, AF_NONE, nullptr, false);
            gn_relu(n0, g128, g128n, true);          // g128n is read only by head.1
            conv(h1, g128n, CD, g32, 3, 1, 1, 1, AF_NONE, nullptr, false);
            gn_relu(n1, g32, feat);
            int act_off = CD + (cf.fusion_feat ? HC * G : 0);
            PredHeads ph{};
            int ph_act_cs = 0;
            for (int j = 0; j < G; ++j) {
                const int k = levels[i][j];
                const float* pw = hw(Hd + HN[k] + "_predictor.predictor.weight", (int64_t)hch[k] * HC);
                const float* pb = hw(Hd + HN[k] + "_predictor.predictor.bias", hch[k]);
                View in = feat;
                in.p = feat.at((long)j * feat.gs);
                if (!dry) c->taps[std::string("feat_") + HN[k]] = in;
                float* act_dst = nullptr;
                int act_cs = 0;
                if (next && cf.fusion_pred) {
                    act_dst = YP[i + 1].at(act_off);
                    act_cs = YP[i + 1].cs;
                    act_off += hch[k];
                }
                if (dry || !pw || !pb) continue;
                const float* dw = upload(std::vector<float>(pw, pw + hch[k] * HC));
                const float* db = upload(std::vector<float>(pb, pb + hch[k]));
                ph.in[ph.n] = in.p; ph.w[ph.n] = dw; ph.bias[ph.n] = db; ph.sm[ph.n] = act_dst; ph.cout[ph.n] = hch[k];
                ph.q_ch0[ph.n] = hplane[k]; ph.act[ph.n] = act_dst ? (k >= 3 ? 1 : 2) : 0;
                if (act_dst) ph_act_cs = act_cs;
                ++ph.n;
            }
            if (!dry && ph.n == G) {         // every predictor of the level in one launch
                const int fcs = feat.cs, fes = feat.es, fh = feat.H, fw = feat.W;
                op([=](int B, hipStream_t st) { return launch_predictors(ph, HC, fcs, fes, fh, fw, q, planes, ph_act_cs, B, st); });
            }
        }
        // x4 bilinear of every plane, offsets scaled by the stride (model.py:689-708)
        quber_ctx* ctx = c;
        op([=](int B, hipStream_t st) {
            return launch_upsample_logits(q, ctx->cur_out, B, planes, h4, w4, 4, ctx->cfg.height, ctx->cfg.width, 0xCu, st);
        });
    }
};

}  // namespace

// ------------------------------------------------------------------------------------------------
// LMFFNet launch plan (reference foreground_segmentation/lmffnet.py:283-341; keys = that module's state_dict keys)
namespace {

struct BnP {   // folded BatchNorm(eps 1e-3) + PReLU vectors, padded with identity / zero slope
    std::vector<float> scale, shift, slope;
};

struct LmffBuilder {
    Builder& b;
    quber_ctx* c;
    bool dry;
    explicit LmffBuilder(Builder& bb) : b(bb), c(bb.c), dry(bb.dry) {}

    BnP bnp(const std::string& n, int C, int Cpad = 0) {
        BnP r;
        const float* w = b.hw(n + ".bn.weight", C);
        const float* bi = b.hw(n + ".bn.bias", C);
        const float* m = b.hw(n + ".bn.running_mean", C);
        const float* v = b.hw(n + ".bn.running_var", C);
        const float* a = b.hw(n + ".acti.weight", C);
        if (Cpad < C) Cpad = C;
        r.scale.assign(Cpad, 1.f); r.shift.assign(Cpad, 0.f); r.slope.assign(Cpad, 0.f);
        if (dry || !w || !bi || !m || !v || !a) return r;
        for (int i = 0; i < C; ++i) {
            r.scale[i] = w[i] * (1.0f / sqrtf(v[i] + 1e-3f));
            r.shift[i] = bi[i] - m[i] * r.scale[i];
            r.slope[i] = a[i];
        }
        return r;
    }
    // dense conv (+ optional fused BN+PReLU): key prefix n -> n.conv.weight, n.bn_prelu.*
    void conv(const std::string& n, const View& in, int cin_real, const View& out, int k, int stride, bool fused) {
        const int Cout = out.C;
        const float* w = b.hw(n + ".conv.weight", (int64_t)Cout * cin_real * k * k);
        BnP e;
        if (fused) e = bnp(n + ".bn_prelu", Cout);
        b.emit_conv(n, {w}, in, cin_real, out, k, stride, k == 3 ? 1 : 0, 1, fused, e.scale, e.shift,
                    fused ? e.slope : std::vector<float>(), nullptr, false);
    }
    void dwconv(const std::string& n, const View& in, const View& out, int dil) {
        const int C = in.C;
        const float* w = b.hw(n + ".conv.weight", (int64_t)C * 9);
        BnP e = bnp(n + ".bn_prelu", C);
        if (dry || !w) return;
        const float* dw = b.upload(std::vector<float>(w, w + (size_t)C * 9));
        const float *ds = b.upload(e.scale), *dh = b.upload(e.shift), *dl = b.upload(e.slope);
        b.op([=](int B, hipStream_t st) { return launch_dwconv3x3(in, out, B, dil, dw, ds, dh, dl, st); });
    }
    void affine(const std::string& n, const View& a, const View* add, const View& out) {
        BnP e = bnp(n, a.C);
        if (dry) return;
        const float *ds = b.upload(e.scale), *dh = b.upload(e.shift), *dl = b.upload(e.slope);
        const bool has = add != nullptr;
        const View addv = has ? *add : View();
        b.op([=](int B, hipStream_t st) { return launch_affine_prelu(a, has ? &addv : nullptr, out, B, ds, dh, dl, st); });
    }
    void pool(const View& in, const View& out, int mode) {
        b.op([=](int B, hipStream_t st) { return launch_pool_s2(in, out, B, mode, st); });
    }
    // SEM_B (lmffnet.py:80-113)
    View sem(const std::string& n, const View& x, int dil, const View& out) {
        const int C = x.C, h = x.H, w = x.W;
        View t = b.make(C / 2, h, w), u = b.make(C / 2, h, w), v = b.make(C / 2, h, w), r = b.make(C, h, w);
        conv(n + ".conv3x3", x, C, t, 3, 1, true);
        dwconv(n + ".dconv_left", Builder::slice(t, 0, C / 4), Builder::slice(u, 0, C / 4), 1);
        dwconv(n + ".dconv_right", Builder::slice(t, C / 4, C / 4), Builder::slice(u, C / 4, C / 4), dil);
        conv(n + ".conv3x3_resume.conv3x3", u, C / 2, v, 3, 1, true);
        conv(n + ".conv3x3_resume.conv1x1_resume", v, C / 2, r, 1, 1, false);
        affine(n + ".bn_relu_1", r, &x, out);
        return out;
    }
    // PMCA (lmffnet.py:172-191): channel attention of `x`, written scaled into `out`
    void pmca(const std::string& n, const View& x, const View& out) {
        const int C = x.C;
        const float* w2 = b.hw(n + ".conv2x2.conv.weight", (int64_t)C * 4);
        const float* f0 = b.hw(n + ".SE_Block.fc.0.weight", (int64_t)(C / 8) * C);
        const float* al = b.hw(n + ".SE_Block.fc.1.weight", 1);
        const float* f2 = b.hw(n + ".SE_Block.fc.2.weight", (int64_t)C * (C / 8));
        if (dry || !w2 || !f0 || !al || !f2) return;
        const float* dw2 = b.upload(std::vector<float>(w2, w2 + C * 4));
        const float* df0 = b.upload(std::vector<float>(f0, f0 + (C / 8) * C));
        const float* dal = b.upload(std::vector<float>(al, al + 1));
        const float* df2 = b.upload(std::vector<float>(f2, f2 + C * (C / 8)));
        float* wts = (float*)b.dalloc_bytes(sizeof(float) * (size_t)b.Bmax * C);
        double* sums = (double*)b.dalloc_bytes(sizeof(double) * (size_t)b.Bmax * C * 5);
        b.op([=](int B, hipStream_t st) {
            int rc = launch_pmca(x, B, dw2, df0, dal, df2, wts, sums, st);
            if (rc) return rc;
            return launch_scale_channels(x, wts, out, B, st);
        });
    }

    void build() {
        const int H = b.H, W = b.W, h2 = H / 2, w2 = W / 2, h4 = H / 4, w4 = W / 4, h8 = H / 8, w8 = W / 8;
        const int ncls = 3;
        View X = b.make(8, H, W);
        if (!dry) c->X = X;
        View x6 = Builder::slice(X, 0, 6);
        // Init block
        View i0 = b.make(32, h2, w2), i1 = b.make(32, h2, w2);
        View A = b.make(40, h2, w2);                       // [init(32) | down_1(6) | pad]
        conv("Init_Block.init_conv.0", X, 6, i0, 3, 2, true);
        conv("Init_Block.init_conv.1", i0, 32, i1, 3, 1, true);
        conv("Init_Block.init_conv.2", i1, 32, Builder::slice(A, 0, 32), 3, 1, true);
        View dn1 = Builder::slice(A, 32, 6);
        pool(x6, dn1, 0);
        // FFM-A
        View An = b.make(40, h2, w2), ffa = b.make(40, h2, w2);
        affine("FFM_A.bn_prelu", Builder::slice(A, 0, 38), nullptr, Builder::slice(An, 0, 38));
        conv("FFM_A.conv1x1", An, 38, Builder::slice(ffa, 0, 38), 1, 1, false);
        // downsample 1: conv(38 -> 26) | maxpool(38) -> 64
        View D = b.make(64, h4, w4), d1 = b.make(64, h4, w4);
        conv("downsample_1.conv3x3", ffa, 38, Builder::slice(D, 0, 26), 3, 2, false);
        pool(Builder::slice(ffa, 0, 38), Builder::slice(D, 26, 38), 1);
        affine("downsample_1.bn_prelu", D, nullptr, d1);
        // SEM-B block 1 -> FFM-B1 input [sem(64) | pmca(d1)(64) | down_2(6) | pad]
        View Bc = b.make(136, h4, w4);
        View cur = d1;
        static const int dil1[3] = {2, 2, 2};
        for (int i = 0; i < 3; ++i) {
            View out = i == 2 ? Builder::slice(Bc, 0, 64) : b.make(64, h4, w4);
            cur = sem("SEM_B_Block1.SEM_B_Block.SEM_Block_1" + std::to_string(i), cur, dil1[i], out);
        }
        pmca("FFM_B1.PMCA", d1, Builder::slice(Bc, 64, 64));
        View dn2 = Builder::slice(Bc, 128, 6);
        {
            View tmp = b.make(8, h2, w2);
            View t6 = Builder::slice(tmp, 0, 6);
            pool(x6, t6, 0);
            pool(t6, dn2, 0);
        }
        View Bn = b.make(136, h4, w4), fb1 = b.make(136, h4, w4);
        affine("FFM_B1.bn_prelu", Builder::slice(Bc, 0, 134), nullptr, Builder::slice(Bn, 0, 134));
        conv("FFM_B1.conv1x1", Bn, 134, Builder::slice(fb1, 0, 134), 1, 1, false);
        // downsample 2 (134 -> 128, no concat) + BN/PReLU fused
        View d2 = b.make(128, h8, w8);
        {
            const float* w = b.hw("downsample_2.conv3x3.conv.weight", (int64_t)128 * 134 * 9);
            BnP e = bnp("downsample_2.bn_prelu", 128);
            b.emit_conv("downsample_2.conv3x3", {w}, fb1, 134, d2, 3, 2, 1, 1, true, e.scale, e.shift, e.slope, nullptr, false);
        }
        View Cc = b.make(264, h8, w8);
        cur = d2;
        static const int dil2[8] = {4, 4, 8, 8, 16, 16, 32, 32};
        for (int i = 0; i < 8; ++i) {
            View out = i == 7 ? Builder::slice(Cc, 0, 128) : b.make(128, h8, w8);
            cur = sem("SEM_B_Block2.SEM_B_Block.SEM_Block_2" + std::to_string(i), cur, dil2[i], out);
        }
        pmca("FFM_B2.PMCA", d2, Builder::slice(Cc, 128, 128));
        {
            View t1 = b.make(8, h2, w2), t2 = b.make(8, h4, w4);
            View a6 = Builder::slice(t1, 0, 6), b6 = Builder::slice(t2, 0, 6);
            pool(x6, a6, 0);
            pool(a6, b6, 0);
            pool(b6, Builder::slice(Cc, 256, 6), 0);
        }
        View Cn = b.make(264, h8, w8), fb2 = b.make(264, h8, w8);
        affine("FFM_B2.bn_prelu", Builder::slice(Cc, 0, 262), nullptr, Builder::slice(Cn, 0, 262));
        conv("FFM_B2.conv1x1", Cn, 262, Builder::slice(fb2, 0, 262), 1, 1, false);
        // MAD (lmffnet.py:232-280)
        View cat48 = b.make(48, h4, w4), dl = b.make(32, h8, w8), dwa = b.make(48, h4, w4), att = b.make(4, h4, w4);
        conv("MAD.mid_layer_1x1", fb1, 134, Builder::slice(cat48, 0, 16), 1, 1, false);
        conv("MAD.deep_layer_1x1", fb2, 262, dl, 1, 1, false);
        {
            View dst = Builder::slice(cat48, 16, 32);
            b.op([=](int B, hipStream_t st) { return launch_bilinear(dl, dst, B, st); });
        }
        dwconv("MAD.DwConv1", cat48, dwa, 1);
        conv("MAD.PwConv1", dwa, 48, Builder::slice(att, 0, ncls), 1, 1, false);
        View dwb = b.make(264, h8, w8), o8 = b.make(4, h8, w8), o4 = b.make(4, h4, w4);
        dwconv("MAD.DwConv2", Builder::slice(fb2, 0, 262), Builder::slice(dwb, 0, 262), 1);
        conv("MAD.PwConv2", dwb, 262, Builder::slice(o8, 0, ncls), 1, 1, false);
        b.op([=](int B, hipStream_t st) { return launch_bilinear(o8, o4, B, st); });
        if (!dry) c->q = (float*)b.dalloc_bytes(sizeof(float) * (size_t)b.Bmax * ncls * h4 * w4);
        float* q = c->q;
        quber_ctx* ctx = c;
        b.op([=](int B, hipStream_t st) {
            int rc = launch_mad_gate(o4, att, q, B, ncls, st);
            if (rc) return rc;
            return launch_upsample_logits(q, ctx->cur_out, B, ncls, h4, w4, 4, ctx->cfg.height, ctx->cfg.width, 0u, st);
        });
        if (!dry) {
            c->taps["ffm_a"] = Builder::slice(ffa, 0, 38);
            c->taps["d1"] = d1;
            c->taps["ffm_b1"] = Builder::slice(fb1, 0, 134);
            c->taps["ffm_b2"] = Builder::slice(fb2, 0, 262);
        }
    }
};

int check_cfg(const quber_config& c) {
    if (c.height <= 0 || c.width <= 0) return fail("height and width must be positive");
    if (c.with_network && (c.height < 16 || c.width < 16)) return fail("frames smaller than 16 x 16 are not supported");
    if (c.with_network == 2) {
        if (c.height % 8 || c.width % 8) return fail("LMFFNet needs height and width to be multiples of 8");
        return c.max_batch >= 1 ? 0 : fail("max_batch must be >= 1");
    }
    if (c.max_batch < 1) return fail("max_batch must be >= 1");
    if (c.max_instances < 1) return fail("max_instances must be >= 1");
    if (c.resnet_depth != 50 && c.resnet_depth != 101 && c.resnet_depth != 152) return fail("resnet_depth must be 50, 101 or 152");
    if (c.res5_dilation != 1 && c.res5_dilation != 2 && c.res5_dilation != 4) return fail("res5_dilation must be 1, 2 or 4");
    if (c.res5_dilation == 1) return fail("res5_dilation 1 (output stride 32) is not supported by this build");
    if (c.error_classes < 2 || c.error_classes > 4) return fail("error_classes must be 2..4");
    if (c.streams != 1 && c.streams != 2) return fail("streams must be 1 or 2");
    if (c.with_network == 1 && c.convs_dim != 128 && c.convs_dim != 256) return fail("convs_dim must be 128 or 256");
    if (c.with_network == 1 && c.head_channels != 32 && c.head_channels != 64) return fail("head_channels must be 32 or 64");
    if (c.compute_dtype < 0 || c.compute_dtype > 3)
        return fail("compute_dtype must be 0 (fp32 MFMA), 1 (bf16 operands), 2 (fp16 operands) or 3 (fp32 operands as 3 bf16 terms)");
    if (c.with_network && c.hierarchical) {
        if (c.n_levels < 1 || c.n_levels > 5) return fail("n_levels must be 1..5");
        int seen[5] = {0, 0, 0, 0, 0};
        for (int i = 0; i < c.n_levels; ++i) {
            if (c.level_heads[i][0] < 0) return fail("empty hierarchy level");
            for (int j = 0; j < 5 && c.level_heads[i][j] >= 0; ++j) {
                const int k = c.level_heads[i][j];
                if (k > 4) return fail("hierarchy head id out of range");
                seen[k]++;
            }
        }
        const int want[5] = {1, 1, 1, c.eee_mask_on ? 1 : 0, c.eee_boundary_on ? 1 : 0};
        for (int k = 0; k < 5; ++k)
            if (seen[k] != want[k]) return fail("the hierarchy must list every enabled head exactly once");
    }
    if (c.top_k < 1 || c.top_k > 254) return fail("top_k must be in 1..254");
    if (c.gaussian_sigma < 1 || c.gaussian_sigma > 40) return fail("gaussian_sigma out of range");
    return 0;
}

}  // namespace

static float* g_op_ws = nullptr;
static int g_op_skip_rows = 0;
static int g_op_bf16 = 0;
static const size_t g_op_ws_floats = (size_t)256 << 20;   // 1 GiB, test harness only

extern "C" {

const char* quber_last_error(void) { return quber::g_err.c_str(); }
const char* quber_version(void) { return "quber-hip 0.1 (gfx950, fp32 MFMA)"; }

void quber_default_config(quber_config* c) {
    memset(c, 0, sizeof(*c));
    c->height = 480; c->width = 640; c->max_batch = 1; c->max_instances = 64;
    c->resnet_depth = 50; c->res5_dilation = 2; c->backbone_fusion_layers = 2; c->head_fusion_layers = 3;
    c->error_classes = 4; c->gaussian_sigma = 10; c->nms_kernel = 7; c->top_k = 200; c->stuff_area = 2048;
    c->min_instance_area = 512; c->label_divisor = 1000; c->with_network = 1;
    c->eee_mask_on = 0; c->eee_boundary_on = 1; c->hierarchical = 1; c->fusion_feat = 1; c->fusion_pred = 1;
    c->n_levels = 2;
    c->streams = 2;
    c->fusion_add = 0;
    c->convs_dim = 128; c->head_channels = 32;
    for (int i = 0; i < 5; ++i)
        for (int j = 0; j < 5; ++j) c->level_heads[i][j] = -1;
    c->level_heads[0][0] = 4;                                   // [[eee_boundary], [foreground, center, offset]]
    c->level_heads[1][0] = 0; c->level_heads[1][1] = 1; c->level_heads[1][2] = 2;
    c->center_threshold = 0.3f; c->boundary_ratio = 0.01f;
    const float mean[6] = {103.53f, 116.28f, 123.675f, 127.5f, 127.5f, 127.5f};
    for (int i = 0; i < 6; ++i) { c->pixel_mean[i] = mean[i]; c->pixel_std[i] = 1.f; }
}

int quber_create(const quber_config* cfg, quber_ctx** out) {
    if (!cfg || !out) return fail("null argument");
    if (check_cfg(*cfg)) return -1;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("no HIP device available");
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return fail("hipGetDevice failed");
    quber_ctx* c = new quber_ctx();
    c->cfg = *cfg;
    c->tune = quber::g_tune;          // the process defaults of this moment; quber_set_option changes this context only
    c->device = device;
    const int B = cfg->max_batch, H = cfg->height, W = cfg->width;
    // Gaussian template (predictor.py:246-251): float64 exp rounded to f32
    const int sg = cfg->gaussian_sigma, side = 6 * sg + 3, c0 = 3 * sg + 1;
    std::vector<float> g((size_t)side * side);
    for (int y = 0; y < side; ++y)
        for (int x = 0; x < side; ++x)
            g[(size_t)y * side + x] = (float)exp(-((double)((x - c0) * (x - c0)) + (double)((y - c0) * (y - c0))) / (2.0 * sg * sg));
    Builder b(c, false);
    c->gauss = b.upload(g);
    c->enc_ws = b.dalloc_bytes(encode_ws_bytes(B, cfg->max_instances, H, W));
    c->enc_bad = (int*)b.dalloc_bytes(16);
    c->err_ws = (uint8_t*)b.dalloc_bytes(errmaps_ws_bytes(B, cfg->max_instances > 0 ? cfg->max_instances : 1, H, W));
    c->post_ws = b.dalloc_bytes(postprocess_ws_bytes(B, H, W, cfg->top_k));
    if (!b.err.empty()) {
        std::string e = b.err;
        quber_destroy(c);
        return fail(e);
    }
    if (cfg->with_network == 2) {
        Builder dry(c, true);
        LmffBuilder(dry).build();
    } else if (cfg->with_network) {
        Builder dry(c, true);
        dry.build();
    }
    *out = c;
    return 0;
}

void quber_destroy(quber_ctx* c) {
    if (!c) return;
    if (c->prof && quber::g_prof == c->prof.get()) quber::g_prof = nullptr;
    for (hipEvent_t e : c->prof_events) (void)hipEventDestroy(e);   // nothing useful to do with a failure while tearing down
    for (int l = 1; l < LANES; ++l) {
        if (c->lane_fork[l]) (void)hipEventDestroy(c->lane_fork[l]);
        if (c->lane_join[l]) (void)hipEventDestroy(c->lane_join[l]);
        if (c->lane_stream[l]) (void)hipStreamDestroy(c->lane_stream[l]);
    }
    for (void* p : c->allocs) (void)hipFree(p);
    delete c;
}

int quber_num_weights(quber_ctx* c) { return c ? (int)c->specs.size() : 0; }
int quber_weight_spec(quber_ctx* c, int i, const char** name, int64_t* numel) {
    if (!c || i < 0 || i >= (int)c->specs.size()) return fail("weight index out of range");
    *name = c->specs[i].first.c_str();
    *numel = c->specs[i].second;
    return 0;
}

int quber_set_weight(quber_ctx* c, const char* name, const float* host, int64_t numel) {
    if (!c || !name || !host || numel <= 0) return fail("bad argument to quber_set_weight");
    if (c->finalized) return fail("weights already finalized");
    c->hostw[name].assign(host, host + numel);
    return 0;
}

int quber_finalize_weights(quber_ctx* c) {
    if (!c) return fail("null context");
    if (!c->cfg.with_network) return fail("context was created with with_network = 0");
    if (c->finalized) return fail("weights already finalized");
    quber::TuneScope tscope(&c->tune);          // the plan is shaped by THIS context's options
    Builder b(c, false);
    c->flops = 0.0;
    c->wino_flops = 0.0;
    c->wino_saved = 0.0;
    c->wino_pad = 0.0;
    if (c->cfg.with_network == 2) {
        c->splitk_floats = (size_t)4 << 20;
        c->splitk_ws = (float*)b.dalloc_bytes(sizeof(float) * c->splitk_floats);
        LmffBuilder(b).build();
    } else {
        b.build();
    }
    if (c->wino_floats) c->wino_ws = (float*)b.dalloc_bytes(sizeof(float) * c->wino_floats);
    if (c->lanes_built) {          // side lanes: streams, fork / join events, workspaces sized for LANE_BATCH frames
        c->lane_splitk_floats = c->splitk_floats;       // as large as the caller's stream's (160 MiB): a launch picks the same split on a lane as off it
        for (int l = 1; l < LANES; ++l) {
            QB_CHECK(hipStreamCreateWithFlags(&c->lane_stream[l], hipStreamNonBlocking));
            QB_CHECK(hipEventCreateWithFlags(&c->lane_fork[l], hipEventDisableTiming));
            QB_CHECK(hipEventCreateWithFlags(&c->lane_join[l], hipEventDisableTiming));
            c->lane_splitk_ws[l] = (float*)b.dalloc_bytes(sizeof(float) * c->lane_splitk_floats);
            if (c->lane_wino_floats[l]) c->lane_wino_ws[l] = (float*)b.dalloc_bytes(sizeof(float) * c->lane_wino_floats[l]);
        }
    }
    if (!b.err.empty()) {
        c->ops.clear();
        return fail(b.err);
    }
    QB_CHECK(hipDeviceSynchronize());
    c->hostw.clear();
    c->finalized = true;
    return 0;
}

double quber_forward_flops(quber_ctx* c) { return c ? c->flops : 0.0; }
double quber_forward_flops_executed(quber_ctx* c) {
    if (!c) return 0.0;
    // a layer planned as Winograd F(m x m,3x3) multiplies (m+2)^2 times per m x m output tile and channel pair (padded tiles
    // included) instead of 9 m^2
    return c->flops - c->wino_saved;
}
double quber_forward_flops_padding(quber_ctx* c) { return c ? c->wino_pad : 0.0; }
void quber_set_tuning(int32_t key, int32_t value) {
    // process defaults: copied by every context created afterwards (quber_create) and used by the stand-alone quber_op_* ops;
    // contexts that already exist keep their own settings (quber_set_option)
    if (key == 2) {   // stand-alone conv op: allocate (value != 0) or drop the split-K workspace
        if (value && !g_op_ws) {
            if (hipMalloc((void**)&g_op_ws, sizeof(float) * g_op_ws_floats) != hipSuccess) g_op_ws = nullptr;
        } else if (!value && g_op_ws) {
            (void)hipFree(g_op_ws);
            g_op_ws = nullptr;
        }
        return;
    }
    if (key == 26) { g_op_wino_reuse = value; return; }   // timing harness: quber_op_conv3x3_winograd reuses the transformed filters its previous call left in u / ws
    if (key == 12) { g_op_bf16 = value; return; }         // stand-alone conv ops: 1 = bf16, 2 = fp16 operands, 3 = fp32 as 3 bf16 terms; fp32 accumulation
    if (key == 11) { g_op_skip_rows = value; return; }    // stand-alone conv op: tap-major K order with padded filter rows skipped (dilated 3x3)
    (void)quber::tuning_set(quber::g_tune, key, value);
}

int quber_set_option(quber_ctx* c, int32_t key, int32_t value) {
    if (!c) return fail("null context");
    if (quber::tuning_plan_time(key) && c->finalized) return fail("option " + std::to_string(key) + " shapes the plan: set it before quber_finalize_weights");
    if (!quber::tuning_set(c->tune, key, value)) return fail("unknown option key " + std::to_string(key));
    return 0;
}

int quber_get_option(quber_ctx* c, int32_t key, int32_t* value) {
    if (!c || !value) return fail("null argument");
    int* f = quber::tuning_field(c->tune, key);
    if (!f) return fail("unknown option key " + std::to_string(key));
    *value = *f;
    return 0;
}

#ifdef WF_STAMPS
int quber_wf_read_stamps(unsigned long long* dst, int n) { return quber::wf_read_stamps(dst, n); }
#endif
#ifdef H8_STAMPS
int quber_h8_read_stamps(unsigned long long* dst, int n) { return quber::h8_read_stamps(dst, n); }
#endif
#ifdef X8_STAMPS
int quber_x8_read_stamps(unsigned long long* dst, int n) { return quber::x8_read_stamps(dst, n); }
#endif
#ifdef PK_STAMPS
int quber_pk_read_stamps(unsigned long long* dst, int n) { return quber::pk_read_stamps(dst, n); }
int quber_pk_read_span(unsigned long long* dst, int n) { return quber::pk_read_span(dst, n); }
#endif

int32_t quber_debug_persistent_segments(int32_t tiles, int32_t blocks, int32_t k_slices, int32_t min_share, int32_t block, int32_t* out4,
                                        int32_t cap) {
    return quber::conv_persistent_segments(tiles, blocks, k_slices, min_share, block, out4, cap);
}
int32_t quber_debug_persistent_fixup(int32_t tiles, int32_t blocks, int32_t k_slices, int32_t min_share, int32_t xcd, int32_t j,
                                     int32_t* tile, int32_t* slots, int32_t cap) {
    return quber::conv_persistent_fixup(tiles, blocks, k_slices, min_share, xcd, j, tile, slots, cap);
}

int quber_profile_begin(quber_ctx* c) {
    if (!c) return fail("null context");
    if (!c->prof) c->prof.reset(new quber::Profiler());
    c->prof->recs.clear();
    c->prof->sums.clear();
    c->prof->used = 0;
    quber::g_prof = c->prof.get();
    return 0;
}

int quber_profile_end(quber_ctx* c, void* stream) {
    if (!c || !c->prof || quber::g_prof != c->prof.get()) return fail("quber_profile_end without quber_profile_begin");
    quber::g_prof = nullptr;
    QB_CHECK(hipStreamSynchronize((hipStream_t)stream));
    quber::Profiler& p = *c->prof;
    p.sums.assign(p.tags.size(), quber::StageSum());
    for (size_t i = 0; i < p.tags.size(); ++i) p.sums[i].name = p.tags[i];
    for (const quber::ProfRec& r : p.recs) {
        float ms = 0.f;
        QB_CHECK(hipEventElapsedTime(&ms, r.e0, r.e1));
        quber::StageSum& s = p.sums[r.tag];
        s.ms += ms; s.bytes += r.bytes; s.flops += r.flops; s.launches += 1;
    }
    return 0;
}

int quber_profile_num_stages(quber_ctx* c) { return (c && c->prof) ? (int)c->prof->sums.size() : 0; }

int quber_profile_stage(quber_ctx* c, int i, const char** name, double* ms, double* bytes, double* flops, int32_t* launches) {
    if (!c || !c->prof || i < 0 || i >= (int)c->prof->sums.size()) return fail("stage index out of range");
    const quber::StageSum& s = c->prof->sums[i];
    *name = s.name.c_str(); *ms = s.ms; *bytes = s.bytes; *flops = s.flops; *launches = s.launches;
    return 0;
}

int quber_num_ops(quber_ctx* c) { return c ? (int)c->ops.size() : 0; }
int quber_op_info(quber_ctx* c, int i, const char** name, int32_t* kind, double* flops, int32_t* launches) {
    if (!c || i < 0 || i >= (int)c->ops.size()) return fail("op index out of range");
    *name = c->ops[i].name.c_str();
    *kind = c->ops[i].kind;
    *flops = c->ops[i].flops;
    *launches = c->ops[i].launches;
    return 0;
}

static int check_batch(quber_ctx* c, int batch) {
    if (!c) return fail("null context");
    if (batch < 1 || batch > c->cfg.max_batch) return fail("batch outside 1..max_batch");
    return 0;
}

int quber_encode_initial_masks(quber_ctx* c, const uint8_t* masks, int32_t batch, int32_t n, float* out, void* stream) {
    if (check_batch(c, batch)) return -1;
    if (n > c->cfg.max_instances) return fail("more initial masks than max_instances");
    if (!masks && n > 0) return fail("null masks");
    return launch_encode(masks, batch, n, c->cfg.height, c->cfg.width, c->gauss, c->cfg.gaussian_sigma,
                         c->cfg.encode_legacy_f32, c->enc_ws, out,
                         (hipStream_t)stream);
}

int quber_encode_label_map(quber_ctx* c, const int32_t* labels, int32_t batch, int32_t n, float* out, void* stream) {
    if (check_batch(c, batch)) return -1;
    if (n > c->cfg.max_instances) return fail("more instances than max_instances");
    if (!labels && n > 0) return fail("null label map");
    return launch_encode_labels(labels, batch, n, c->cfg.height, c->cfg.width, c->gauss, c->cfg.gaussian_sigma,
                                c->cfg.encode_legacy_f32, c->enc_ws, out, c->enc_bad, (hipStream_t)stream);
}

int64_t quber_workspace_bytes(quber_ctx* c) { return c ? (int64_t)c->alloc_bytes : 0; }

int quber_explicit_error_maps(quber_ctx* c, const uint8_t* init, int32_t n_init, const uint8_t* gt, int32_t n_gt,
                              int32_t batch, uint8_t* out, void* stream) {
    if (check_batch(c, batch)) return -1;
    const int H = c->cfg.height, W = c->cfg.width;
    // util.py:80-83: dilation = max(1, int(round(ratio * diag)))   (Python round = half to even)
    int d = (int)rint((double)c->cfg.boundary_ratio * sqrt((double)H * H + (double)W * W));
    if (d < 1) d = 1;
    return launch_errmaps(init, n_init, gt, n_gt, batch, c->cfg.max_instances > 0 ? c->cfg.max_instances : 1, H, W, d, c->err_ws,
                          out, (hipStream_t)stream);
}

int quber_forward(quber_ctx* c, const uint8_t* bgr, const uint8_t* depth, const float* offs, int32_t batch,
                  float* logits, void* stream) {
    if (check_batch(c, batch)) return -1;
    if (!c->finalized) return fail("quber_forward before quber_finalize_weights");
    quber::TuneScope tscope(&c->tune);          // every launcher below reads this context's options (tune())
    if (c->cfg.with_network == 2) {   // LMFFNet: (bgr, depth) -> 3 class planes; `offs` is unused
        if (!bgr || !depth || !logits) return fail("null tensor");
        hipStream_t s2 = (hipStream_t)stream;
        c->cur_out = logits;
        int r2 = launch_lmff_preprocess(bgr, depth, (long)batch * c->cfg.height * c->cfg.width, c->X.p, s2);
        for (size_t i = 0; !r2 && i < c->ops.size(); ++i)
            if (!c->ops[i].ctl) r2 = c->ops[i].run(batch, s2);
        return r2;
    }
    if (!bgr || (!depth && c->cfg.streams == 2) || !offs || !logits) return fail("null tensor");
    hipStream_t st = (hipStream_t)stream;
    c->cur_out = logits;
    c->cur_bgr = bgr; c->cur_depth = depth; c->cur_off = offs;
    int rc = c->stem_fused ? 0 : launch_preprocess(bgr, depth, offs, c->X, batch, c->cfg.max_batch, c->cfg.height, c->cfg.width,
                                                   c->cfg.pixel_mean, c->cfg.pixel_std, c->cfg.streams, st);
    if (rc) return rc;
    // side lanes: at small batches the independent branches of the plan (Builder::fork / join) run on streams of the context
    c->lanes_on = c->lanes_built && tune().lanes && batch <= (c->cfg.compute_dtype == 0 || c->cfg.compute_dtype == 3 ? LANE_BATCH_F32 : LANE_BATCH) && c->lane_stream[1] != nullptr &&
                  quber::g_prof == nullptr;
    auto lane_used = [&](int l) { return c->lanes_on && l > 0; };
    for (auto& op : c->ops) {
        if (op.ctl == 1) {
            if (lane_used(op.lane)) {
                QB_CHECK(hipEventRecord(c->lane_fork[op.lane], st));
                QB_CHECK(hipStreamWaitEvent(c->lane_stream[op.lane], c->lane_fork[op.lane], 0));
            }
            continue;
        }
        if (op.ctl == 2) {
            if (lane_used(op.lane)) {
                QB_CHECK(hipEventRecord(c->lane_join[op.lane], c->lane_stream[op.lane]));
                QB_CHECK(hipStreamWaitEvent(st, c->lane_join[op.lane], 0));
            }
            continue;
        }
        c->lane_now = lane_used(op.lane) ? op.lane : 0;
        rc = op.run(batch, c->lane_now ? c->lane_stream[op.lane] : st);
        if (rc) return rc;
    }
    c->lane_now = 0;
    c->lanes_on = false;
    return 0;
}

int quber_forward_profiled(quber_ctx* c, const uint8_t* bgr, const uint8_t* depth, const float* offs, int32_t batch,
                           float* logits, void* stream, double* kind_ms, int32_t* kind_launches) {
    if (check_batch(c, batch)) return -1;
    if (!c->finalized) return fail("quber_forward_profiled before quber_finalize_weights");
    quber::TuneScope tscope(&c->tune);
    if (!bgr || (!depth && c->cfg.streams == 2) || !offs || !logits || !kind_ms || !kind_launches) return fail("null argument");
    hipStream_t st = (hipStream_t)stream;
    c->cur_out = logits;
    const size_t n = c->ops.size();
    if (c->prof_events.size() < 2 * n) {
        const size_t old = c->prof_events.size();
        c->prof_events.resize(2 * n);
        for (size_t i = old; i < 2 * n; ++i) QB_CHECK(hipEventCreate(&c->prof_events[i]));
    }
    c->cur_bgr = bgr; c->cur_depth = depth; c->cur_off = offs;
    int rc = c->stem_fused ? 0 : launch_preprocess(bgr, depth, offs, c->X, batch, c->cfg.max_batch, c->cfg.height, c->cfg.width,
                                                   c->cfg.pixel_mean, c->cfg.pixel_std, c->cfg.streams, st);
    if (rc) return rc;
    c->lanes_on = false;          // one stream: every op in plan order
    for (size_t i = 0; i < n; ++i) {
        QB_CHECK(hipEventRecord(c->prof_events[2 * i], st));
        if (!c->ops[i].ctl) {
            rc = c->ops[i].run(batch, st);
            if (rc) return rc;
        }
        QB_CHECK(hipEventRecord(c->prof_events[2 * i + 1], st));
    }
    QB_CHECK(hipStreamSynchronize(st));
    for (int k = 0; k < OP_KINDS; ++k) { kind_ms[k] = 0.0; kind_launches[k] = 0; }
    for (size_t i = 0; i < n; ++i) {
        float ms = 0.f;
        QB_CHECK(hipEventElapsedTime(&ms, c->prof_events[2 * i], c->prof_events[2 * i + 1]));
        kind_ms[c->ops[i].kind] += ms;
        kind_launches[c->ops[i].kind] += 1;
    }
    return 0;
}

int quber_postprocess(quber_ctx* c, const float* logits, int32_t n_planes, int32_t batch, float* pan, int32_t* count,
                      float* labels, float* scores, float* boxes, int32_t* centers, int32_t* ncenters, void* stream) {
    if (check_batch(c, batch)) return -1;
    PostCfg pc;
    pc.threshold = c->cfg.center_threshold; pc.nms_kernel = c->cfg.nms_kernel; pc.top_k = c->cfg.top_k;
    pc.stuff_area = c->cfg.stuff_area; pc.min_area = c->cfg.min_instance_area; pc.label_divisor = c->cfg.label_divisor;
    pc.cap = c->cfg.top_k;
    return launch_postprocess(logits, n_planes, batch, c->cfg.height, c->cfg.width, pc, c->post_ws, pan, count, labels,
                              scores, boxes, centers, ncenters, (hipStream_t)stream);
}

int quber_extract_masks(quber_ctx* c, const float* pan, const float* labels, int32_t batch, int32_t max_inst,
                        uint8_t* masks, void* stream) {
    if (check_batch(c, batch)) return -1;
    return launch_extract_masks(pan, labels, batch, c->cfg.height, c->cfg.width, c->cfg.top_k, max_inst, masks,
                                (hipStream_t)stream);
}

int64_t quber_contingency_workspace_bytes(int32_t cap) { return (int64_t)contingency_ws_bytes(cap); }

int quber_label_contingency(const int32_t* pred, const int32_t* gt, int64_t n_pixels, int32_t cap, void* workspace,
                            void* stream) {
    if (!pred || !gt || !workspace || n_pixels < 1 || cap < 1 || cap > 1024) return fail("bad argument to quber_label_contingency");
    return launch_contingency(pred, gt, n_pixels, cap, workspace, (hipStream_t)stream);
}

int64_t quber_boundary_workspace_bytes(int32_t h, int32_t w, int32_t n_masks) {
    return (int64_t)boundary_ws_bytes(h, w, n_masks);
}

int quber_boundary_overlap(const int32_t* pred, const int32_t* gt, int32_t h, int32_t w, const int32_t* labels, int32_t n_pred,
                           int32_t n_gt, int32_t bound_pix, void* workspace, uint32_t* out, void* stream) {
    if (!pred || !gt || !labels || !workspace || !out || h < 1 || w < 1) return fail("bad argument to quber_boundary_overlap");
    return launch_boundary_overlap(pred, gt, h, w, labels, n_pred, n_gt, bound_pix, workspace, out, (hipStream_t)stream);
}

int quber_foreground_filter(const float* fg_logits, int32_t n_classes, int32_t fg_class, const uint8_t* masks,
                            int32_t batch, int32_t n_masks, int64_t hw, uint8_t* fg_mask, uint64_t* counts, void* stream) {
    if (!fg_logits || !fg_mask || batch < 1 || hw < 1 || n_classes < 2) return fail("bad argument to quber_foreground_filter");
    hipStream_t st = (hipStream_t)stream;
    int rc = launch_argmax_fg(fg_logits, batch, hw, n_classes, fg_class, fg_mask, st);
    if (rc || n_masks == 0) return rc;
    if (!masks || !counts) return fail("null masks / counts");
    return launch_mask_overlap(masks, fg_mask, batch, n_masks, hw, (unsigned long long*)counts, st);
}

int quber_normalize_depth(const void* depth, int32_t is_float32, int64_t n_pixels, double min_val, double max_val,
                          uint8_t* out3, uint8_t* zero, void* stream) {
    if (!depth || !out3 || n_pixels <= 0) return fail("bad argument to quber_normalize_depth");
    if (!(max_val > min_val)) return fail("normalize_depth: max_val must exceed min_val");
    return launch_normalize_depth(depth, is_float32, n_pixels, min_val, max_val, out3, zero, (hipStream_t)stream);
}

int quber_inpaint_telea_u8(const uint8_t* host_img, const uint8_t* host_mask, int32_t h, int32_t w, int32_t radius,
                           uint8_t* host_out) {
    return inpaint_telea_u8_host(host_img, host_mask, h, w, radius, host_out);
}

int quber_inpaint_depth_u8(const uint8_t* host_depth3, int32_t h, int32_t w, int32_t kernel, uint8_t* host_out3) {
    return inpaint_depth_u8_host(host_depth3, h, w, kernel, host_out3);
}

int64_t quber_inpaint_depth_workspace_bytes(int32_t batch, int32_t h, int32_t w) { return (int64_t)inpaint_depth_ws_bytes(batch, h, w); }

int quber_inpaint_depth_device(const uint8_t* dev_depth3, int32_t batch, int32_t h, int32_t w, int32_t kernel, void* dev_workspace,
                               int64_t workspace_bytes, uint8_t* dev_out3, void* stream) {
    return launch_inpaint_depth(dev_depth3, batch, h, w, kernel, dev_workspace, (size_t)workspace_bytes, dev_out3, (hipStream_t)stream);
}

int quber_resize_u8(const uint8_t* src, int32_t src_h, int32_t src_w, int32_t channels, uint8_t* dst, int32_t dst_h,
                    int32_t dst_w, int32_t linear, void* stream) {
    if (!src || !dst) return fail("bad argument to quber_resize_u8");
    return launch_resize_u8(src, src_h, src_w, channels, dst, dst_h, dst_w, linear, (hipStream_t)stream);
}

int quber_debug_tensor(quber_ctx* c, const char* name, float** ptr, int32_t* dims4, int32_t* cs) {
    if (!c || !name) return fail("null argument");
    auto it = c->taps.find(name);
    if (it == c->taps.end()) return fail(std::string("no intermediate named '") + name + "'");
    *ptr = it->second.p;
    dims4[0] = it->second.B; dims4[1] = it->second.H; dims4[2] = it->second.W; dims4[3] = it->second.C;
    *cs = it->second.cs;
    return 0;
}
int32_t quber_debug_tensor_elem_size(quber_ctx* c, const char* name) {
    if (!c || !name) return 0;
    auto it = c->taps.find(name);
    return it == c->taps.end() ? 0 : it->second.es;
}

// ---------------- stand-alone ops (tests / micro-benchmarks) ----------------
__global__ void pack_oihw_kernel(const float* __restrict__ w, int O, int I, int k, int Kpad, int kmode,
                                 float* __restrict__ out) {
    const long total = (long)O * Kpad;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int o = i / Kpad, kk = i % Kpad;
        float v = 0.f;
        if (kk < k * k * I) {
            int tap, ci;
            if (kmode) {
                const int cb = kk / (k * k * 32), rem = kk % (k * k * 32);
                tap = rem / 32;
                ci = cb * 32 + rem % 32;
            } else {
                tap = kk / I;
                ci = kk % I;
            }
            v = w[((long)o * I + ci) * k * k + tap];
        }
        out[i] = v;
    }
}

int quber_op_conv2d(const float* x, int32_t B, int32_t h, int32_t w, int32_t cin, const float* w_oihw, int32_t cout,
                    int32_t k, int32_t stride, int32_t pad, int32_t dil, const float* scale, const float* shift,
                    const float* residual, int32_t relu, float* packed, float* y, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int K = k * k * cin, Kpad = (K + 31) / 32 * 32;
    const bool skip_rows = g_op_skip_rows && k == 3 && stride == 1 && dil > 1 && cin % 32 == 0;
    const int kmode = (k > 1 && cin % 32 == 0 && !skip_rows) ? 1 : 0;
    hipLaunchKernelGGL(pack_oihw_kernel, dim3(256), dim3(256), 0, st, w_oihw, cout, cin, k, Kpad, kmode, packed);
    ConvP p{};
    p.in = x; p.w = packed; p.scale = scale; p.shift = shift; p.res = residual; p.out = y;
    p.B = B; p.H = h; p.W = w; p.Cin = cin; p.in_cs = cin;
    p.OH = (h + 2 * pad - dil * (k - 1) - 1) / stride + 1;
    p.OW = (w + 2 * pad - dil * (k - 1) - 1) / stride + 1;
    p.Cout = cout; p.out_cs = cout; p.res_cs = cout; p.K = K; p.Kpad = Kpad;
    p.kh = k; p.kw = k; p.stride = stride; p.pad = pad; p.dil = dil; p.relu = relu;
    p.kmode = kmode;
    p.skip_rows = skip_rows;
    p.bf16 = g_op_bf16;
    p.M = B * p.OH * p.OW;
    p.ohw = p.OH * p.OW;
    p.w_gs = 0; p.ss_gs = 0;
    // the stand-alone op splits K only when the test harness asked for a workspace (tuning key 2)
    p.ws = g_op_ws;
    p.ws_floats = g_op_ws ? g_op_ws_floats : 0;
    if (g_op_bf16 == 3 && k == 1 && tune().x8) {       // the pre-split weight planes conv_x8.hip reads (test harness: a grow-only scratch of the process)
        static void* planes = nullptr;
        static size_t cap = 0;
        const size_t need = (size_t)cout * Kpad * 3 * sizeof(unsigned short);
        if (need > cap) {
            if (planes) (void)hipFree(planes);
            planes = nullptr; cap = 0;
            if (hipMalloc(&planes, need) != hipSuccess) return fail("conv2d: cannot allocate the bf16x3 weight planes");
            cap = need;
        }
        const int rc = launch_split_bf16x3(packed, (long)cout * Kpad, planes, st);
        if (rc) return rc;
        p.w3 = planes; p.w3_plane = (long)cout * Kpad;
    }
    return launch_conv(p, 1, st);
}

// the fp16 data path's 1x1 convolution on fp16 tensors (x, w [cout][cin], residual, y: fp16 in HBM; scale / shift fp32)
int quber_op_conv1x1_f16(const void* x, int32_t B, int32_t h, int32_t w, int32_t cin, const void* w_oi, int32_t cout,
                         const float* scale, const float* shift, const void* residual, int32_t relu, void* y, void* stream) {
    if (cin % 64) return fail("conv1x1_f16: cin must be a multiple of 64");
    ConvP p{};
    p.in = (const float*)x; p.w = (const float*)w_oi; p.scale = scale; p.shift = shift; p.res = (const float*)residual; p.out = (float*)y;
    p.B = B; p.H = h; p.W = w; p.Cin = cin; p.in_cs = cin; p.OH = h; p.OW = w;
    p.Cout = cout; p.out_cs = cout; p.res_cs = cout; p.K = cin; p.Kpad = cin;
    p.kh = 1; p.kw = 1; p.stride = 1; p.pad = 0; p.dil = 1; p.relu = relu;
    p.bf16 = 2; p.es = 2;
    p.M = B * h * w; p.ohw = h * w;
    return launch_conv(p, 1, (hipStream_t)stream);
}

int quber_op_conv2d_f16(const void* x, int32_t B, int32_t h, int32_t w, int32_t cin, const void* w_packed, int32_t cout, int32_t ksize,
                        int32_t stride, int32_t pad, int32_t dil, int32_t kmode, const float* scale, const float* shift,
                        const void* residual, int32_t relu, double* gn_sums, int32_t gn_groups, void* y, void* stream) {
    if (cin % 8 || (kmode && cin % 64)) return fail("conv2d_f16: cin must be a multiple of 8 (slice-major K order: of 64)");
    if (ksize < 1 || stride < 1 || dil < 1 || pad < 0) return fail("conv2d_f16: bad geometry");
    ConvP p{};
    p.in = (const float*)x; p.w = (const float*)w_packed; p.scale = scale; p.shift = shift; p.res = (const float*)residual; p.out = (float*)y;
    p.B = B; p.H = h; p.W = w; p.Cin = cin; p.in_cs = cin;
    p.OH = (h + 2 * pad - dil * (ksize - 1) - 1) / stride + 1; p.OW = (w + 2 * pad - dil * (ksize - 1) - 1) / stride + 1;
    if (p.OH < 1 || p.OW < 1) return fail("conv2d_f16: empty output");
    p.Cout = cout; p.out_cs = cout; p.res_cs = cout; p.K = ksize * ksize * cin; p.Kpad = (p.K + 63) / 64 * 64;      // (filter rows zero-filled up to Kpad)
    p.kh = ksize; p.kw = ksize; p.stride = stride; p.pad = pad; p.dil = dil; p.relu = relu;
    p.kmode = kmode;
    p.bf16 = 2; p.es = 2;
    p.M = B * p.OH * p.OW; p.ohw = p.OH * p.OW;
    p.w_gs = (long)cout * p.Kpad; p.ss_gs = cout;
    if (gn_sums) {
        if (gn_groups < 1 || cout % gn_groups) return fail("conv2d_f16: channels must divide into the norm groups");
        p.gn_sum = gn_sums; p.gn_groups = gn_groups; p.gn_cpg = cout / gn_groups;
    }
    return launch_conv(p, 1, (hipStream_t)stream);
}

// y: [B][oh][ow][mid], x: [B][h2][w2][cin] (sampled at `stride`), w: [cout][mid + cin] (BN scales already folded in),
// out = relu?(y . w[:, :mid] + x[::stride, ::stride] . w[:, mid:] + shift)
int quber_op_conv1x1_dual(const float* y, const float* x, int32_t B, int32_t oh, int32_t ow, int32_t mid, int32_t h2, int32_t w2,
                          int32_t cin, int32_t stride, const float* w, const float* shift, const float* ones, int32_t cout,
                          int32_t relu, float* out, void* stream) {
    ConvP p{};
    p.in = y; p.in2 = x; p.w = w; p.scale = ones; p.shift = shift; p.out = out;
    p.B = B; p.H = oh; p.W = ow; p.Cin = mid; p.in_cs = mid;
    p.H2 = h2; p.W2 = w2; p.in2_cs = cin; p.stride2 = stride; p.K1 = mid;
    p.OH = oh; p.OW = ow; p.Cout = cout; p.out_cs = cout;
    p.K = mid + cin; p.Kpad = mid + cin; p.kh = 1; p.kw = 1; p.stride = 1; p.pad = 0; p.dil = 1; p.relu = relu;
    p.bf16 = g_op_bf16;
    p.M = B * oh * ow; p.ohw = oh * ow; p.ss_gs = 0;
    p.ws = g_op_ws; p.ws_floats = g_op_ws ? g_op_ws_floats : 0;
    if (g_op_bf16 == 3 && tune().x8) {       // the pre-split weight planes conv_x8.hip reads (as quber_op_conv2d: a grow-only scratch of the process)
        static void* planes = nullptr;
        static size_t cap = 0;
        const size_t need = (size_t)cout * p.Kpad * 3 * sizeof(unsigned short);
        if (need > cap) {
            if (planes) (void)hipFree(planes);
            planes = nullptr; cap = 0;
            if (hipMalloc(&planes, need) != hipSuccess) return fail("conv1x1_dual: cannot allocate the bf16x3 weight planes");
            cap = need;
        }
        const int rs = launch_split_bf16x3(w, (long)cout * p.Kpad, planes, (hipStream_t)stream);
        if (rs) return rs;
        p.w3 = planes; p.w3_plane = (long)cout * p.Kpad;
    }
    const int rc = launch_conv_dual(p, 1, (hipStream_t)stream);
    if (rc == 1) return fail("conv1x1_dual: launch not covered by the dual kernel (workspace: tuning key 2)");
    return rc;
}

static View mkview(const float* p, int B, int h, int w, int c);

int quber_op_conv3x3_winograd(const float* x, int32_t B, int32_t h, int32_t w, int32_t cin, const float* w_oihw, int32_t cout,
                              int32_t dil, int32_t m, const float* scale, const float* shift, int32_t relu, float* u, float* ws,
                              int64_t ws_floats, float* y, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!winograd_eligible(3, 1, dil, dil, cin, cout)) return fail("winograd: unsupported channel counts");
    if ((scale == nullptr) != (shift == nullptr)) return fail("winograd: scale and shift go together");
    if (m != 2 && m != 4 && m != 6) return fail("winograd: the output tile edge is 2, 4 or 6");
    int rc = g_op_wino_reuse ? 0 : launch_winograd_weights(w_oihw, cout, cin, m, u, st);
    if (rc) return rc;
    WinoP q{};
    q.in = mkview(x, B, h, w, cin); q.out = mkview(y, B, h, w, cout);
    q.u = u; q.scale = scale; q.shift = shift; q.ss_gs = 0; q.relu = relu; q.dil = dil; q.m = m;
    q.dtype = g_op_bf16;
    q.ws = ws; q.ws_floats = (size_t)ws_floats;
    // the single-kernel form where it applies and the workspace also holds its filter order (36 * cout * cin floats)
    if (m == 4 && tune().wino_fused && cout % 32 == 0 && (size_t)ws_floats >= (size_t)36 * cout * cin) {
        q.uf = ws;
        if (winograd_fused_ok(q, B, 1)) {
            rc = winograd_fused_prepare();
            if (rc) return rc;
            rc = g_op_wino_reuse ? 0 : launch_winograd_fused_pack(u, cout, cin, ws, st);
            if (rc) return rc;
            q.ws = ws + (size_t)36 * cout * cin; q.ws_floats = (size_t)ws_floats - (size_t)36 * cout * cin;
        } else {
            q.uf = nullptr;
        }
    }
    q.splitk_ws = g_op_ws; q.splitk_floats = g_op_ws ? g_op_ws_floats : 0;
    return launch_conv_winograd(q, B, 1, st);
}

static View mkview(const float* p, int B, int h, int w, int c) {
    View v;
    v.p = const_cast<float*>(p); v.B = B; v.H = h; v.W = w; v.C = c; v.cs = c; v.gs = 0;
    return v;
}

int quber_op_groupnorm(const float* x, int32_t B, int32_t h, int32_t w, int32_t c, int32_t groups, const float* gamma,
                       const float* beta, float eps, int32_t relu, double* stats, float* y, void* stream) {
    View in = mkview(x, B, h, w, c), out = mkview(y, B, h, w, c);
    int rc = launch_gn_stats(in, B, 1, groups, stats, (hipStream_t)stream);
    if (rc) return rc;
    return launch_gn_apply(in, out, B, 1, groups, stats, gamma, beta, 0, eps, relu, (hipStream_t)stream);
}

int quber_op_bilinear(const float* x, int32_t B, int32_t h, int32_t w, int32_t c, int32_t oh, int32_t ow, float* y,
                      void* stream) {
    if (c % 4) return fail("bilinear: channels must be a multiple of 4");
    return launch_bilinear(mkview(x, B, h, w, c), mkview(y, B, oh, ow, c), B, (hipStream_t)stream);
}

int quber_op_group_pixels(const float* logits, int32_t n_planes, int32_t batch, int32_t h, int32_t w, int32_t cap,
                          const int32_t* centers, const int32_t* ncenters, uint8_t* ids, uint32_t* area, void* stream) {
    if (!logits || !centers || !ncenters || !ids || !area || batch < 1 || h < 1 || w < 1) return fail("bad argument to quber_op_group_pixels");
    return launch_group_pixels(logits, n_planes, batch, h, w, cap, centers, ncenters, ids, area, (hipStream_t)stream);
}

int quber_op_maxpool3x3s2(const float* x, int32_t B, int32_t h, int32_t w, int32_t c, float* y, void* stream) {
    if (c % 4) return fail("maxpool: channels must be a multiple of 4");
    return launch_maxpool3x3s2(mkview(x, B, h, w, c), mkview(y, B, (h + 1) / 2, (w + 1) / 2, c), B, 1, (hipStream_t)stream);
}

}  // extern "C"
