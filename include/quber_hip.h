/* libquber_hip.so - C ABI of the MI355X (gfx950) QuBER mask-refinement hot path.
 *
 * The reference (gist-ailab/QuBER) has no native interface on this path: everything below the Python
 * predictor is stock torch ops.  The entry points therefore replace *Python call sites*; each one cites the
 * reference code it stands in for.  All `const T* dev_*` / `T* dev_*` arguments are DEVICE pointers
 * (e.g. torch.Tensor.data_ptr() of a ROCm tensor); every hot call is asynchronous on `stream`
 * (a hipStream_t passed as void*), allocates nothing, and never synchronises.  Return value: 0 on
 * success, negative on error with the text available from quber_last_error() (thread-local).
 * One context per GPU / stream; a context is not re-entrant.
 */
#ifndef QUBER_HIP_H
#define QUBER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct quber_ctx quber_ctx;

/* Architecture + post-processing constants; defaults mirror
 * configs/uoais-sim/instance-segmentation/seed77/mask-refiner-rgbd-concat-l2-gn-hf-b-fco-l3-b8.yaml on top of
 * Base-Mask-Refiner.yaml and maskrefiner/config.py:6-102 (see quber_default_config). */
typedef struct quber_config {
    int32_t height, width;           /* frame size; multiples of 16 */
    int32_t max_batch;               /* frames per call the workspace is sized for */
    int32_t max_instances;           /* initial masks per frame a call may carry (any number; > 254 are encoded in chunks) */
    int32_t resnet_depth;            /* MODEL.RESNETS.DEPTH: 50 | 101 | 152 */
    int32_t res5_dilation;           /* MODEL.RESNETS.RES5_DILATION (2) */
    int32_t backbone_fusion_layers;  /* MODEL.BACKBONE.NUM_FUSION_LAYERS (2) */
    int32_t head_fusion_layers;      /* MODEL.INS_EMBED_HEAD.NUM_FUSION_LAYERS (3) */
    int32_t error_classes;           /* ERROR_TYPE e3 -> 4 */
    int32_t gaussian_sigma;          /* predictor.py:246 (10) */
    int32_t nms_kernel;              /* PANOPTIC_DEEPLAB.NMS_KERNEL (7) */
    int32_t top_k;                   /* PANOPTIC_DEEPLAB.TOP_K_INSTANCE (200) */
    int32_t stuff_area;              /* PANOPTIC_DEEPLAB.STUFF_AREA (2048) */
    int32_t min_instance_area;       /* post_processing.py:145 (512) */
    int32_t label_divisor;           /* register_uoais_sim_panoptic.py:177-186 (1000) */
    int32_t with_network;            /* 0: only the encode / error-map / post-processing kernels are usable;
                                        1: the refiner network; 2: the LMFFNet foreground network of the post-filter
                                           (foreground_segmentation/lmffnet.py; quber_forward then maps (bgr, depth) to
                                           3 class planes [B][3][H][W], dev_offsets is ignored) */
    float center_threshold;          /* PANOPTIC_DEEPLAB.CENTER_THRESHOLD (0.3) */
    float boundary_ratio;            /* explicit_error_estimation/util.py:92 dilation_ratio (0.01) */
    float pixel_mean[6];             /* MODEL.PIXEL_MEAN */
    float pixel_std[6];              /* MODEL.PIXEL_STD */
    /* prediction-head wiring (maskrefiner/modeling/mask_refiner/model.py:545-608, 738-762).
     * head ids: 0 foreground, 1 center, 2 offset, 3 eee_mask, 4 eee_boundary */
    int32_t eee_mask_on;             /* INS_EMBED_HEAD.EEE_MASK_ON */
    int32_t eee_boundary_on;         /* INS_EMBED_HEAD.EEE_BOUNDARY_ON */
    int32_t hierarchical;            /* INS_EMBED_HEAD.HIERARCHICAL_FUSION_ON */
    int32_t fusion_feat;             /* "feat" in INS_EMBED_HEAD.FUSION_TARGET */
    int32_t fusion_pred;             /* "pred" in INS_EMBED_HEAD.FUSION_TARGET */
    int32_t n_levels;                /* len(INS_EMBED_HEAD.HIERARCHY) */
    int32_t level_heads[5][5];       /* head ids per level, -1 padded */
    int32_t fusion_add;              /* MODEL.BACKBONE.FUSION_STRATEGY == "add" (0 = "concat") */
    int32_t streams;                 /* 2: rgb + depth streams (build_resnet_deeplab_rgbd_fusion_backbone); 1: single stream
                                        (build_resnet_deeplab_fusion_backbone: rgb-only or depth-only, pixel_mean[0..2]) */
    int32_t compute_dtype;           /* arithmetic of the convolutions: 0 = exact fp32 MFMA (default; the 1e-4 parity bar),
                                        3 = fp32-equivalent on the bf16 matrix pipe: every fp32 operand split into 3 bf16 terms,
                                        6 exact partial products per multiply, fp32 accumulation (dropped terms < 2^-26; same
                                        plan, same 1e-4 bar, 6/16 of the matrix time),
                                        2 = the fp16 data path (BASELINE.json configs[4] "fp16 MFMA path" stand-in): activations and
                                        packed weights fp16 in device memory, fp32 accumulation, fp32 norm statistics and logits;
                                        own tolerance (tests/test_gpu_loud_parity.py HALF_TOL),
                                        1 = bf16 operands on fp32 tensors (8x coarser) */
    int32_t encode_legacy_f32;       /* a1 offset arithmetic (predictor.py:345-346, `np.float64 scalar - float32 array`):
                                        0 = numpy >= 2 promotion (float64, rounded once; what the reference computes when run
                                        under this image's numpy 2.2, pinned by tests/golden/encode_*.npz);
                                        1 = numpy < 2 value-based casting (all float32; the reference's pinned numpy==1.23.1,
                                        INSTALL.md:14) - differs by 1 ulp on about a third of the mask pixels */
    int32_t convs_dim;               /* INS_EMBED_HEAD.CONVS_DIM (128): channels of the decoder's res3 / res2 stages, of the head
                                        convolutions and of the head-fusion stacks (model.py:610-651 decoder_channels); 128 | 256 */
    int32_t head_channels;           /* INS_EMBED_HEAD.HEAD_CHANNELS (32): channels in front of every 1x1 predictor
                                        (model.py:514-531, 413-422); 32 | 64 */
} quber_config;

/* logit planes produced by quber_forward: [fg, centre, off_y, off_x, eee_boundary x classes (if on), eee_mask x classes (if on)] */
#define QUBER_LOGIT_BASE 4

void quber_default_config(quber_config* cfg);
const char* quber_last_error(void);
const char* quber_version(void);

/* Build a context on the current HIP device.  Replaces MaskRefinerPredictor.__init__'s build_model(cfg)
 * (maskrefiner/predictor.py:209-229) minus its dataset / output-dir side effects. */
int quber_create(const quber_config* cfg, quber_ctx** out);
void quber_destroy(quber_ctx* ctx);

/* Weights: one call per state_dict entry, `name` being the detectron2-compatible key, `host`
 * a HOST pointer to `numel` contiguous fp32 values in torch's layout (OIHW for convs).  Replaces
 * DetectionCheckpointer.load (predictor.py:228-229).  quber_finalize_weights folds the frozen / eval
 * batch-norms into per-channel affines, re-lays the filters out for the implicit-GEMM kernel, uploads them and
 * fails (listing the first missing key) if the architecture needs a tensor that was not supplied. */
int quber_set_weight(quber_ctx* ctx, const char* name, const float* host, int64_t numel);
int quber_finalize_weights(quber_ctx* ctx);
/* number of tensors the configured architecture expects, and the i-th name / element count */
int quber_num_weights(quber_ctx* ctx);
int quber_weight_spec(quber_ctx* ctx, int index, const char** name, int64_t* numel);

/* a1 - initial masks -> (centre heat-map, off_y, off_x).  Replaces the numpy loop of
 * MaskRefinerPredictor.predict (maskrefiner/predictor.py:304-357).
 *   dev_masks u8 [B][N][H][W] (non-zero = inside)  ->  dev_out f32 [B][3][H][W] */
int quber_encode_initial_masks(quber_ctx* ctx, const uint8_t* dev_masks, int32_t batch, int32_t n_masks,
                               float* dev_out, void* stream);

/* a1 from a label map: dev_labels i32 [B][H][W] with values 0 (no instance) or 1..n_instances (<= 254), i.e. the n
 * non-overlapping masks (labels == i + 1); same output as quber_encode_initial_masks on those masks.  Values outside
 * 0..n_instances count as 0. */
int quber_encode_label_map(quber_ctx* ctx, const int32_t* dev_labels, int32_t batch, int32_t n_instances, float* dev_out,
                           void* stream);

/* device bytes the context owns (weights in kernel layout, activations, workspaces).  Everything is allocated by
 * quber_create / quber_finalize_weights; no hot call allocates. */
int64_t quber_workspace_bytes(quber_ctx* ctx);

/* a2 - explicit quadruple error maps.  Replaces masks_to_fg_mask / masks_to_boundary
 * (explicit_error_estimation/util.py:62-99) and the TP/TN/FP/FN logic of tools/ours/panoptic2eee.py:110-123.
 *   dev_init u8 [B][N][H][W], dev_gt u8 [B][Ng][H][W]  ->  dev_out u8 [B][2 (region,boundary)][4 (TP,TN,FP,FN)][H][W] */
int quber_explicit_error_maps(quber_ctx* ctx, const uint8_t* dev_init, int32_t n_init, const uint8_t* dev_gt,
                              int32_t n_gt, int32_t batch, uint8_t* dev_out, void* stream);

/* a3-a7 - the network.  Replaces MaskRefiner.forward up to the head outputs
 * (maskrefiner/modeling/mask_refiner/model.py:137-156, 244-250, 689-708).
 *   dev_bgr u8 [B][H][W][3], dev_depth u8 [B][H][W][3] (NULL when streams == 1: dev_bgr then carries the single
 *   image, rgb or depth), dev_offsets f32 [B][3][H][W]
 *   -> dev_logits f32 [B][4+classes][H][W]  (fg logit, centre, off_y px, off_x px, boundary-error logits) */
int quber_forward(quber_ctx* ctx, const uint8_t* dev_bgr, const uint8_t* dev_depth, const float* dev_offsets,
                  int32_t batch, float* dev_logits, void* stream);

/* quber_forward with a HIP-event pair around every launch group of the plan, on `stream` (synchronises it).
 *   kind_ms[3] / kind_launches[3]: summed device time and launch-group count of
 *   [0] implicit-GEMM convolutions, [1] GroupNorm (stats + apply), [2] everything else.  Benchmark use only. */
int quber_forward_profiled(quber_ctx* ctx, const uint8_t* dev_bgr, const uint8_t* dev_depth,
                           const float* dev_offsets, int32_t batch, float* dev_logits, void* stream,
                           double* kind_ms, int32_t* kind_launches);

/* Stage profile (benchmark use): every quber_* call made on `ctx` by this thread between begin and end brackets each of
 * its kernels with a HIP-event pair on the launch stream.  quber_profile_end synchronises `stream` and sums, per stage
 * name ("encode_reduce", "preprocess", "conv_gemm", "wino_input", "wino_gemm", "wino_output", "splitk_reduce",
 * "gn_apply", "upsample_logits", "post_nms", "post_group", "extract_masks", "errmaps_pack", ...): device milliseconds,
 * the stage's ALGORITHMIC bytes (every operand read / written once) and FLOPs, and the number of brackets. */
int quber_profile_begin(quber_ctx* ctx);
int quber_profile_end(quber_ctx* ctx, void* stream);
int quber_profile_num_stages(quber_ctx* ctx);
int quber_profile_stage(quber_ctx* ctx, int index, const char** name, double* ms, double* bytes, double* flops,
                        int32_t* launches);

/* a8-a11 - centre NMS/top-k, pixel grouping, 512-px merge, scores and boxes.  Replaces get_panoptic_segmentation
 * (maskrefiner/modeling/mask_refiner/post_processing.py:165-221) and the instance loop of model.py:313-356.
 *   dev_logits f32 [B][n_planes][H][W] (planes 0..3 used)
 *   -> dev_panoptic f32 [B][H][W]  labels {-1, 1000, 1001, ...}
 *      dev_count   i32 [B]         instances per frame
 *      dev_labels  f32 [B][top_k]  their labels, ascending, -1 padded
 *      dev_scores  f32 [B][top_k], dev_boxes f32 [B][top_k][4] (x0,y0,x1,y1)
 *      dev_centers i32 [B][top_k][2] (y,x) and dev_ncenters i32 [B]  (the surviving centre points) */
int quber_postprocess(quber_ctx* ctx, const float* dev_logits, int32_t n_planes, int32_t batch,
                      float* dev_panoptic, int32_t* dev_count, float* dev_labels, float* dev_scores,
                      float* dev_boxes, int32_t* dev_centers, int32_t* dev_ncenters, void* stream);

/* pred_masks of model.py:334 for the first `max_inst` labels of every frame:
 *   -> dev_masks u8 [B][max_inst][H][W] in {0,1} (all zero for slots beyond the frame's count) */
int quber_extract_masks(quber_ctx* ctx, const float* dev_panoptic, const float* dev_labels, int32_t batch,
                        int32_t max_inst, uint8_t* dev_masks, void* stream);

/* evaluation support - all pairwise overlap counts of two label maps in one pass.  Replaces the per-pair
 * np.count_nonzero loops of eval/evaluation.py:180-199 (multilabel_metrics).
 *   dev_pred, dev_gt i32 [n_pixels], label values in 0..65535, at most `cap` (<= 1024) distinct values per map
 *   workspace (quber_contingency_workspace_bytes(cap) bytes, device) receives, in this order:
 *     u32 flags[2][65536] | u64 table[cap][cap] (row = gt index, column = pred index) | i32 labels[2][cap] (sorted unique
 *     values: [0] pred, [1] gt) | i32 counts[4] = (n_pred, n_gt, out-of-range flag, 0) | u16 lut[2][65536] */
int64_t quber_contingency_workspace_bytes(int32_t cap);
int quber_label_contingency(const int32_t* dev_pred, const int32_t* dev_gt, int64_t n_pixels, int32_t cap,
                            void* dev_workspace, void* stream);

/* evaluation support - the boundary half of multilabel_metrics: for every (ground-truth object i, predicted object j) the
 * true-positive counts of eval/evaluation.py:21-54 `boundary_overlap` and the per-object boundary sizes of :165-175.
 * seg2bmap (eval/utilities.py:672-697: cv2.findContours RETR_EXTERNAL + drawContours) and the disk dilation are restated
 * from their published algorithms (no OpenCV / skimage in the build image; parity unpinned).
 *   dev_pred, dev_gt i32 [h][w] label maps; dev_labels i32 [n_pred + n_gt]: the object labels, predicted ones first
 *   bound_pix: disk radius = ceil(0.003 * hypot(h, w)) in the reference (evaluation.py:33-34)
 *   -> dev_out u32: boundary size [n_pred + n_gt] | precision_tps [n_gt][n_pred] | recall_tps [n_gt][n_pred]
 *   dev_workspace: quber_boundary_workspace_bytes(h, w, n_pred + n_gt) bytes; h * ceil(w / 64) * 8 must fit 160 KiB of LDS */
int64_t quber_boundary_workspace_bytes(int32_t h, int32_t w, int32_t n_masks);
int quber_boundary_overlap(const int32_t* dev_pred, const int32_t* dev_gt, int32_t h, int32_t w, const int32_t* dev_labels,
                           int32_t n_pred, int32_t n_gt, int32_t bound_pix, void* dev_workspace, uint32_t* dev_out,
                           void* stream);

/* post-filter of eval/refiner_model.py:273-277 on LMFFNet logits (foreground_segmentation/predictor.py:85,98):
 *   dev_fg_logits f32 [B][n_classes][HW] -> dev_fg_mask u8 [B][HW] = (argmax == fg_class)
 *   dev_masks u8 [B][n_masks][HW] (may be NULL with n_masks = 0)
 *   -> dev_counts u64 [B][n_masks][2] = (|mask & fg|, |mask|); the caller keeps masks with inter / area > 0.3 */
int quber_foreground_filter(const float* dev_fg_logits, int32_t n_classes, int32_t fg_class, const uint8_t* dev_masks,
                            int32_t batch, int32_t n_masks, int64_t hw, uint8_t* dev_fg_mask, uint64_t* dev_counts,
                            void* stream);

/* adapter pre-processing - depth normalisation.  Replaces normalize_depth (eval/preprocess_utils.py:12-28) and the
 * zero-depth bookkeeping of eval/refiner_model.py:250.
 *   dev_depth: uint16 [n] (is_float32 = 0, e.g. PNG millimetres, evaluated in float64 like numpy) or f32 [n]
 *   -> dev_out3 u8 [n][3] (the value replicated to 3 channels), dev_zero u8 [n] = (depth == 0), may be NULL */
int quber_normalize_depth(const void* dev_depth, int32_t is_float32, int64_t n_pixels, double min_val, double max_val,
                          uint8_t* dev_out3, uint8_t* dev_zero, void* stream);

/* adapter pre-processing - cv2.resize of an interleaved uint8 image (eval/refiner_model.py:229 `cv2.resize(rgb, (w, h))`,
 * :232 / :254 INTER_NEAREST on masks / depth, :246).  linear = 1: cv2.INTER_LINEAR's 8-bit path (11-bit fixed-point
 * weights; the area filter at exactly half scale), linear = 0: cv2.INTER_NEAREST.  OpenCV is absent from the build image:
 * restated from its published algorithm, parity unpinned.
 *   dev_src u8 [src_h][src_w][channels] (channels 1..4) -> dev_dst u8 [dst_h][dst_w][channels] */
int quber_resize_u8(const uint8_t* dev_src, int32_t src_h, int32_t src_w, int32_t channels, uint8_t* dev_dst,
                    int32_t dst_h, int32_t dst_w, int32_t linear, void* stream);

/* adapter pre-processing - cv2.inpaint(img, mask, radius, cv2.INPAINT_TELEA) of one 8-bit channel (inpaint_depth,
 * eval/preprocess_utils.py:44-64).  The ONE entry point that takes HOST pointers and runs on the host, like the OpenCV call
 * it replaces (the fast-marching method is sequential; it runs once per frame outside the refiner's timed region).
 * Restated from Telea's published algorithm in the form OpenCV implements it; parity unpinned, own tolerance.
 *   host_img u8 [h][w], host_mask u8 [h][w] (non-zero = to be filled)  ->  host_out u8 [h][w] */
int quber_inpaint_telea_u8(const uint8_t* host_img, const uint8_t* host_mask, int32_t h, int32_t w, int32_t radius,
                           uint8_t* host_out);
/* inpaint_depth(depth, kernel_size) of eval/preprocess_utils.py:44-64 (factor 1) in one host call: mask = pixels whose three channels
 * are 0, dilated by a kernel x kernel square; TELEA in-painting (radius = kernel) per channel; only the zero pixels are replaced.
 *   host_depth3 u8 [h][w][3]  ->  host_out3 u8 [h][w][3] */
int quber_inpaint_depth_u8(const uint8_t* host_depth3, int32_t h, int32_t w, int32_t kernel, uint8_t* host_out3);
/* The same on the DEVICE, asynchronous on `stream`, for a batch of frames, bit-equal to the host function above (csrc/inpaint_dev.hip):
 * the hole regions that can influence each other (connected components of the mask dilated by radius + 2) are marched one wave each
 * with the host's queue order; a pixel's (2 r + 1)^2 neighbours are evaluated one per lane and summed in the host's order.
 * kernel <= 3 (the reference calls it with 3: eval/refiner_model.py:255).
 *   dev_depth3 u8 [batch][h][w][3] -> dev_out3 u8 [batch][h][w][3]; dev_workspace: quber_inpaint_depth_workspace_bytes(batch, h, w) bytes */
int64_t quber_inpaint_depth_workspace_bytes(int32_t batch, int32_t h, int32_t w);
int quber_inpaint_depth_device(const uint8_t* dev_depth3, int32_t batch, int32_t h, int32_t w, int32_t kernel, void* dev_workspace,
                               int64_t workspace_bytes, uint8_t* dev_out3, void* stream);

/* ---- introspection / kernel-level entry points used by the parity tests and the benchmark ---- */
/* device pointer + NHWC geometry of a named intermediate of the last quber_forward ("res2", "res3", "res5", "y", ...) */
int quber_debug_tensor(quber_ctx* ctx, const char* name, float** dev_ptr, int32_t* dims4, int32_t* channel_stride);
/* bytes per element of that intermediate: 4 (fp32), or 2 (fp16) in the fp16 data path (compute_dtype 2); 0 = no such tensor */
int32_t quber_debug_tensor_elem_size(quber_ctx* ctx, const char* name);
/* algorithmic FLOPs of one forward at batch 1 (2 * MACs of every convolution) */
double quber_forward_flops(quber_ctx* ctx);
/* FLOPs the matrix pipe actually executes per forward at batch 1: a layer planned as Winograd F(m x m,3x3) counts
 * (m+2)^2 / (9 m^2) of its algorithmic FLOPs, padded tiles included (transform additions not counted) */
double quber_forward_flops_executed(quber_ctx* ctx);
/* ... and the part of those executed FLOPs that is tile padding (ragged maps, short phases of dilated layers): a 30x40 map under 4x4 tiles
 * carries 6.7 % of it */
double quber_forward_flops_padding(quber_ctx* ctx);
/* Options.  Every knob that shapes a plan or changes the arithmetic / work distribution of a launch belongs to a CONTEXT:
 *   quber_set_option(ctx, key, value)   this context only.  "plan" keys act when quber_finalize_weights builds the plan and
 *                                       are refused afterwards; "launch" keys may change between forwards.
 *   quber_get_option(ctx, key, &value)
 *   quber_set_tuning(key, value)        the PROCESS DEFAULTS: what a context created afterwards starts from (quber_create copies
 *                                       them) and what the stand-alone quber_op_* test ops, which have no context, use.  Contexts
 *                                       that already exist are not affected.  Keys 2, 11, 12, 26 exist only here (test harness).
 * Two engines with different options coexist in one process and may run on different threads.
 * Context keys (default; when it acts):
 * key 3  (0; launch, test harness) force the number of K partitions of convolutions that have a workspace (0 = automatic);
 * key 4  (0; launch, test harness) force the convolution tile shape (1 = 64x64, 2 = 128x128, 3 = 128x64, 4 = 256x32);
 * key 5  (1; launch) split the ragged last round of large convolution launches into K-pieces: when the cost model favours
 *         it (1), never (0), whenever feasible (2);
 * key 6  (0; plan) Winograd path of the eligible 3x3 layers: 0 = where it pays, 1 = never, 2 = always;
 * key 7  (32; plan) smallest input width (channels) routed to the Winograd path (below 128 channels a layer takes it only in the
 *         single-kernel form, key 25);
 * key 8  (67; plan) Winograd only while its multiplies are <= value % of the direct kernel's (dilated layers);
 * key 9  (0; plan) Winograd output tile edge of the eligible layers: 0 = automatic (F(4x4), or F(2x2) where its tiles fit the
 *         map better; both measure the direct kernel's error), 2, 4, or 6 = opt into F(6x6,3x3) where it executes >= 10 % fewer
 *         multiplies still (2.5x the error at tap level, +4.5 % throughput at batch 16); the algorithm of every layer is fixed at
 *         plan time from its geometry alone, never from the batch of a launch;
 * key 10 (32; plan) smallest output width routed to the Winograd path;
 * key 13 (1; launch) persistent convolution launches (csrc/conv_persist.hip): 0 = never (one tile per block everywhere),
 *         1 = the 128x128-tile launches, 2 = every tile shape;
 * key 14 (32; launch) persistent launches: shortest K, in 32-wide slices, whose remainder tiles are shared between blocks;
 * key 15 (256; launch) persistent launches: fewest tiles (all groups) of a launch that goes persistent (smaller launches - small
 *         batches - keep the one-tile-per-block kernel and its split-K model);
 * key 16 (0; diagnostics) 1 = the output stores of the persistent convolution kernel are dropped by the range check;
 * key 17 (0; launch) Winograd F(4x4) transforms on channel pairs instead of quads (measured: input transform 6 % slower);
 * key 18 (1; plan) the projection block of every ResNet stage runs conv3 and its shortcut as ONE 1x1 GEMM over the
 *         concatenated inputs (reference: detectron2 BottleneckBlock as built by maskrefiner/modeling/backbone/resnet.py:37-63)
 *         or as two convolutions (0);
 * key 19 (1; launch) 128x64 tiles for the convolutions with 33-64 output channels and no residual, or 64x64 (0);
 * key 20 (0; launch) Winograd pipeline layers in passes whose V | M intermediates stay below `value` MiB (0 = the whole batch at
 *         once; measured slower at every size: profiles/r03c_wino_subbatch_rejected.md);
 * key 21 (2; launch, ARITHMETIC) K-slices per chunk of the two-level fp32 accumulation of the GEMM kernels (the bf16x3 mode folds
 *         every `value` slices, the exact fp32 mode every slice; 0 = one sequential chain over K);
 * key 24 (1; launch) side lanes at batches <= 16 (exact fp32 and bf16x3: <= 12), or everything on the caller's stream (0);
 * key 25 (1; plan) Winograd F(4x4,3x3) layers as ONE kernel - input transform, the 36 position GEMMs and the output transform, no
 *         V | M intermediates in HBM (csrc/wino_fused.hip): the eligible layers, or never (0);
 * key 27 (160; plan, ARITHMETIC) widest input, in channels, that takes the single-kernel form.  Its two accumulation chains are
 *         Cin / 2 long - 80 channels at the default, with which the float64-anchor ratios stay 0.64-1.11; admitting 256 / 320
 *         channels (chains of 128-160) measures 1.20 / 1.21 and no speed-up inside the network (DECISIONS.md section 4,
 *         profiles/r05_fused_anchor.md, r05_wino_fused_layers.md).
 * key 29 (1; plan) the input normalisation + concatenation (a3, model.py:137-153) inside the first stem convolution's kernel
 *         (csrc/stem.hip: same fmaf chains as the implicit GEMM, no normalised input tensor in HBM; fp16 data path: the layer on the
 *         matrix pipe from 16-byte fp16 pixels staged in LDS, equal to the preprocess kernel + implicit GEMM up to fp32 summation
 *         order; 2 = the fp16 data path keeps the vector-FMA form of the kernel), or as a kernel of its own (0).
 * key 30 (1; launch) implicit GEMM loader: block-uniform filter taps in scalar registers, a tap-validity bit mask and a 32-bit byte
 *         offset per row, buffer loads whose out-of-range offset returns the zero padding (conv_igemm.hip LEAN; layers with
 *         Cin % 32 == 0 - fp16 data path: % 64 - and tensors below 2 GiB), or per-thread tap arithmetic everywhere (0).  Same bits.
 * key 31 (1; plan + launch) fp16 data path: the layers with >= 256 output channels and K a multiple of 64 on 256 x 256 tiles with the LDS-DMA
 *         operand pipeline (csrc/conv_h8.hip: resnet.py:395-449 res4 / res5 bottlenecks, :472-485 fusion convolutions, the ASPP
 *         branches of model.py:610-651), 0 = the 128-tile kernel everywhere, 2 = also narrower outputs (tests).  Same arithmetic
 *         (fp16 products, fp32 accumulation), another summation order inside a K-tile.
 *         (plan + launch, keys 31 and 38: the value at quber_finalize_weights decides the SHAPE of the plan - with 31 on, the three dilated
 *         ASPP branches are one grouped op; with 31, 38 and 39 on, a norm in front of a patch-kernel layer is planned as absorbed - so op
 *         list, GroupNorm slots and quber_op_info names differ between plans built under different values.  The value at LAUNCH only
 *         selects kernels: with a key switched off afterwards the same ops run on conv_igemm.hip and an absorbed norm as the pass it was.)
 * key 32 (224; launch) fewest tiles (all groups) of a launch that key 31 takes: its blocks own a CU each.
 * keys 33, 34: retired (the exact-fp32 256 x 128 LDS-DMA kernel was no faster in the network and was removed: profiles/r11_f8.md).
 * key 35 (1; launch) bf16x3 mode: the wide 1x1 launches and the Winograd position GEMMs on csrc/conv_x8.hip (256 x 128 tiles, LDS-DMA
 *         pipeline, the weights pre-split into three bf16 planes when the plan is built, the activations split in registers by the
 *         wave that is not multiplying); the same six partial products in the same order as conv_igemm.hip's bf16x3 kernels.
 *         0 = those kernels everywhere, 2 = every covered launch (tests).
 * key 36 (2; launch) fewest rounds of tiles (tiles / CUs), key 37 (8; launch) fewest K-slices of 32 of a launch that key 35 = 1 takes.
 * key 38 (1; plan + launch) fp16 data path: the undilated 3x3 / stride 1 layers with the pixel operand as an LDS patch - a tile is 8 x 32 output
 *         pixels, the 10 x 34-pixel patch of a 64-channel block is fetched once and the nine taps read it at shifted addresses, instead of
 *         nine DMA gathers (conv_h8.hip: conv_h8p_kernel up to 128 output channels, conv_h8w_kernel 256 and more, conv_h8s_kernel the stem's
 *         32-channel inputs with LDS-resident filters); 0 = key 31's DMA-gather kernels / conv_igemm.hip there, 2 = only up to 128 channels.
 * key 39 (1; plan) fp16 data path: a patch-kernel layer of up to 128 output channels that is the only reader of a GroupNorm + ReLU output
 *         (decoder fuse convolutions, the heads' second convolution: model.py:386-403, 610-651) applies that norm to its LDS patches - the
 *         arithmetic of the norm pass, bit for bit, without the pass over the tensor in HBM; 0 = every norm is a pass of its own.
 * key 41 (1; plan) small batches (side lanes, key 24): the dilated ASPP branches d = 6 / 12 (model.py:610-651) on the two side lanes,
 *         beside the d = 18 and 1x1 branches on the caller's stream; 0 = one after the other.  Same launches, same bits.
 * key 42 (1; launch) exact fp32 (and the bf16x3 mode's launches that conv_x8.hip does not take): 1x1 GEMMs - bottleneck / fusion / ASPP 1x1
 *         convolutions of resnet.py:395-449, 472-485 and model.py:610-651, Winograd position GEMMs - on 64 x 64 tiles, one per block, instead of
 *         the 128 x 128 split-K / persistent launch when (a) the 128 x 128 tiling has at most 1 280 tiles (small batches), or (b, exact fp32 only)
 *         K <= 1024 whatever the size; 2 = rule (a) only, 0 = neither.  The same K order per output element: a re-association at most.
 * key 43 (1; launch) the dilated 3x3 layers whose launch skips the filter rows that lie in the zero padding (ASPP d = 18, model.py:610-651, on
 *         64-row tiles) skip the padded filter COLUMNS as well: the GEMM rows of an image run through three column zones (left tap padded |
 *         both taps inside or both padded | right tap padded) as a serpentine, a tile multiplies only the filter columns its zones meet.
 *         The skipped taps multiply zeros: same bits unless the launch is split over K (then a re-association); 0 = rows only.
 * Process-only keys (quber_set_tuning): key 2 = give the stand-alone conv ops a split-K workspace (value != 0) or drop it (0);
 * key 11 = stand-alone conv op: dilated 3x3 layers in tap-major K order with the zero-padding filter rows skipped;
 * key 12 = stand-alone conv ops: quber_config.compute_dtype of the launch (1 = bf16 / 2 = fp16 operands, 3 = bf16x3);
 * key 26 = timing harness: quber_op_conv3x3_winograd reuses the transformed filters of its previous call (u / ws untouched) */
int quber_set_option(quber_ctx* ctx, int32_t key, int32_t value);
int quber_get_option(quber_ctx* ctx, int32_t key, int32_t* value);
void quber_set_tuning(int32_t key, int32_t value);
/* Host view of the work distribution of a persistent convolution launch (csrc/conv_persist.hip), for the CPU tests: no GPU.
 * _segments: the work list of block `block` of a launch of `blocks` blocks over `tiles` tiles of `k_slices` K-slices each
 *   (min_share = shortest K share of the remainder, 0 = remainder tiles whole): up to cap x (tile, first slice, end slice,
 *   workspace slot or -1 for a whole tile); returns their number.
 * _fixup: for remainder tile j of XCD run xcd: its tile index and the workspace slots the fix-up pass sums, in order;
 *   returns their number, 0 when the tile is computed whole, -1 past the remainder. */
int32_t quber_debug_persistent_segments(int32_t tiles, int32_t blocks, int32_t k_slices, int32_t min_share, int32_t block,
                                        int32_t* out4, int32_t cap);
int32_t quber_debug_persistent_fixup(int32_t tiles, int32_t blocks, int32_t k_slices, int32_t min_share, int32_t xcd, int32_t j,
                                     int32_t* tile, int32_t* slots, int32_t cap);
/* the launch plan of quber_forward, in execution order (after the input pre-processing kernel):
 * kind 0 = convolution, 1 = GroupNorm, 2 = other; flops = algorithmic FLOPs at batch 1 */
int quber_num_ops(quber_ctx* ctx);
int quber_op_info(quber_ctx* ctx, int index, const char** name, int32_t* kind, double* flops, int32_t* launches);
/* stand-alone convolution: y = act((conv(x, w) * scale + shift) + residual), NHWC, weights OIHW on the device */
int quber_op_conv2d(const float* dev_x, int32_t batch, int32_t h, int32_t w, int32_t cin, const float* dev_w_oihw,
                    int32_t cout, int32_t ksize, int32_t stride, int32_t pad, int32_t dil, const float* dev_scale,
                    const float* dev_shift, const float* dev_residual, int32_t relu, float* dev_packed_scratch,
                    float* dev_y, void* stream);
/* the 1x1 / stride 1 convolution of the fp16 data path (compute_dtype 2 with fp16 tensors in HBM): x [batch][h][w][cin],
 * w [cout][cin], residual and y [batch][h][w][cout] are fp16, scale / shift fp32; cin a multiple of 64 (test hook: the
 * network reaches this kernel only through quber_forward). */
int quber_op_conv1x1_f16(const void* dev_x, int32_t batch, int32_t h, int32_t w, int32_t cin, const void* dev_w_oi,
                         int32_t cout, const float* dev_scale, const float* dev_shift, const void* dev_residual,
                         int32_t relu, void* dev_y, void* stream);
/* a convolution of the fp16 data path on fp16 tensors: x [batch][h][w][cin] (cin a multiple of 8; kmode 1: of 64), w_packed [cout][Kpad]
 * (Kpad = k*k*cin rounded up to a multiple of 64, rows zero-filled) in the kernels' K order - kmode 0: k = (tap, channel); kmode 1:
 * k = (channel / 64, tap, channel % 64) - residual and y fp16, scale /
 * shift fp32; gn_sums (or null): f64 [batch][gn_groups][2], the sums and sums of squares of the stored outputs per norm group
 * are ADDED to it (the GroupNorm statistics the network's convolutions gather in their epilogue).  Test hook: the layers of
 * maskrefiner/modeling/backbone/resnet.py:395-449, 472-485 reach these kernels through quber_forward. */
int quber_op_conv2d_f16(const void* dev_x, int32_t batch, int32_t h, int32_t w, int32_t cin, const void* dev_w_packed,
                        int32_t cout, int32_t ksize, int32_t stride, int32_t pad, int32_t dil, int32_t kmode,
                        const float* dev_scale, const float* dev_shift, const void* dev_residual, int32_t relu,
                        double* dev_gn_sums, int32_t gn_groups, void* dev_y, void* stream);
/* one 1x1 GEMM over two inputs: out = relu?(y . w[:, :mid] + x[::stride, ::stride] . w[:, mid:] + shift), NHWC;
 * y [batch][oh][ow][mid], x [batch][h2][w2][cin], w [cout][mid + cin], `dev_ones` = cout ones (the kernel's affine scale).
 * Needs the op workspace (key 2) and fp32 / bf16x3 arithmetic (key 12 = 0 / 3). */
int quber_op_conv1x1_dual(const float* dev_y, const float* dev_x, int32_t batch, int32_t oh, int32_t ow, int32_t mid,
                          int32_t h2, int32_t w2, int32_t cin, int32_t stride, const float* dev_w, const float* dev_shift,
                          const float* dev_ones, int32_t cout, int32_t relu, float* dev_out, void* stream);
/* the same for a 3x3 / stride 1 / pad = dilation convolution through the Winograd F(m x m, 3x3) path, m = 2, 4 or 6
 * (cin >= 128 and a multiple of 32, cout >= 128): with P = (m+2)^2, dev_u_scratch holds P*cout*cin floats (transformed
 * weights), dev_ws at least P * batch * dil^2 * ceil(ceil(h/dil)/m) * ceil(ceil(w/dil)/m) * (cin + cout) floats.
 * m = 4 with cin <= key 27, 32 | cout and a dev_ws of at least 36 * cout * cin floats: the single-kernel form (key 25). */
int quber_op_conv3x3_winograd(const float* dev_x, int32_t batch, int32_t h, int32_t w, int32_t cin,
                              const float* dev_w_oihw, int32_t cout, int32_t dil, int32_t m, const float* dev_scale,
                              const float* dev_shift,
                              int32_t relu, float* dev_u_scratch, float* dev_ws, int64_t ws_floats, float* dev_y,
                              void* stream);
int quber_op_groupnorm(const float* dev_x, int32_t batch, int32_t h, int32_t w, int32_t c, int32_t groups,
                       const float* dev_gamma, const float* dev_beta, float eps, int32_t relu,
                       double* dev_stats_scratch, float* dev_y, void* stream);
int quber_op_bilinear(const float* dev_x, int32_t batch, int32_t h, int32_t w, int32_t c, int32_t oh, int32_t ow,
                      float* dev_y, void* stream);
int quber_op_maxpool3x3s2(const float* dev_x, int32_t batch, int32_t h, int32_t w, int32_t c, float* dev_y,
                          void* stream);
/* a9 alone, on a caller-supplied centre list in ANY order - group_pixels(ctr, offsets) of post_processing.py:44-76
 * (quber_postprocess derives its centres from the centre plane, i.e. always in raster order; the reference's function takes
 * whatever list it is handed).  The grouping kernel quber_postprocess launches, unchanged.
 *   dev_logits f32 [B][n_planes][H][W]: plane 0 foreground logit (a pixel is grouped iff sigmoid(x).round() == 1),
 *              planes 2, 3 the (y, x) offsets; dev_centers i32 [B][cap][2] (y, x), dev_ncenters i32 [B], cap <= 254
 *   -> dev_ids u8 [B][H][W]: 0 = background, 1..K = index of the nearest centre + 1 (first index wins ties), 255 = foreground
 *      of a frame without centres; dev_area u32 [B][256]: pixels per id */
int quber_op_group_pixels(const float* dev_logits, int32_t n_planes, int32_t batch, int32_t h, int32_t w, int32_t cap,
                          const int32_t* dev_centers, const int32_t* dev_ncenters, uint8_t* dev_ids, uint32_t* dev_area,
                          void* stream);

#ifdef __cplusplus
}
#endif
#endif
