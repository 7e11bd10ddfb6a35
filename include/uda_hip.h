/* uda_hip.h — C ABI of the MI355X (gfx950) hot path of uncertainty-detection-autolabeling.
 *
 * One shared library (uncertainty-detection-autolabeling_amd/csrc/libuda_hip.so), plain
 * pointers and sizes, no torch / C++ types.  It replaces what the reference reaches
 * through `infer_lib.ServingDriver` (reference src/infer_lib.py:118-296):
 *
 *   uda_create            <- KerasDriver.__init__: build EfficientDetModel + restore weights
 *                            (infer_lib.py:416-440; efficientdet_keras.py:850-970)
 *   uda_set_images_u8     <- the uint8 [N,h,w,3] `image_arrays` argument of serve()  (infer_lib.py:337-343,442-448)
 *   uda_run               <- EfficientDetModel.call: _preprocessing -> EfficientDetNet.call (MC loop)
 *                            -> _postprocess                     (efficientdet_keras.py:1076-1146, 979-1050)
 *   uda_get_detections    <- the output tuple of postprocess_global / postprocess_per_class
 *                            (postprocess.py:472-621, 624-740)
 *   uda_serve             <- ServingDriver.serve                (set_images + run + get_detections)
 *   uda_predict           <- ServingDriver.predict with only_network=True: float images -> raw head outputs
 *                            (infer_lib.py:345-350,450-457)
 *   uda_postprocess_heads <- ServingDriver._postprocess = postprocess_global on given head outputs
 *                            (infer_lib.py:263-267)
 *
 * Conventions: every function returns 0 on success, non-zero on error (message via
 * uda_last_error); inputs are borrowed, outputs are caller-allocated; a handle owns one
 * GPU's weights, workspace and HIP stream, is not thread-safe, and every call that
 * returns data synchronises its stream before returning.  Host orchestration (config,
 * topology -> op list, weight packing) stays in Python (plan.py) exactly as the
 * reference keeps model building in Python above the TF runtime.
 */
#ifndef UDA_HIP_H_
#define UDA_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UDA_ABI_VERSION 4
#define UDA_MAX_LEVELS 8
#define UDA_MAX_FUSE_INPUTS 3

/* ---- activation buffers (NHWC float32) ---------------------------------------------- */
typedef struct uda_buf_desc {
  int32_t H, W, C;      /* one sample */
  int32_t per_sample;   /* 0: one row per image (shared by all MC samples); 1: one row per (image, sample) */
  int64_t offset;       /* float offset inside the per-chunk arena (liveness-planned by plan.py) */
  int32_t kind;         /* 0 arena, 1 network input image, 2 class head output, 3 box head output */
  int32_t level;        /* pyramid level index for kind 2/3 */
} uda_buf_desc_t;

/* ---- ops ------------------------------------------------------------------------------ */
enum uda_op_kind {
  UDA_OP_STEM = 1,  /* 3x3 stride-2 conv (Cin=3) + BN + swish          efficientnet_model.py:588-612 */
  UDA_OP_PW = 2,    /* 1x1 conv (+SE input scale)(+bias)(+BN)(+swish)(+dropout)(+residual)  :358-373,403-418,471-486 */
  UDA_OP_DW = 3,    /* depthwise kxk stride s SAME (+BN)(+swish)(+dropout)(+SE partial sums) :376-391,459-464 */
  UDA_OP_SE = 4,    /* mean -> fc+bias -> swish -> fc+bias -> sigmoid   :219-232 */
  UDA_OP_FUSE = 5,  /* BiFPN weighted fusion of resampled inputs + swish  efficientdet_keras.py:86-127,229-231 */
  UDA_OP_POOL = 6,  /* max pool (stride+1) x (stride+1), stride s, SAME   efficientdet_keras.py:280-290 */
  UDA_OP_MBX = 7,   /* fused MBConv front half: 1x1 expand + BN + swish + dropout -> depthwise kxk/s + BN + swish
                       + dropout + SE tile sums, the 6x-expanded tensor never leaves the CU
                       (efficientnet_model.py:446-464).  With se_scale >= 0 the op also absorbs the PREVIOUS block's
                       projection (efficientnet_model.py:471-486): in[0] is that block's gated-depthwise tensor D
                       [C0 <= 32 channels], se_scale its per-row gate [rows, C0], se_w1_off the projection kernel
                       [C0][se_mid], se_b1_off / se_w2_off its BN scale / shift, se_mid = 16 projected channels */
  UDA_OP_SEP = 8    /* fused SeparableConv2D: depthwise 3x3 stride 1 SAME (kernel at w2_off) -> 1x1 (kernel at w_off)
                       (+bias)(+BN)(+swish)(+dropout); the depthwise result never leaves the CU
                       (efficientdet_keras.py:207-227,421-446,584-626) */
};
/* utils.activation_fn (reference utils.py:42-59): swish / silu / swish_native are one function; srelu (it carries a trainable
 * beta) is refused by the planner (ValueError, as the reference raises for an unknown act_type) */
enum uda_act { UDA_ACT_NONE = 0, UDA_ACT_SWISH = 1, UDA_ACT_RELU = 2, UDA_ACT_RELU6 = 3, UDA_ACT_HSWISH = 4, UDA_ACT_MISH = 5 };
enum uda_resample { UDA_RS_NONE = 0, UDA_RS_NEAREST_UP = 1, UDA_RS_MAXPOOL = 2 };

typedef struct uda_op {
  int32_t kind;
  int32_t in[UDA_MAX_FUSE_INPUTS]; /* buffer ids, -1 = unused */
  int32_t out;                     /* buffer id */
  int32_t se_scale;                /* PW: buffer id of the [rows, Cin] SE gate applied to the input, or -1 */
  int32_t se_partial;              /* DW: buffer id receiving per-tile channel sums; SE: the same buffer as input */
  int32_t residual;                /* PW: buffer id added to the output, or -1 */
  int32_t k, stride;               /* DW / POOL / STEM */
  int32_t act;                     /* uda_act, applied after bias/BN, before dropout */
  int64_t w_off;                   /* float offsets into the weight blob; -1 = absent */
  int64_t bias_off;                /*   conv bias [Cout]                                */
  int64_t bn_scale_off;            /*   gamma * rsqrt(var + 1e-3)  [Cout]               */
  int64_t bn_shift_off;            /*   beta - mean * scale         [Cout]               */
  int64_t se_w1_off, se_b1_off, se_w2_off, se_b2_off; /* SE: [C][mid], [mid], [mid][C], [C] */
  int32_t se_mid;
  int32_t drop_site;               /* index of the MC-dropout site applied to the output, -1 = none */
  int32_t resample[UDA_MAX_FUSE_INPUTS]; /* FUSE: uda_resample per input */
  float fuse_w[UDA_MAX_FUSE_INPUTS];     /* FUSE: relu(w_i) / (sum relu(w) + 1e-4) */
  int32_t n_in;
  int32_t drop_site2;              /* MBX: dropout site after the depthwise stage (drop_site = after the expand stage).
                                      SEP (plain, not fuse_in): a DEFERRED site of the producing op - keep-scales [sample row][C] of
                                      this op's INPUT channels, which the producer (a per-image tensor shared by the T samples) did not
                                      apply: a per-channel factor commutes with the depthwise conv, so the op computes the depthwise
                                      result once per image and serves the T samples from it.  -1 = none */
  int64_t w2_off;                  /* MBX: depthwise kernel [k*k][Cmid]; SEP: depthwise kernel [9][C] */
  int64_t bn2_scale_off, bn2_shift_off; /* MBX: BN after the depthwise stage */
  int32_t launch_group;            /* SEP: n > 1 on the first of n consecutive, mutually independent ops of one shape class
                                      (same channels, activation, sample axes) that may share ONE launch - the pyramid levels
                                      of a head layer; the planner keeps every buffer they touch alive to the end of the run.
                                      0 / 1 elsewhere. */
  int32_t fuse_in;                 /* SEP: 1 = the conv input is the BiFPN fusion swish(sum_i fuse_w[i] * resample[i](in[i])) of in[0 .. n_in),
                                      computed on the fly for the tile (no fused tensor in memory); 0 elsewhere */
  int32_t fuse_act;                /* SEP with fuse_in: uda_act of that fusion (act_type; UDA_ACT_NONE under conv_bn_act_pattern,
                                      efficientdet_keras.py:229-236) */
} uda_op_t;

/* ---- MC dropout sites -------------------------------------------------------------------- */
typedef struct uda_drop_site {
  int32_t channels;
  float rate;            /* keep iff u >= rate, scale 1/(1-rate)  (SpatialDropout2D, noise shape [N,1,1,C]) */
} uda_drop_site_t;

/* ---- model / post-processing description ---------------------------------------------------- */
enum uda_decode { UDA_DECODE_PLAIN = 0, UDA_DECODE_LNORM = 1, UDA_DECODE_FALSEDEC = 2,
                  UDA_DECODE_SAMPLE = 3 /* utils_box.py:162-184: moments of decode_nsamples decoded Normal draws (Philox stream) */ };
enum uda_post_mode { UDA_POST_GLOBAL = 0, UDA_POST_PER_CLASS = 1 };

typedef struct uda_model {
  int32_t abi_version;
  int32_t image_h, image_w;         /* network input size (H, W) = parse_image_size(image_size) */
  float mean_rgb[3], stddev_rgb[3];
  int32_t num_levels;
  int32_t level_h[UDA_MAX_LEVELS], level_w[UDA_MAX_LEVELS];
  int32_t anchors_per_loc;          /* num_scales * len(aspect_ratios) */
  int32_t num_classes;
  int32_t loss_attenuation;         /* box head has 8*A channels: [4A box | 4A sigma]  (postprocess.py:448-462) */
  int32_t mc_samples;               /* T (1 when mc_dropout is off) */
  int32_t cls_stacked, box_stacked; /* head outputs carry the T axis                  (efficientdet_keras.py:1026-1049) */
  int32_t has_uncert;               /* loss_attenuation or mc_dropout: uncertainty columns are emitted (postprocess.py:443) */
  int32_t decode_method;            /* uda_decode */
  int32_t enable_softmax;           /* logits output present */
  /* nms (postprocess.py:373-400) */
  float nms_soft_sigma;             /* sigma/2 handed to NonMaxSuppressionV5; 0 = hard */
  float nms_iou_thresh, nms_score_thresh;
  int32_t max_output_size;
  int32_t max_nms_inputs;           /* 0: argmax per anchor; >0: top-k over anchors*classes (postprocess.py:96-121) */
  int32_t post_mode;                /* default uda_post_mode used by uda_run */
  /* planning */
  int32_t chunk_images;             /* images processed per pass of the op list */
  int32_t max_images;               /* capacity of one uda_run */
  int64_t arena_floats;             /* per-chunk arena size */
  int32_t n_drop_sites;
  int32_t decode_nsamples;          /* UDA_DECODE_SAMPLE: draws per (anchor, MC sample); config `decode_nsamples` (100) */
} uda_model_t;

typedef struct uda_ctx uda_ctx_t;

/* Build a handle on HIP device `device`: uploads `weights`, the anchor table, allocates
 * the arena, head-output and NMS workspaces.  Returns NULL-handle + error on failure. */
int uda_create(const uda_model_t* model, const uda_buf_desc_t* bufs, int32_t n_bufs,
               const uda_op_t* ops, int32_t n_ops, const uda_drop_site_t* sites,
               const float* weights, int64_t n_weights, const float* anchors /* [A_tot*4] */,
               int32_t device, uda_ctx_t** out);
void uda_destroy(uda_ctx_t* ctx);
/* last error of `ctx` (or of the last failed uda_create when ctx == NULL) */
const char* uda_last_error(const uda_ctx_t* ctx);

/* CRC-32C of a host buffer (continue from `crc`, 0 to start): the per-tensor checksum of the TensorFlow checkpoint bundles
 * `utils_keras.restore_ckpt` reads (utils_keras.py:125-235); host code, no GPU call. */
uint32_t uda_crc32c(const void* data, uint64_t n, uint32_t crc);

/* Raw uint8 images [n,h,w,3] from host memory into the handle's device staging buffer (replaces the feed of
 * ServingDriver.serve, infer_lib.py:139-151,232-249).  The bytes are gathered into a pinned host buffer owned by the
 * handle and leave by one DMA: the caller's array is free as soon as the call returns. */
int uda_set_images_u8(uda_ctx_t* ctx, const uint8_t* images, int32_t n, int32_t h, int32_t w);
/* A batch whose raw sizes differ per image - KITTI frames are 370-376 x 1224-1242 (dataset_data.py:105) and the
 * reference's callers feed them one file at a time (validate_model.py:479-483, infer_model.py:554-581): images[i] points
 * at [h[i], w[i], 3] bytes.  Every image gets its own resize scale / scaled size / image_scale, exactly as
 * dataloader.py:123-152 computes them per image. */
int uda_set_images_u8_ragged(uda_ctx_t* ctx, const uint8_t* const* images, int32_t n, const int32_t* h, const int32_t* w);
/* Feed hiding: upload the NEXT batch into the handle's second input slot on a copy stream while the current batch is
 * being processed (call it after uda_run has queued the current batch); uda_swap_prefetched then makes that batch the
 * input of the following uda_run (the compute stream waits for the upload event, no host synchronisation).  The reference
 * times serve() including its feed (validate_model.py:154-158); with these two calls the feed costs no device time. */
int uda_prefetch_images_u8(uda_ctx_t* ctx, const uint8_t* images, int32_t n, int32_t h, int32_t w);
int uda_prefetch_images_u8_ragged(uda_ctx_t* ctx, const uint8_t* const* images, int32_t n, const int32_t* h, const int32_t* w);
int uda_swap_prefetched(uda_ctx_t* ctx);
/* Same, from a device pointer the caller owns (device-to-device copy on the stream). */
int uda_set_images_u8_device(uda_ctx_t* ctx, const void* images_dev, int32_t n, int32_t h, int32_t w);
/* The handle's own device copy of the current uint8 batch (one raw size): lets the members of a deep ensemble share ONE
 * upload - member 0 takes the host batch, the others uda_set_images_u8_device from its buffer (BASELINE configs[3]). */
int uda_input_u8_device(uda_ctx_t* ctx, const void** images_dev, int32_t* n, int32_t* h, int32_t* w);
/* Preprocessed float images [n,H,W,3] (the `only_network` input); scales default to 1. */
int uda_set_images_f32(uda_ctx_t* ctx, const float* images, int32_t n, const float* image_scales);

/* Dropout masks: either generated on the device from `seed` (Philox4x32-10)
 * or injected: `masks` = concatenation over sites of float32 [n*T, channels] keep-scales. */
int uda_set_dropout_seed(uda_ctx_t* ctx, uint64_t seed);
/* Index of the handle's first image inside the global batch (multi-GPU image shards): the Philox row
 * of image n, sample t is (offset + n) * T + t, so a sharded batch draws the masks of the unsharded one. */
int uda_set_dropout_image_offset(uda_ctx_t* ctx, int64_t first_image);
/* MC samples sharded over ranks (north_star: "images (and optionally MC samples) shard"): the handle's T local samples are
 * samples t_first + j * t_stride (j < T) of a global axis of t_total samples - the Philox row of image n, local sample j is
 * (offset + n) * t_total + t_first + j * t_stride, so the ranks together draw exactly the masks of one handle that runs all
 * t_total samples.  t_total = 0 restores "this handle runs every sample".  (dist.serve_sample_sharded) */
int uda_set_dropout_sample_shard(uda_ctx_t* ctx, int32_t t_first, int32_t t_stride, int32_t t_total);
int uda_set_dropout_masks(uda_ctx_t* ctx, const float* masks, int64_t n_floats);
int uda_get_dropout_masks(uda_ctx_t* ctx, float* masks, int64_t n_floats);

/* preprocess (if uint8 images are set) -> network x T -> post-process, asynchronously on the
 * handle's stream.  post_mode < 0 uses the model default; run_post == 0 stops after the heads. */
int uda_run(uda_ctx_t* ctx, int32_t post_mode, int32_t run_post);
int uda_synchronize(uda_ctx_t* ctx);
/* Pipelined serving of a stream of batches (infer_lib.ServingDriver.serve_images called batch after batch,
 * infer_lib.py:504-540; the reference runs them strictly one after the other): uda_run_async queues network + post-process
 * of the current input like uda_run(ctx, post_mode, 1) but does NOT order the handle's main stream behind the
 * post-process - the next uda_run_async's network starts at once and only its first head-writing op waits for it, so the
 * ~4 ms of latency-bound aggregate / NMS / gather launches of batch k run beside the backbone of batch k + 1.
 * *ticket (0 | 1) names the run; at most two may be in flight.  uda_collect waits for that run's post-process only and
 * copies its detections (same arrays as uda_get_detections; NULL = skip) - the detection outputs and the image scales
 * exist once per ticket, so a run queued behind it cannot disturb them.  Typical loop:
 *     run_async(&t0);  for each further batch: { set / swap images; run_async(&t1); collect(t0, ...); t0 = t1; }  collect(t0, ...)
 * Failures are loud: a range / barrier flag raised by a run whose candidates a newer run has replaced fails the collect
 * (the batch has to be run again); entry points that rewrite head buffers refuse while a run is in flight.  Results are
 * bit-identical to uda_run's. */
int uda_run_async(uda_ctx_t* ctx, int32_t post_mode, int32_t* ticket);
int uda_collect(uda_ctx_t* ctx, int32_t ticket, float* boxes, float* scores, float* classes, int32_t* valid, float* logits);
/* The same for the multi-GPU gather: the run's detections as ONE device-resident record buffer (see uda_detections_device
 * for the layout and `rows`); packed on a stream of its own, complete when the call returns; closes the ticket. */
int uda_collect_device(uda_ctx_t* ctx, int32_t ticket, int32_t rows, int32_t with_logits, void** dev_ptr, int32_t* cols);
/* Abandon every pipelined run in flight (results discarded, tickets closed, flags cleared): what a consumer that drops
 * ServingDriver.serve_stream half way, or whose uda_collect failed, calls before it uses the handle synchronously again. */
int uda_drain(uda_ctx_t* ctx);
/* The default scheme splits the operands of the 1x1 contractions into two fp16 pieces (float32-class products at the
 * matrix-core cost of bf16): an operand above 65504 cannot be split.  The reference computes in float32 and never rejects
 * an input on magnitude (utils.py:595-609), so neither does the handle: the op that saw such an operand is re-packed with
 * three bf16 pieces (float32 exponent range), the run is served again from its unchanged inputs before any reader sees
 * it, and the op stays that way.  This counts the ops re-packed so far (0 for every weight set the tests initialise). */
int64_t uda_range_demotions(const uda_ctx_t* ctx);
/* Global NMS over the whole anchor set runs on the score prefix that can be selected at all, checked on the
 * device; this counts the images / problems that failed the check and were redone on the full set (the results
 * are identical either way). */
int64_t uda_nms_prefix_fallbacks(const uda_ctx_t* ctx);
/* Global NMS over the whole anchor set normally runs as ONE launch of a co-resident grid whose blocks exchange keys
 * through memory with a bounded spin.  When a spin runs out (the grid was not co-resident: e.g.
 * another process interleaving such grids on the same GPU) the post-process is redone with two launches per epoch and
 * the handle stays on that version; this counts those redone post-process runs (0 in normal operation; results are
 * identical either way).  UDA_NMS_COOP_SPIN=<polls> shortens the bound (test hook). */
int64_t uda_nms_coop_fallbacks(const uda_ctx_t* ctx);
/* NMS runs that were eligible for the single-launch grid but did NOT get it (capacity query failed, grid larger than the
 * device holds, launch refused) and therefore ran the slower prefix / two-launches-per-epoch versions: 0 in normal
 * operation; a non-zero value explains a slow post-process that uda_nms_coop_fallbacks (time-outs only) would not show. */
int64_t uda_nms_coop_not_launched(const uda_ctx_t* ctx);

/* Detections of the last uda_run (synchronises).  Shapes for n images, M = max_output_size:
 *   boxes   [n, M, box_cols]   box_cols = 4 (+4 aleatoric sigma)(+4 epistemic sigma)
 *   scores  [n, M]
 *   classes [n, M, cls_cols]   cls_cols = 1 (+num_classes MC std of the logits)
 *   valid   [n] int32
 *   logits  [n, M, num_classes] (may be NULL)                       (postprocess.py:610-621) */
int uda_get_detections(uda_ctx_t* ctx, float* boxes, float* scores, float* classes,
                       int32_t* valid, float* logits);
int uda_detection_cols(const uda_ctx_t* ctx, int32_t post_mode, int32_t* box_cols, int32_t* cls_cols);
/* The detections of the last post-process as ONE device-resident float32 buffer [rows, max_output_size, cols], a row per
 * (image, detection): boxes (box_cols) | score | classes (cls_cols) | logits (num_classes, when with_logits and the global
 * post-process ran) | valid_len - the record of dist.pack_detections.  rows >= the images of the last post-process; the
 * rows beyond them are zero (padding of a ragged shard to the collective's common size).  The multi-GPU layer all-gathers
 * straight out of this buffer (RCCL): the detections do not cross PCIe before they have been collected (the
 * reference has no inference collective - infer_lib.py:337-343 returns host arrays of one process).  The buffer belongs to
 * the handle and is valid until its next call; the handle's stream has been synchronised when the call returns. */
int uda_detections_device(uda_ctx_t* ctx, int32_t rows, int32_t with_logits, void** dev_ptr, int32_t* cols);

/* What every caller of serve() computes next from `logits` (validate_model.py:159-166,
 * infer_model.py:585-600, utils_class.py:36-41), on the device: probs [n, M, num_classes] = stable softmax of the
 * selected rows' mean logits, entropy [n, M] = -sum p * log2(max(p, 1e-7)).  Global post-process only. */
int uda_get_class_probs(uda_ctx_t* ctx, float* probs, float* entropy);

/* serve = set_images_u8 + run + get_detections */
int uda_serve(uda_ctx_t* ctx, const uint8_t* images, int32_t n, int32_t h, int32_t w,
              float* boxes, float* scores, float* classes, int32_t* valid, float* logits);

/* Calibrated box uncertainty of the last global post-process (CalibrateBoxUncert.calibrate_boxuncert,
 * utils_box.py:404-524), on the device.  col0 = first of the four uncertainty columns inside the boxes output (4:
 * aleatoric or the only one, 8: epistemic when both exist).  Isotonic tables are the fitted thresholds of the
 * reference's sklearn IsotonicRegression models: table t = xs/ys[tab_off[t] .. tab_off[t+1]); 1 table (ISO_ALL), 4
 * (ISO_PERCOO: ymin, xmin, ymax, xmax) or 4 * num_classes (ISO_PERCLSCOO, class-major, class ids 1..C);
 * relative != 0 = the rel_iso_perclscoo variant.  temps: 1 or 4 divisors for the TS modes.  out [n, M, 4]. */
enum uda_calib_mode { UDA_CALIB_TS_ALL = 0, UDA_CALIB_TS_PERCOO = 1, UDA_CALIB_ISO_ALL = 2, UDA_CALIB_ISO_PERCOO = 3,
                      UDA_CALIB_ISO_PERCLSCOO = 4 };
int uda_calibrate_box(uda_ctx_t* ctx, int32_t col0, int32_t mode, int32_t relative, int32_t n_tables,
                      const int32_t* tab_off, const double* xs, const double* ys, const float* temps, float* out);

/* Calibrated class probabilities of the last global post-process (class half; CalibrateClass._perform_class_calib /
 * calibrate_class, utils_class.py:109-272), on the device next to the logits they refine.  UDA_CLS_TS: logits / temps[c]
 * (ts_all: the same temperature num_classes times; ts_percls: one per class), stable softmax.  UDA_CLS_ISO_ALL / _PERCLS:
 * stable softmax, then the fitted isotonic table (1, or one per class; thresholds as in uda_calibrate_box), re-normalised to
 * sum 1.  draws == 0: the mean logits are calibrated (model without MC class uncertainty).  draws > 0 (the reference uses 10):
 * that many logit vectors ~ Normal(mean logits, MC std of the logits) are calibrated; probs = their mean, uncert = their
 * population std, entropy = entropy of the mean (needs the class-std columns: MC dropout on the class head, argmax path).
 * The draws come from the build's Philox stream with `seed` (TFP's stream cannot be reproduced).
 * probs [n, M, num_classes], entropy [n, M], uncert [n, M, num_classes] (may be NULL). */
enum uda_class_calib_mode { UDA_CLS_TS = 0, UDA_CLS_ISO_ALL = 1, UDA_CLS_ISO_PERCLS = 2 };
int uda_calibrate_class(uda_ctx_t* ctx, int32_t mode, int32_t n_tables, const int32_t* tab_off, const double* xs,
                        const double* ys, const float* temps, int32_t draws, uint64_t seed, float* probs, float* entropy,
                        float* uncert);

/* Raw head outputs of the last run, level `level`: class [T_c, n, h, w, A*C] and
 * box [T_b, n, h, w, 4A or 8A] in the reference's stacking order (T axis first; T_x = 1
 * and the axis is dropped by the caller when that head is not stacked). */
int uda_get_head_outputs(uda_ctx_t* ctx, int32_t level, float* cls, float* box);
/* Inject head outputs (same layout) and run only the post-process on them.  cls_floats / box_floats = number of floats
 * the host buffers hold; they must equal T_x * n * h * w * channels of the handle's layout (an error otherwise: the
 * caller passed an unstacked array to a stacked head or the other way round). */
int uda_set_head_outputs(uda_ctx_t* ctx, int32_t level, int32_t n, const float* cls, int64_t cls_floats,
                         const float* box, int64_t box_floats);
/* post-process (ServingDriver._postprocess, infer_lib.py:263-267) on the head outputs RESIDENT in the handle - injected
 * with uda_set_head_outputs / uda_copy_heads / written through uda_head_outputs_device, or left by the last uda_run. */
int uda_postprocess_heads(uda_ctx_t* ctx, int32_t n, const float* image_scales, int32_t post_mode);
/* Device address of the handle's head-output buffer of `level` (which: 0 class, 1 box): rows [image][sample], each of
 * *floats_per_row floats, *rows_per_image = T for a stacked head else 1; capacity max_images images.  For device-side
 * exchanges (RCCL over xGMI: ensemble re-shard) without a host hop; the caller orders its own stream against
 * the handle's with uda_synchronize.  uda_set_num_images tells the handle how many images such a write filled in. */
int uda_head_outputs_device(uda_ctx_t* ctx, int32_t level, int32_t which, void** dev_ptr, int64_t* floats_per_row,
                            int32_t* rows_per_image);
int uda_set_num_images(uda_ctx_t* ctx, int32_t n);
/* Deep ensembles (BASELINE configs[3]; the reference has no ensemble code): copy the
 * head outputs of the last run of `src` (a deterministic member network, T = 1) into sample slot
 * `sample` of `dst` (a handle whose model has mc_samples = number of members and stacked heads);
 * uda_postprocess_heads(dst) then aggregates the members exactly like MC samples (a8 / a14).
 * Device-to-device on dst's stream; both handles must live on the same GPU and share the geometry. */
int uda_copy_heads(uda_ctx_t* dst, const uda_ctx_t* src, int32_t n, int32_t sample);
/* predict = set_images_f32 + run(no post) ; read back with uda_get_head_outputs */
int uda_predict(uda_ctx_t* ctx, const float* images, int32_t n);

/* Pre-NMS candidates of the last run (debug / parity): boxes [n,K,4], scores [n,K], classes [n,K] int32 */
int uda_get_candidates(uda_ctx_t* ctx, float* boxes, float* scores, int32_t* classes,
                       float* u_cls, float* u_al, float* u_ep);
int32_t uda_num_candidates(const uda_ctx_t* ctx);

/* Read back activation buffer `buf` of the LAST chunk processed (debug / parity). */
int uda_read_buffer(uda_ctx_t* ctx, int32_t buf, float* host, int64_t n_floats);
int uda_get_preprocessed(uda_ctx_t* ctx, float* images /* [n,H,W,3] */, float* scales /* [n] */);

/* Standalone NMS on host arrays (parity tests of the NonMaxSuppressionV5 kernel):
 * boxes [n_img, k, 4], scores [n_img, k] -> idx [n_img, M], out_scores [n_img, M], valid [n_img] */
int uda_nms(uda_ctx_t* ctx, const float* boxes, const float* scores, int32_t n_img, int32_t k,
            int32_t max_out, float iou_thresh, float score_thresh, float soft_sigma, int32_t pad,
            int32_t* idx, float* out_scores, int32_t* valid);

/* Standalone 1x1 convolution on host arrays (op-level parity tests and timing of the pointwise kernels):
 * out[r, p, :] = (act(bn(in[r / in_div, p, :] * se[r / in_div, :] @ w + bias)) * mask[r, :]) + res[r, p, :]
 * in [rows/in_div, hw, cin], w [cin, cout], se [rows/in_div, cin], mask [rows, cout], res/out [rows, hw, cout];
 * optional arguments may be NULL.  terms: 0 = f32-input MFMA, 3 / 6 = split-bf16 MFMA with 3 / 6 cross terms, 16 = two fp16
 * pieces per operand with 3 cross terms (the default scheme of the network, UDA_PW_SCHEME=f16x2; fails - it does not
 * return infinities - when an input exceeds fp16's 65504) (kernels_pwb.hip).  The launch is repeated `reps` times for
 * *avg_ms (HIP events). */
int uda_debug_pw(int32_t device, const float* in, const float* w, const float* bias, const float* bn_scale,
                 const float* bn_shift, const float* se, const float* mask, const float* res,
                 int32_t rows, int32_t in_div, int32_t hw, int32_t cin, int32_t cout, int32_t act,
                 int32_t terms, int32_t reps, float* out, float* avg_ms);

/* The numpy NMS family of the reference (row a18, src/nms_np.py:30-278): x1,y1,x2,y2 boxes with the +1 pixel
 * convention.  method: 0 hard_nms, 1 diou_nms, 2 soft_nms gaussian, 3 soft_nms linear (thresholds as the caller
 * resolved them: `iou_thresh or 0.5` ...).  uda_nms_np: float64 dets [n,5] -> kept rows out [<= n, 5] in selection
 * order, *n_out rows.  uda_per_class_nms_np = nms_np.per_class_nms on float32 inputs (boxes y1,x1,y2,x2; classes
 * 0-based): out [max_boxes, 7] rows [image_id, x1, y1, x2, y2, score, class + 1], dummy rows score -1e5, boxes
 * multiplied by image_scale. */
int uda_nms_np(int32_t device, const double* dets, int32_t n, int32_t method, double iou_thresh, double sigma,
               double score_thresh, double* out, int32_t* n_out);
int uda_per_class_nms_np(int32_t device, const float* boxes, const float* scores, const int32_t* classes, int32_t k,
                         float image_id, float image_scale, int32_t num_classes, int32_t max_boxes, int32_t method,
                         float iou_thresh, float sigma, float score_thresh, float* out);

/* Per-op-kind device timing with HIP events recorded on the handle's stream.
 * kind_mask: bit (1 << uda_op_kind) selects op kinds; bit 16 post-process aggregate, bit 17 NMS. */
int uda_profile_enable(uda_ctx_t* ctx, uint32_t kind_mask);
int uda_profile_read(uda_ctx_t* ctx, int32_t kind, double* total_ms, int64_t* launches, int32_t reset);

#ifdef __cplusplus
}
#endif
#endif /* UDA_HIP_H_ */
