// Host side of the C ABI (include/uda_hip.h): owns one GPU's weights, arena, head-output and
// post-process workspaces and a HIP stream, and executes the op list that plan.py lowered
// from the reference's model description.  No torch, no Python types.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <numeric>
#include <string>
#include <vector>

#include "uda_internal.h"

using namespace uda;

static thread_local std::string g_create_error;

struct ProfSlot {
  double total_ms = 0;
  int64_t launches = 0;       // in units of PLANNED ops: a launch that covers a group of n ops counts n (plan.op_costs counts per op)
  struct Pending { hipEvent_t first, second; int weight; };
  std::vector<Pending> pending;
};

struct uda_ctx {
  uda_model_t model;
  std::vector<uda_buf_desc_t> bufs;
  std::vector<uda_op_t> ops;
  std::vector<uda_drop_site_t> sites;
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;

  float* d_weights = nullptr;
  int64_t n_weights = 0;
  // split-bf16 copies of the 1x1 kernels in MFMA fragment order (kernels_pwb.hip); -1 = op keeps the f32 path
  uint16_t* d_wsplit = nullptr;
  std::vector<int64_t> wsplit_off;
  std::vector<int64_t> wpar_off;   // MBX: offset (uint16 units) of the per-slab depthwise operand block inside d_wsplit
  int pw_parts = UDA_SPLIT_F16X2;  // requested split scheme of the 1x1 contractions (UDA_PW_SCHEME / UDA_PW_TERMS, parse_pw_scheme)
  std::vector<int> wscheme;        // per op: the scheme its packed weights use (an op whose weights do not suit fp16 pieces keeps bf16 x3)
  std::vector<float> wunscale;     // per op: 1 / (power-of-two factor folded into the packed weights); 1 unless fp16 pieces
  int n_f16_ops = 0, n_f16_demoted = 0;
  // fp16 pieces: a kernel that splits an operand above 65504 sets bit 0 of ITS OP's flag word.  Two arrays of n_ops + 1
  // words (index n_ops: launches outside the op list): pipelined run s raises its flags in array s, everything else in array 0.
  unsigned* d_oor = nullptr;
  unsigned* oor_cur = nullptr;     // the array the launches being queued raise their flags in
  int oor_half = 0;                // the array the readers of the current results look at (check_split_range)
  bool oor_armed = false;          // a run with fp16-piece ops has been queued since the flags were last read
  // An op that raises its flag is re-packed with three bf16 pieces (float32 exponent range) and the run is served again
  // on the same handle (demote_ops / replay_run): the reference computes in float32 and never rejects an input on magnitude.
  std::vector<float> h_weights;    // host copy of the weight blob (for the re-packing)
  std::vector<uint16_t*> wovr;     // per op: device copy of its re-packed weights (null: its slice of d_wsplit)
  std::vector<int64_t> wovr_par;   // per op: uint16 offset of the parameter block inside wovr (-1: none)
  int64_t range_demotions = 0;     // ops re-packed so far (uda_range_demotions)
  // What a run read, so that it can be served again: input slot / float image generation, seed, image offset, masks.
  struct RunRec {
    bool valid = false, do_post = false, have_u8 = false, masks_injected = false;
    int pm = 0, cur = 0, n = 0;
    uint64_t slot_gen = 0, f32_gen = 0, masks_gen = 0, seed = 0;
    int64_t image_offset = 0;
  };
  RunRec last_run;                 // the last synchronous uda_run
  const RunRec* replay_rec = nullptr;   // the run whose results the readers are looking at (null: cannot be served again)
  uint64_t f32_gen = 0, masks_gen = 0;
  float* d_arena = nullptr;
  uint4* d_w0frag[2] = {nullptr, nullptr};   // gated, split projection kernel per gate row for the fused block-1 kernel (launch_w0gate), per chunk lane
  size_t w0frag_cap[2] = {0, 0};
  // chunk lanes: consecutive chunks alternate between independent (stream, arena) pairs so that the
  // barrier-heavy kernels of one chunk overlap the streaming kernels of the other
  int n_lanes = 1;
  hipStream_t lane_stream[2] = {nullptr, nullptr};
  float* lane_arena[2] = {nullptr, nullptr};
  hipEvent_t ev_start = nullptr, ev_done[2] = {nullptr, nullptr};
  int last_lane = 0;
  // post-process of chunk i (aggregate, NMS, gather: small latency-bound launches) runs on its own stream
  // beside the conv stack of chunk i + 1; only the last chunk's post-process is exposed
  hipStream_t post_stream = nullptr;
  std::vector<hipEvent_t> ev_chunk;
  hipEvent_t ev_post = nullptr;
  int post_overlap = 1;
  float* d_anchors = nullptr;
  int A_tot = 0;
  int a_off[UDA_MAX_LEVELS + 1];

  // inputs.  uint8 batches go through one of two slots (device buffer + pinned host staging buffer each): `cur` feeds the
  // next uda_run; the other one takes a batch that is uploaded on the copy stream while the current one is being
  // processed (uda_prefetch_images_u8 / uda_swap_prefetched) - the feed then costs no device time.
  struct U8Slot {
    uint8_t* d = nullptr;        // device: geometry table, then the images back to back (image i at d_img + geo[i].off)
    size_t cap = 0;
    uint8_t* pinned = nullptr;   // host staging (hipHostMalloc): pageable caller memory is copied here, DMA reads this
    size_t pcap = 0;
    int n = 0;
    bool valid = false, uploaded = false;
    uint64_t gen = 0;            // bumped by every upload into this slot (a run can be served again only from unchanged inputs)
    std::vector<PreGeo> geo;     // per image: offset, raw size, scaled size, sampling ratios (dataloader.py:123-152)
    PreGeo* d_geo = nullptr;     // = d (the table leads the buffer)
    uint8_t* d_img = nullptr;    // = d + header
    std::vector<float> scales;   // image_scale per image (1 / resize scale)
    hipEvent_t ev = nullptr;     // upload complete (copy stream)
  } u8[2];
  int cur = 0;
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_pre_done[2] = {nullptr, nullptr};   // the preprocess kernel has consumed slot i (its buffer may be refilled)
  bool have_u8 = false;
  int stem_act = UDA_ACT_SWISH; // activation of the stem op (the uint8 stem is a swish kernel)
  bool stem_from_u8 = false;   // this run's stem ops read the uint8 slot (set by run_network)
  bool pre_valid = false;      // d_images holds the preprocessed current batch (false: the stem read the uint8 images itself)
  int stem_co = 0;             // output channels of the stem op (0: no stem op in the plan)
  float* d_images = nullptr;   // [max_images, H, W, 3]
  float* d_scales = nullptr;   // [max_images]
  std::vector<float> h_scales;
  int n_images = 0;
  int sh = 0, sw = 0;

  // dropout
  float* d_masks = nullptr;
  int64_t mask_cap = 0;        // floats
  int64_t sum_site_ch = 0;
  int max_c4 = 0;
  std::vector<int64_t> site_off;
  int64_t* d_site_off = nullptr;
  int32_t* d_site_ch = nullptr;
  float* d_site_rate = nullptr;
  bool masks_injected = false;
  int masks_rows = 0;
  uint64_t seed = 0;
  int64_t image_offset = 0;
  int t_first = 0, t_stride = 1, t_total = 0;      // this handle's samples inside the global sample axis (0: all of them; uda_set_dropout_sample_shard)

  // head outputs [max_images * Tx, hw, ch] per level
  float* d_cls[UDA_MAX_LEVELS] = {};
  float* d_box[UDA_MAX_LEVELS] = {};
  int cls_ch = 0, box_ch = 0;

  // candidates
  float *d_cboxes = nullptr, *d_cscores = nullptr, *d_clogits = nullptr;
  int32_t* d_cclasses = nullptr;
  float *d_ucls = nullptr, *d_ual = nullptr, *d_uep = nullptr;
  int Kc = 0;                  // candidates per image: A_tot, or max_nms_inputs on the top-k path
  float* d_clsmean = nullptr;  // [max_images, A_tot*C]  (top-k path)
  int32_t* d_cand_flat = nullptr;  // [max_images, Kc]   (top-k path)
  void* d_topk_ws = nullptr;       // scratch of the multi-block top-k selection
  // nms workspaces: [0] global mode (one problem per image), [1] per-class mode (images*classes problems)
  struct NmsWs {
    float *stale = nullptr, *tent = nullptr, *ub = nullptr, *sel_score = nullptr, *sel_box = nullptr;
    int32_t *ev = nullptr, *begin = nullptr, *sel_idx = nullptr, *nsel = nullptr, *done = nullptr;
    unsigned long long *bound = nullptr, *win = nullptr;
    bool ready = false;
  } ws[2];
  // NMS on a score prefix (global mode with the whole anchor set as candidates): sub-problem arrays + workspace
  struct PrefixWs {
    int32_t *sub_idx = nullptr, *bad = nullptr;
    float *sub_scores = nullptr, *sub_boxes = nullptr;
    uint32_t* excl = nullptr;
    NmsWs ws;
    int Lcap = 0;
  } pfx;
  std::vector<std::pair<int, int>> pfx_pending;   // image ranges whose prefix flags the host has not looked at yet
  bool pfx_off = false;                            // set while finish_post redoes rejected images
  int64_t pfx_fallbacks = 0;                       // images redone on the full candidate set so far
  int pfx_skip = 0, pfx_backoff = 0;               // runs left without the prefix / length of the last pause
  // cooperative single-launch NMS: per-problem barrier counters + one error word (barrier timed out)
  unsigned long long* d_coop_bar = nullptr;       // exchange slots, max_images x nms_coop_slot_words(max_output_size)
  int* d_coop_err = nullptr;
  bool coop_used = false;
  bool coop_off = false;                           // set after a barrier time-out: this handle stays on the two-launch version
  int64_t coop_fallbacks = 0;                      // post-process runs redone with two launches per epoch after such a time-out
  int64_t coop_not_launched = 0;                   // NMS runs that wanted the single-launch grid and did not get it (capacity query / launch refused)
  unsigned long long* d_merge_keys = nullptr;
  // outputs
  float *d_oboxes = nullptr, *d_oscores = nullptr, *d_oclasses = nullptr, *d_ologits = nullptr;
  float *d_oprobs = nullptr, *d_oentropy = nullptr;   // stable softmax / entropy of the selected rows (lazy)
  float* d_opacked = nullptr;                        // packed detection records for the multi-GPU gather (lazy, uda_detections_device)
  int32_t* d_ovalid = nullptr;
  int last_post_mode = 0;
  int last_n = 0;
  int last_chunk_i0 = 0, last_chunk_n = 0;
  // Pipelined runs (uda_run_async / uda_collect): the post-process of run k (aggregate, NMS, gather: ~4 ms of latency-bound
  // launches on the post stream) is NOT joined into the main stream; run k + 1's network starts at once and only its first
  // head-writing op waits for it.  What run k's post-process reads or writes and run k + 1 could touch exists twice, by
  // ticket: the detection outputs and the image scales (snapshot taken on the main stream when the run is queued).
  struct AsyncSlot {
    hipEvent_t ev = nullptr;             // post-process of this run done
    bool open = false;                   // queued, not collected yet
    bool joined = true;                  // the main stream has been made to wait for `ev`
    int64_t seq = 0;
    int n = 0, mode = 0;
    bool coop_used = false, oor_armed = false;
    bool cands_lost = false;             // an older run was served again after this one: its candidates are gone (no redo of its post-process)
    RunRec rec;                          // what this run read (replay_run)
    std::vector<std::pair<int, int>> pending;      // prefix-NMS ranges the host has not checked (rare path: no cooperative NMS)
    float *oboxes = nullptr, *oscores = nullptr, *oclasses = nullptr, *ologits = nullptr, *scales = nullptr;
    int32_t* ovalid = nullptr;
  };
  AsyncSlot as[2];
  int as_next = 0;
  int64_t as_seq = 0;
  bool as_ready = false;
  const float* d_scales_post = nullptr;  // what the post-process reads as image scales (null: d_scales)
  hipStream_t aux_stream = nullptr;      // uda_collect_device packs on it
  hipEvent_t gate_ev = nullptr;          // run_network: head-writing ops wait for this first (the previous run's post-process)

  uint32_t prof_mask = 0;
  ProfSlot prof[32];
};

static void free_prefix_ws(uda_ctx::PrefixWs& w);

static int fail(uda_ctx* c, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (c) c->err = buf; else g_create_error = buf;
  return 1;
}

#define HIPC(ctx, expr)                                                                   \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return fail(ctx, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

template <typename T>
static hipError_t dalloc(T** p, size_t n) {
  return hipMalloc((void**)p, (n ? n : 1) * sizeof(T));
}

// UDA_PW_SCHEME = f16x2 | bf16x3 | bf16x2 | f32 (mirror: plan.pw_scheme); the older UDA_PW_TERMS = 6 | 3 | 0 names the last three
static int parse_pw_scheme(std::string* err) {
  const char* v = getenv("UDA_PW_SCHEME");
  if (v && *v) {
    if (!strcmp(v, "f16x2")) return UDA_SPLIT_F16X2;
    if (!strcmp(v, "bf16x3")) return UDA_SPLIT_BF16X3;
    if (!strcmp(v, "bf16x2")) return UDA_SPLIT_BF16X2;
    if (!strcmp(v, "f32")) return UDA_SPLIT_NONE;
    if (err) *err = std::string("UDA_PW_SCHEME=") + v + ": expected f16x2, bf16x3, bf16x2 or f32";
    return -1;
  }
  const char* t = getenv("UDA_PW_TERMS");
  if (!t || !*t) return UDA_SPLIT_F16X2;
  const int terms = atoi(t);
  if (terms == 0) return UDA_SPLIT_NONE;
  if (terms == 6) return UDA_SPLIT_BF16X3;
  if (terms == 3) return UDA_SPLIT_BF16X2;
  if (err) *err = std::string("UDA_PW_TERMS=") + t + ": expected 6, 3 or 0";
  return -1;
}

static inline int same_pad_before(int in, int out, int k, int s) {
  int total = (out - 1) * s + k - in;
  if (total < 0) total = 0;
  return total / 2;
}

// ------------------------------------------------------------------------------------ profiling helpers
struct ProfScope {
  uda_ctx* c;
  int kind;
  hipStream_t st;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  bool on;
  int weight;
  ProfScope(uda_ctx* c_, int kind_, hipStream_t st_ = nullptr, int weight_ = 1)
      : c(c_), kind(kind_), st(st_ ? st_ : c_->stream), weight(weight_) {
    on = (c->prof_mask >> kind) & 1u;
    if (on) {
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      hipEventRecord(e0, st);
    }
  }
  ~ProfScope() {
    if (on) {
      hipEventRecord(e1, st);
      c->prof[kind].pending.push_back({e0, e1, weight});
    }
  }
};

static void prof_collect(uda_ctx* c, int kind) {
  ProfSlot& s = c->prof[kind];
  for (auto& pr : s.pending) {
    hipEventSynchronize(pr.second);
    float ms = 0;
    if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
      s.total_ms += ms;
      s.launches += pr.weight;
    }
    hipEventDestroy(pr.first);
    hipEventDestroy(pr.second);
  }
  s.pending.clear();
}

// ------------------------------------------------------------------------------------ create / destroy
extern "C" const char* uda_last_error(const uda_ctx_t* ctx) {
  return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

extern "C" void uda_destroy(uda_ctx_t* c) {
  if (!c) return;
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  if (c->post_stream) hipStreamSynchronize(c->post_stream);      // (a pipelined run nobody collected)
  for (int k = 0; k < 32; ++k) prof_collect(c, k);
  if (c->copy_stream) hipStreamSynchronize(c->copy_stream);
  if (c->as_ready) {      // the second output set of the pipelined runs; the first one is the handle's own (freed below)
    const uda_ctx::AsyncSlot& a0 = c->as[0];
    c->d_oboxes = a0.oboxes; c->d_oscores = a0.oscores; c->d_oclasses = a0.oclasses; c->d_ologits = a0.ologits; c->d_ovalid = a0.ovalid;
    const uda_ctx::AsyncSlot& a1 = c->as[1];
    void* sp[] = {a1.oboxes, a1.oscores, a1.oclasses, a1.ologits, a1.ovalid, a0.scales, a1.scales};
    for (void* p : sp)
      if (p) hipFree(p);
    for (auto& a : c->as) if (a.ev) hipEventDestroy(a.ev);
  }
  for (auto& sl : c->u8) {
    if (sl.d) hipFree(sl.d);
    if (sl.pinned) hipHostFree(sl.pinned);
    if (sl.ev) hipEventDestroy(sl.ev);
  }
  for (auto& e : c->ev_pre_done) if (e) hipEventDestroy(e);
  if (c->copy_stream) hipStreamDestroy(c->copy_stream);
  for (auto& p : c->d_w0frag) if (p) hipFree(p);
  if (c->d_oor) hipFree(c->d_oor);
  for (uint16_t* p : c->wovr) if (p) hipFree(p);
  void* ptrs[] = {c->d_weights, c->d_wsplit, c->d_arena, c->d_anchors, c->d_images, c->d_scales, c->d_masks,
                  c->d_site_off, c->d_site_ch, c->d_site_rate, c->d_cboxes, c->d_cscores, c->d_clogits,
                  c->d_cclasses, c->d_ucls, c->d_ual, c->d_uep, c->d_clsmean, c->d_cand_flat, c->d_merge_keys,
                  c->d_oboxes, c->d_oscores, c->d_oclasses, c->d_ologits, c->d_ovalid, c->d_oprobs, c->d_oentropy,
                  c->d_opacked};
  for (void* p : ptrs)
    if (p) hipFree(p);
  free_prefix_ws(c->pfx);
  if (c->d_topk_ws) hipFree(c->d_topk_ws);
  if (c->d_coop_bar) hipFree(c->d_coop_bar);
  if (c->d_coop_err) hipFree(c->d_coop_err);
  for (auto& w : c->ws) {
    void* wp[] = {w.stale, w.tent, w.ub, w.sel_score, w.sel_box, w.ev, w.begin, w.sel_idx, w.nsel, w.done, w.bound, w.win};
    for (void* p : wp)
      if (p) hipFree(p);
  }
  for (int l = 0; l < UDA_MAX_LEVELS; ++l) {
    if (c->d_cls[l]) hipFree(c->d_cls[l]);
    if (c->d_box[l]) hipFree(c->d_box[l]);
  }
  for (int l = 1; l < 2; ++l) {
    if (c->lane_stream[l]) hipStreamDestroy(c->lane_stream[l]);
    if (c->lane_arena[l]) hipFree(c->lane_arena[l]);
  }
  if (c->post_stream) hipStreamDestroy(c->post_stream);
  if (c->aux_stream) hipStreamDestroy(c->aux_stream);
  for (auto e : c->ev_chunk) if (e) hipEventDestroy(e);
  if (c->ev_post) hipEventDestroy(c->ev_post);
  if (c->ev_start) hipEventDestroy(c->ev_start);
  for (auto e : c->ev_done) if (e) hipEventDestroy(e);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
}

static int box_cols_of(const uda_model_t& m, int post_mode) {
  if (post_mode == UDA_POST_PER_CLASS) return 4;
  int cols = 4;
  if (m.has_uncert && m.loss_attenuation) cols += 4;
  if (m.has_uncert && m.box_stacked) cols += 4;
  return cols;
}
static int cls_cols_of(const uda_model_t& m, int post_mode) {
  if (post_mode == UDA_POST_PER_CLASS) return 1;
  // top-k path gathers ONE class-std value per (anchor, class) candidate (postprocess.py:117-121)
  return 1 + ((m.has_uncert && m.cls_stacked) ? (m.max_nms_inputs > 0 ? 1 : m.num_classes) : 0);
}

static hipError_t alloc_nms_ws(uda_ctx::NmsWs& w, size_t problems, size_t K, size_t M) {
  if (w.ready) return hipSuccess;
  hipError_t e;
#define WS(ptr, n) if ((e = dalloc(&ptr, (n))) != hipSuccess) return e
  WS(w.stale, problems * K); WS(w.tent, problems * K); WS(w.ub, problems * K); WS(w.ev, problems * K);
  WS(w.begin, problems * K); WS(w.sel_idx, problems * M); WS(w.sel_score, problems * M);
  WS(w.sel_box, problems * M * 4); WS(w.bound, problems * M); WS(w.win, problems * M);
  WS(w.nsel, problems); WS(w.done, problems);
#undef WS
  w.ready = true;
  return hipSuccess;
}

// UDA_NMS_SOLO = candidates per problem up to which the single-launch NMS kernel is used directly (0 = never)
static int solo_limit() {
  static int solo = -1;
  if (solo < 0) { const char* e = getenv("UDA_NMS_SOLO"); solo = e ? atoi(e) : 8192; }
  return solo;
}

// candidates passed on to the prefix NMS: at least UDA_NMS_PREFIX (default 2048, 0 = always the full set), at most twice that
static int prefix_target() {
  static int L = -1;
  if (L < 0) {
    const char* e = getenv("UDA_NMS_PREFIX");
    L = e ? atoi(e) : 2048;
    if (L < 0) L = 0;
    if (L > 0 && L < 128) L = 128;
    if (L > 4096) L = 4096;
  }
  return L;
}

static hipError_t alloc_prefix_ws(uda_ctx::PrefixWs& w, size_t problems, int Lcap, size_t M) {
  if (w.Lcap) return hipSuccess;
  hipError_t e;
  if ((e = dalloc(&w.sub_idx, problems * Lcap)) != hipSuccess) return e;
  if ((e = dalloc(&w.sub_scores, problems * Lcap)) != hipSuccess) return e;
  if ((e = dalloc(&w.sub_boxes, problems * Lcap * 4)) != hipSuccess) return e;
  if ((e = dalloc(&w.excl, problems)) != hipSuccess) return e;
  if ((e = dalloc(&w.bad, problems)) != hipSuccess) return e;
  if ((e = alloc_nms_ws(w.ws, problems, (size_t)Lcap, M)) != hipSuccess) return e;
  w.Lcap = Lcap;
  return hipSuccess;
}

static void free_prefix_ws(uda_ctx::PrefixWs& w) {
  void* p[] = {w.sub_idx, w.sub_scores, w.sub_boxes, w.excl, w.bad, w.ws.stale, w.ws.tent, w.ws.ub, w.ws.sel_score, w.ws.sel_box,
               w.ws.ev, w.ws.begin, w.ws.sel_idx, w.ws.nsel, w.ws.done, w.ws.bound, w.ws.win};
  for (void* q : p)
    if (q) hipFree(q);
  w = uda_ctx::PrefixWs();
}

extern "C" int uda_detection_cols(const uda_ctx_t* c, int32_t post_mode, int32_t* box_cols, int32_t* cls_cols) {
  if (!c) return 1;
  const int pm = post_mode < 0 ? c->model.post_mode : post_mode;
  if (box_cols) *box_cols = box_cols_of(c->model, pm);
  if (cls_cols) *cls_cols = cls_cols_of(c->model, pm);
  return 0;
}

extern "C" int uda_create(const uda_model_t* model, const uda_buf_desc_t* bufs, int32_t n_bufs,
                          const uda_op_t* ops, int32_t n_ops, const uda_drop_site_t* sites,
                          const float* weights, int64_t n_weights, const float* anchors,
                          int32_t device, uda_ctx_t** out) {
  if (!out) return fail(nullptr, "uda_create: out is NULL");
  *out = nullptr;
  if (!model || !bufs || !ops || !weights || !anchors) return fail(nullptr, "uda_create: NULL argument");
  if (model->abi_version != UDA_ABI_VERSION)
    return fail(nullptr, "uda_create: ABI version %d, library has %d", model->abi_version, UDA_ABI_VERSION);
  if (model->num_levels < 1 || model->num_levels > UDA_MAX_LEVELS)
    return fail(nullptr, "uda_create: num_levels %d out of range", model->num_levels);
  if (model->max_output_size < 1 || model->max_output_size > 128)
    return fail(nullptr, "uda_create: max_output_size %d outside [1, 128]", model->max_output_size);
  if (model->chunk_images < 1 || model->max_images < 1 || model->mc_samples < 1)
    return fail(nullptr, "uda_create: chunk_images/max_images/mc_samples must be >= 1");
  if (model->decode_method == UDA_DECODE_SAMPLE && (model->decode_nsamples < 1 || model->decode_nsamples > 4096))
    return fail(nullptr, "uda_create: decode_nsamples %d outside [1, 4096]", model->decode_nsamples);
  if (model->mc_samples > 96)      // the aggregate kernel parks T logits + 4 T box corners per candidate in LDS
    return fail(nullptr, "uda_create: mc_samples %d > 96 unsupported", model->mc_samples);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, "uda_create: no HIP device available (the HIP path has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(nullptr, "uda_create: device %d of %d", device, ndev);

  uda_ctx* c = new uda_ctx();
  c->model = *model;
  c->bufs.assign(bufs, bufs + n_bufs);
  c->ops.assign(ops, ops + n_ops);
  for (int i = 0; i < n_ops; ++i)
    if (ops[i].kind == UDA_OP_STEM && ops[i].out >= 0 && ops[i].out < n_bufs) {
      c->stem_co = bufs[ops[i].out].C;
      c->stem_act = ops[i].act;
    }
  if (model->n_drop_sites > 0 && sites) c->sites.assign(sites, sites + model->n_drop_sites);
  c->device = device;
  c->n_weights = n_weights;
#define CK(expr)                                                                                  \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      fail(nullptr, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);  \
      uda_destroy(c);                                                                             \
      return 1;                                                                                   \
    }                                                                                             \
  } while (0)
  CK(hipSetDevice(device));
  CK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));

  const uda_model_t& m = c->model;
  const int T = m.mc_samples;
  // validate the op list against the buffer table before anything is launched
  for (int i = 0; i < n_ops; ++i) {
    const uda_op_t& o = ops[i];
    auto okbuf = [&](int id) { return id >= 0 && id < n_bufs; };
    if (!okbuf(o.out)) { fail(nullptr, "op %d: bad out buffer %d", i, o.out); uda_destroy(c); return 1; }
    for (int j = 0; j < o.n_in; ++j)
      if (!okbuf(o.in[j])) { fail(nullptr, "op %d: bad in[%d] buffer %d", i, j, o.in[j]); uda_destroy(c); return 1; }
    if (o.drop_site >= m.n_drop_sites || ((o.kind == UDA_OP_MBX || o.kind == UDA_OP_SEP) && o.drop_site2 >= m.n_drop_sites)) {
      fail(nullptr, "op %d: bad drop site %d", i, o.drop_site);
      uda_destroy(c);
      return 1;
    }
    if (o.drop_site >= 0 && c->sites[o.drop_site].channels != bufs[o.out].C) {
      fail(nullptr, "op %d: drop site %d has %d channels, output has %d", i, o.drop_site,
           c->sites[o.drop_site].channels, bufs[o.out].C);
      uda_destroy(c);
      return 1;
    }
    if (o.kind == UDA_OP_MBX && o.se_scale >= 0 &&
        (!okbuf(o.se_scale) || o.se_mid != 16 || bufs[o.in[0]].C > 32 || bufs[o.in[0]].C % 8 || bufs[o.se_scale].C != bufs[o.in[0]].C ||
         o.se_w1_off < 0 || o.se_b1_off < 0 || o.se_w2_off < 0)) {
      fail(nullptr, "op %d: fused projection needs a [rows, C0 <= 32] gate, a 16-channel projection and its kernel / BN offsets", i);
      uda_destroy(c);
      return 1;
    }
    if (o.act < UDA_ACT_NONE || o.act > UDA_ACT_MISH || o.fuse_act < UDA_ACT_NONE || o.fuse_act > UDA_ACT_MISH) {
      fail(nullptr, "op %d: unknown activation %d / %d", i, o.act, o.fuse_act);
      uda_destroy(c);
      return 1;
    }
    if (o.kind == UDA_OP_MBX && o.act != UDA_ACT_SWISH) {
      // the fused MBConv kernels fold the swish into their BN scales (swish_core / swish_folded): the planner keeps any
      // other act_type on the unfused expand / depthwise ops
      fail(nullptr, "op %d: the fused MBConv front half is a swish kernel (act %d)", i, o.act);
      uda_destroy(c);
      return 1;
    }
    if (o.kind == UDA_OP_SEP && o.drop_site2 >= 0) {
      // deferred dropout site of the producer (plan.py: first head layer under head-only MC): keep-scales of the INPUT channels
      const uda_buf_desc_t& ib = bufs[o.in[0]];
      const uda_buf_desc_t& ob = bufs[o.out];
      if (o.fuse_in || ib.per_sample || (!ob.per_sample && T > 1) || c->sites[o.drop_site2].channels != ib.C) {
        fail(nullptr, "op %d: a deferred input dropout site needs a plain separable conv with a per-image input of %d channels and a "
                      "per-sample output", i, c->sites[o.drop_site2].channels);
        uda_destroy(c);
        return 1;
      }
    }
    if (o.fuse_in && o.kind != UDA_OP_SEP) {
      fail(nullptr, "op %d: fuse_in is a separable-conv field (kind %d)", i, o.kind);
      uda_destroy(c);
      return 1;
    }
    if (o.kind == UDA_OP_SEP && (o.w_off < 0 || o.w2_off < 0 || !sep_supported(bufs[o.in[0]].C, bufs[o.out].C) ||
                                 (!o.fuse_in && (bufs[o.in[0]].H != bufs[o.out].H || bufs[o.in[0]].W != bufs[o.out].W)))) {
      fail(nullptr, "op %d: fused separable conv %d->%d unsupported (needs both kernels, C %% 8 == 0, 16 <= C <= 128, same size)",
           i, bufs[o.in[0]].C, bufs[o.out].C);
      uda_destroy(c);
      return 1;
    }
    const int64_t offs[] = {o.w_off, o.bias_off, o.bn_scale_off, o.bn_shift_off, o.se_w1_off, o.se_b1_off, o.se_w2_off, o.se_b2_off,
                            (o.kind == UDA_OP_MBX || o.kind == UDA_OP_SEP) ? o.w2_off : -1, o.kind == UDA_OP_MBX ? o.bn2_scale_off : -1,
                            o.kind == UDA_OP_MBX ? o.bn2_shift_off : -1};
    for (int64_t off : offs)
      if (off >= n_weights) { fail(nullptr, "op %d: weight offset %lld beyond blob (%lld)", i, (long long)off, (long long)n_weights); uda_destroy(c); return 1; }
    if ((o.kind == UDA_OP_PW || o.kind == UDA_OP_DW || o.kind == UDA_OP_MBX) && (bufs[o.in[0]].C % 4 || bufs[o.out].C % 1)) {
      fail(nullptr, "op %d: channel count %d not a multiple of 4", i, bufs[o.in[0]].C);
      uda_destroy(c);
      return 1;
    }
  }
  for (int i = 0; i < n_bufs; ++i) {
    const uda_buf_desc_t& b = bufs[i];
    if (b.kind == 0) {
      const int64_t rows = (int64_t)m.chunk_images * (b.per_sample ? T : 1);
      const int64_t end = b.offset + rows * b.H * b.W * b.C;
      if (b.offset < 0 || end > m.arena_floats) {
        fail(nullptr, "buffer %d [%dx%dx%d] exceeds the arena (%lld > %lld)", i, b.H, b.W, b.C,
             (long long)end, (long long)m.arena_floats);
        uda_destroy(c);
        return 1;
      }
    }
  }

  CK(dalloc(&c->d_weights, (size_t)n_weights));
  CK(hipMemcpy(c->d_weights, weights, (size_t)n_weights * sizeof(float), hipMemcpyHostToDevice));
  {
    // split-precision copies of every 1x1 kernel, packed once (host) in B-fragment order
    std::string perr;
    c->pw_parts = parse_pw_scheme(&perr);
    if (c->pw_parts < 0) { fail(nullptr, "uda_create: %s", perr.c_str()); uda_destroy(c); return 1; }
    c->wsplit_off.assign(n_ops, -1);
    c->wpar_off.assign(n_ops, -1);
    c->wscheme.assign(n_ops, c->pw_parts);
    c->wunscale.assign(n_ops, 1.0f);
    CK(dalloc(&c->d_oor, 2 * ((size_t)n_ops + 1)));
    CK(hipMemset(c->d_oor, 0, 2 * ((size_t)n_ops + 1) * sizeof(unsigned)));
    c->oor_cur = c->d_oor;
    c->wovr.assign(n_ops, nullptr);
    c->wovr_par.assign(n_ops, -1);
    c->h_weights.assign(weights, weights + n_weights);
    const char* em = getenv("UDA_MBX_BF16");
    const bool mbx_bf16 = em ? atoi(em) != 0 : true;
    if (c->pw_parts) {
      std::vector<uint16_t> packed;
      for (int i = 0; i < n_ops; ++i) {
        const uda_op_t& o = ops[i];
        if (o.w_off < 0) continue;
        const int K = bufs[o.in[0]].C, Nn = bufs[o.out].C;
        const size_t at = packed.size();
        // diagnostic: UDA_F16_KINDS=pw,sep,mbx (any subset) keeps fp16 pieces for those op kinds only, three bf16 pieces elsewhere
        static const char* f16_kinds = getenv("UDA_F16_KINDS");
        const bool kind_f16 = !f16_kinds || strstr(f16_kinds, o.kind == UDA_OP_PW ? "pw" : (o.kind == UDA_OP_SEP ? "sep" : "mbx")) != nullptr;
        if (o.kind == UDA_OP_PW || o.kind == UDA_OP_SEP) {
          // fp16 pieces: the kernel times a power of two that puts its largest entry in [2^13, 2^14) - every low piece of a
          // weight that matters is then a normal fp16 number; the epilogue multiplies the accumulator by the inverse (exact)
          float scale = 1.0f;
          int sch = c->pw_parts;
          if (sch == UDA_SPLIT_F16X2 && !kind_f16) sch = UDA_SPLIT_BF16X3;
          if (sch == UDA_SPLIT_F16X2) {
            scale = split_weight_scale(weights + o.w_off, (size_t)K * Nn);
            c->wunscale[i] = 1.0f / scale;
            ++c->n_f16_ops;
          }
          c->wscheme[i] = sch;
          const size_t w_elems = (pwb_packed_elems(K, Nn, sch) + 7) / 8 * 8;
          // fp16 pieces, separable conv: the A operand is the depthwise result, whose magnitude nothing bounds from below (a
          // BiFPN / head feature times nine small taps: rms 0.01-0.1 under the reference initialisers - its low pieces would
          // all be subnormal, 2^-25 absolute instead of 2^-22 relative: measured 3.2e-6 / 6.2e-6 relative rms on the heads of
          // D0 / D2, all of it from these ops; with the shift 2.4e-7 / 3.1e-7, the float32 floor of 2.3e-7 / 3.3e-7 that three
          // bf16 pieces reach).  The depthwise kernel is therefore applied times 2^UDA_F16_SEP_SHIFT (default 6; a host-side copy
          // of the 9 x C taps behind the packed 1x1 kernel, no device cost) and the epilogue's factor carries the inverse;
          // the range flag watches the scaled values (65504 / 64 = 1023 for the depthwise result itself).
          size_t dw_fl = 0;
          float ascale = 1.0f;
          if (o.kind == UDA_OP_SEP && sch == UDA_SPLIT_F16X2 && o.w2_off >= 0) {
            static const int shift = getenv("UDA_F16_SEP_SHIFT") ? atoi(getenv("UDA_F16_SEP_SHIFT")) : 6;
            ascale = ldexpf(1.0f, shift < 0 ? 0 : (shift > 12 ? 12 : shift));
            dw_fl = (size_t)9 * K;
          }
          packed.resize(at + w_elems + 2 * dw_fl);
          pwb_pack_weights(weights + o.w_off, K, Nn, sch, packed.data() + at, scale);
          if (dw_fl) {
            std::vector<float> wd(dw_fl);
            for (size_t j = 0; j < dw_fl; ++j) wd[j] = weights[o.w2_off + j] * ascale;
            memcpy(packed.data() + at + w_elems, wd.data(), dw_fl * sizeof(float));
            c->wpar_off[i] = (int64_t)(at + w_elems);
            c->wunscale[i] /= ascale;
          }
        } else if (o.kind == UDA_OP_MBX && mbx_bf16 && o.bn_scale_off >= 0 && o.bn_shift_off >= 0 &&
                   (mbxb_supported(o.se_scale >= 0 ? o.se_mid : K, Nn, o.k, o.stride) || mbxd_supported(K, Nn, o.k, o.stride))) {
          if (o.w2_off < 0 || o.bn2_scale_off < 0 || o.bn2_shift_off < 0) continue;
          const bool fuse0 = o.se_scale >= 0;      // the previous block's projection is computed in this op's prologue
          const int Ke = fuse0 ? o.se_mid : K;     // input channels of the expand
          // [split expand weights | 16-byte aligned float block of the depthwise-side operands | (fuse0) projection block]
          // fp16 pieces in a fused MBConv op: the expand accumulator feeds the swish directly, so the packed kernel (times
          // the BN scale, with the BN shift row) cannot carry a power-of-two factor.  It keeps fp16 pieces when its entries
          // sit where two pieces resolve them (largest below 2^15, rms at least 2^-6: a low piece that is subnormal resolves
          // 2^-25 absolutely); otherwise THIS op keeps three bf16 pieces - decided here, per op, never silently degraded.
          int sch = c->pw_parts;
          if (sch == UDA_SPLIT_F16X2) {
            std::vector<uint16_t> probe(mbxb_packed_elems(Ke, Nn, UDA_SPLIT_F16X2));
            float st[2] = {0.f, 0.f};
            mbxb_pack_weights(weights + o.w_off, weights + o.bn_scale_off, weights + o.bn_shift_off, Ke, Nn, probe.data(), fuse0, UDA_SPLIT_F16X2, st);
            static const float min_rms = getenv("UDA_F16_MIN_RMS") ? (float)atof(getenv("UDA_F16_MIN_RMS")) : 0.015625f;
            if (!(st[0] < 32768.0f) || !(st[1] >= min_rms) || !kind_f16) { sch = UDA_SPLIT_BF16X3; ++c->n_f16_demoted; }
            else ++c->n_f16_ops;
          }
          c->wscheme[i] = sch;
          const size_t we_elems = (mbxb_packed_elems(Ke, Nn, sch) + 7) / 8 * 8;
          const size_t par_fl = mbx_par_floats(Nn, o.k);
          const size_t proj_fl = fuse0 ? 32 * 32 + 32 : 0;
          packed.resize(at + we_elems + 2 * (par_fl + proj_fl));
          mbxb_pack_weights(weights + o.w_off, weights + o.bn_scale_off, weights + o.bn_shift_off, Ke, Nn, packed.data() + at, fuse0, sch);
          std::vector<float> par(par_fl + proj_fl);
          mbx_pack_params(weights + o.w2_off, weights + o.bn2_scale_off, weights + o.bn2_shift_off, Nn, o.k, par.data());
          if (fuse0) mbxb_pack_proj(weights + o.se_w1_off, weights + o.se_b1_off, weights + o.se_w2_off, K, Ke, par.data() + par_fl);
          memcpy(packed.data() + at + we_elems, par.data(), par.size() * sizeof(float));
          c->wpar_off[i] = (int64_t)(at + we_elems);
        } else {
          continue;
        }
        c->wsplit_off[i] = (int64_t)at;
      }
      // LDS budget of every fused launch, checked now and by name (a template / shape pair that asks for more than a CU has
      // would otherwise surface as a refused launch in the middle of the first run: round 3, D2 under six-term products)
      for (int i = 0; i < n_ops; ++i) {
        if (c->wsplit_off[i] < 0) continue;
        const uda_op_t& o = ops[i];
        const int K = bufs[o.in[0]].C, Nn = bufs[o.out].C;
        size_t lds = 0;
        if (o.kind == UDA_OP_MBX) lds = mbx_lds_bytes(o.se_scale >= 0 ? o.se_mid : K, Nn, o.k, o.stride, c->wscheme[i], bufs[o.out].H, bufs[o.out].W);
        else if (o.kind == UDA_OP_SEP) {
          lds = o.fuse_in ? sepf_lds_bytes(K, Nn, c->wscheme[i]) : sep_lds_bytes(K, Nn, c->wscheme[i]);
          if (o.fuse_in && !sepf_supported(K, Nn, c->wscheme[i])) {
            fail(nullptr, "op %d: separable conv %d -> %d (split scheme %d) has no fused-input kernel (planner: plan.sepf_supported)", i, K, Nn, c->wscheme[i]);
            uda_destroy(c);
            return 1;
          }
        }
        if (lds > (size_t)160 * 1024) {
          fail(nullptr, "op %d (%s, %d -> %d channels, k %d, stride %d, split scheme %d): its launch needs %zu bytes of LDS, a gfx950 "
                        "CU has 163840", i, o.kind == UDA_OP_MBX ? "fused MBConv front half" : "fused separable conv", K, Nn, o.k, o.stride,
               c->wscheme[i], lds);
          uda_destroy(c);
          return 1;
        }
      }
      CK(dalloc(&c->d_wsplit, packed.size()));
      if (!packed.empty())
        CK(hipMemcpy(c->d_wsplit, packed.data(), packed.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
  }
  // The arena starts out as zeros and ends in a guard: some kernels read a few floats past the end of their input (the
  // k-padding of the last pixel's matrix fragment, a dead lane's clamped window) and multiply them by zero weights - that must
  // never meet a NaN pattern in fresh memory (one NaN pixel reaches a whole image through the squeeze-excite mean: round 5,
  // DESIGN 4.1) nor the end of the allocation.
  constexpr size_t ARENA_GUARD = 4096;      // floats
  CK(dalloc(&c->d_arena, (size_t)m.arena_floats + ARENA_GUARD));
  CK(hipMemset(c->d_arena, 0, ((size_t)m.arena_floats + ARENA_GUARD) * sizeof(float)));
  {
    const char* e = getenv("UDA_LANES");
    c->n_lanes = e ? atoi(e) : 1;   // 2 overlaps consecutive chunks on two streams: +4 % throughput, but per-kernel timings then include the sharing
    if (c->n_lanes < 1) c->n_lanes = 1;
    if (c->n_lanes > 2) c->n_lanes = 2;
    if (m.max_images <= m.chunk_images) c->n_lanes = 1;     // a single chunk per run: nothing to overlap
    c->lane_stream[0] = c->stream;
    c->lane_arena[0] = c->d_arena;
    CK(hipEventCreateWithFlags(&c->ev_start, hipEventDisableTiming));
    const char* ep = getenv("UDA_POST_OVERLAP");
    c->post_overlap = ep ? atoi(ep) : 1;
    {   // highest priority: the post-process is a chain of ~200 tiny dependent launches that must slip in
        // between the workgroups of the conv kernels instead of queueing behind them
      int lo = 0, hi = 0;
      CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
      CK(hipStreamCreateWithPriority(&c->post_stream, hipStreamNonBlocking, hi));
    }
    CK(hipEventCreateWithFlags(&c->ev_post, hipEventDisableTiming));
    c->ev_chunk.assign((size_t)(m.max_images + m.chunk_images - 1) / m.chunk_images, nullptr);
    for (auto& e : c->ev_chunk) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (int l = 0; l < c->n_lanes; ++l) CK(hipEventCreateWithFlags(&c->ev_done[l], hipEventDisableTiming));
    for (int l = 1; l < c->n_lanes; ++l) {
      CK(hipStreamCreateWithFlags(&c->lane_stream[l], hipStreamNonBlocking));
      CK(dalloc(&c->lane_arena[l], (size_t)m.arena_floats + ARENA_GUARD));
      CK(hipMemset(c->lane_arena[l], 0, ((size_t)m.arena_floats + ARENA_GUARD) * sizeof(float)));
    }
  }

  c->a_off[0] = 0;
  for (int l = 0; l < m.num_levels; ++l)
    c->a_off[l + 1] = c->a_off[l] + m.level_h[l] * m.level_w[l] * m.anchors_per_loc;
  c->A_tot = c->a_off[m.num_levels];
  CK(dalloc(&c->d_anchors, (size_t)c->A_tot * 4));
  CK(hipMemcpy(c->d_anchors, anchors, (size_t)c->A_tot * 4 * sizeof(float), hipMemcpyHostToDevice));

  const size_t N = (size_t)m.max_images;
  CK(dalloc(&c->d_images, N * m.image_h * m.image_w * 3));
  CK(dalloc(&c->d_scales, N));
  c->h_scales.assign(N, 1.0f);

  // dropout sites
  c->sum_site_ch = 0;
  c->max_c4 = 0;
  std::vector<int32_t> ch;
  std::vector<float> rate;
  for (auto& s : c->sites) {
    c->sum_site_ch += s.channels;
    ch.push_back(s.channels);
    rate.push_back(s.rate);
    if ((s.channels + 3) / 4 > c->max_c4) c->max_c4 = (s.channels + 3) / 4;
  }
  c->mask_cap = c->sum_site_ch * (int64_t)N * T;
  c->site_off.assign(c->sites.size() + 1, 0);
  if (!c->sites.empty()) {
    CK(dalloc(&c->d_masks, (size_t)c->mask_cap));
    CK(dalloc(&c->d_site_off, c->sites.size()));
    CK(dalloc(&c->d_site_ch, c->sites.size()));
    CK(dalloc(&c->d_site_rate, c->sites.size()));
    CK(hipMemcpy(c->d_site_ch, ch.data(), ch.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    CK(hipMemcpy(c->d_site_rate, rate.data(), rate.size() * sizeof(float), hipMemcpyHostToDevice));
  }

  // head outputs
  c->cls_ch = m.anchors_per_loc * m.num_classes;
  c->box_ch = m.anchors_per_loc * (m.loss_attenuation ? 8 : 4);
  for (int l = 0; l < m.num_levels; ++l) {
    const size_t hw = (size_t)m.level_h[l] * m.level_w[l];
    CK(dalloc(&c->d_cls[l], N * (m.cls_stacked ? T : 1) * hw * c->cls_ch));
    CK(dalloc(&c->d_box[l], N * (m.box_stacked ? T : 1) * hw * c->box_ch));
  }

  // candidates + nms + outputs
  if (m.max_nms_inputs > 0) {
    if (m.max_nms_inputs > 8192 || (int64_t)m.max_nms_inputs > (int64_t)c->A_tot * m.num_classes) {
      fail(nullptr, "uda_create: max_nms_inputs %d must be <= min(8192, anchors*classes = %lld)", m.max_nms_inputs,
           (long long)c->A_tot * m.num_classes);
      uda_destroy(c);
      return 1;
    }
    c->Kc = m.max_nms_inputs;
    CK(dalloc(&c->d_clsmean, N * (size_t)c->A_tot * m.num_classes));
    CK(dalloc(&c->d_cand_flat, N * (size_t)c->Kc));
    CK(hipMalloc(&c->d_topk_ws, topk_workspace_bytes((int)N, c->Kc)));
  } else {
    c->Kc = c->A_tot;
  }
  const size_t K = (size_t)c->Kc, M = (size_t)m.max_output_size, C = (size_t)m.num_classes;
  CK(dalloc(&c->d_cboxes, N * K * 4));
  CK(dalloc(&c->d_cscores, N * K));
  CK(dalloc(&c->d_cclasses, N * K));
  CK(dalloc(&c->d_clogits, N * K * C));
  if (m.has_uncert && m.cls_stacked) CK(dalloc(&c->d_ucls, N * K * C));
  if (m.has_uncert && m.loss_attenuation) CK(dalloc(&c->d_ual, N * K * 4));
  if (m.has_uncert && m.box_stacked) CK(dalloc(&c->d_uep, N * K * 4));
  CK(alloc_nms_ws(c->ws[0], N, K, M));
  CK(dalloc(&c->d_coop_bar, N * nms_coop_slot_words((int)M)));
  CK(dalloc(&c->d_coop_err, 1));
  CK(hipMemset(c->d_coop_err, 0, sizeof(int)));
  if (prefix_target() > 0 && (int)K > solo_limit() && K > (size_t)2 * prefix_target() && M <= 128)
    CK(alloc_prefix_ws(c->pfx, N, 2 * prefix_target(), M));
  CK(dalloc(&c->d_oboxes, N * M * 12));
  CK(dalloc(&c->d_oscores, N * M));
  CK(dalloc(&c->d_oclasses, N * M * (1 + C)));
  CK(dalloc(&c->d_ologits, N * M * C));
  CK(dalloc(&c->d_ovalid, N));
#undef CK
  *out = c;
  return 0;
}

// ------------------------------------------------------------------------------------ inputs
// Per-image geometry of the resize (dataloader.py:123-135): scale = min(H/h, W/w) in float32, scaled size =
// int(h*scale), int(w*scale), image_scale = 1/scale; the bilinear sampler maps an output pixel with the ratios
// raw / scaled (half-pixel centres).
static PreGeo geo_for_raw(const uda_model_t& m, int h, int w, size_t off) {
  const float sy = (float)m.image_h / (float)h;
  const float sx = (float)m.image_w / (float)w;
  const float s = sx < sy ? sx : sy;
  PreGeo g{};
  g.off = (unsigned long long)off;
  g.h = h; g.w = w;
  g.sh = (int)((float)h * s);
  g.sw = (int)((float)w * s);
  if (g.sh > m.image_h) g.sh = m.image_h;
  if (g.sw > m.image_w) g.sw = m.image_w;
  g.scale_y = (float)h / (float)g.sh;
  g.scale_x = (float)w / (float)g.sw;
  g.inv_scale = 1.0f / s;
  return g;
}

static int ensure_input_streams(uda_ctx* c) {
  if (!c->copy_stream) HIPC(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  for (int i = 0; i < 2; ++i) {
    if (!c->u8[i].ev) HIPC(c, hipEventCreateWithFlags(&c->u8[i].ev, hipEventDisableTiming));
    if (!c->ev_pre_done[i]) HIPC(c, hipEventCreateWithFlags(&c->ev_pre_done[i], hipEventDisableTiming));
  }
  return 0;
}

// Fills slot `si` with n images and starts their upload on stream `st`.  `images` (host or device, uniform size) or
// `ragged` (host pointers, per-image sizes): host bytes are first gathered into the slot's pinned staging buffer, so the
// DMA never reads pageable memory (a pageable hipMemcpyAsync is a synchronous, chunked copy) and the caller's arrays are
// free as soon as this returns.
static int fill_slot(uda_ctx* c, int si, const void* images, const uint8_t* const* ragged, int n, const int32_t* hs, const int32_t* ws,
                     hipMemcpyKind kind, hipStream_t st, bool staged) {
  if (n < 1 || n > c->model.max_images) return fail(c, "set_images: n=%d outside [1, %d]", n, c->model.max_images);
  HIPC(c, hipSetDevice(c->device));
  if (ensure_input_streams(c)) return 1;
  uda_ctx::U8Slot& sl = c->u8[si];
  sl.geo.resize(n);
  sl.scales.resize(n);
  size_t total = 0;
  for (int i = 0; i < n; ++i) {
    const int h = hs[ragged ? i : 0], w = ws[ragged ? i : 0];
    if (h < 1 || w < 1) return fail(c, "set_images: bad size %dx%d of image %d", h, w, i);
    if (ragged && !ragged[i]) return fail(c, "set_images: NULL image %d", i);
    sl.geo[i] = geo_for_raw(c->model, h, w, total);
    sl.scales[i] = sl.geo[i].inv_scale;
    total += (size_t)h * w * 3;
  }
  // the slot's previous batch must have left: its upload done and the preprocess kernel that read it finished
  HIPC(c, hipEventSynchronize(sl.ev));
  HIPC(c, hipEventSynchronize(c->ev_pre_done[si]));
  // device / pinned layout: [geometry table, max_images entries, padded to 256 B][images back to back] - ONE copy moves both
  const size_t hdr = ((size_t)c->model.max_images * sizeof(PreGeo) + 255) & ~(size_t)255;
  if (hdr + total > sl.cap) {
    if (sl.d) HIPC(c, hipFree(sl.d));
    sl.d = nullptr; sl.cap = 0;
    HIPC(c, hipMalloc((void**)&sl.d, hdr + total + 16));      // (+16: the uint8 stem reads 12-byte window rows, 3 bytes past the last pixel)
    sl.cap = hdr + total;
  }
  sl.d_geo = (PreGeo*)sl.d;
  sl.d_img = sl.d + hdr;
  if (kind == hipMemcpyDeviceToDevice) {
    HIPC(c, hipMemcpyAsync(sl.d_img, images, total, kind, st));
    HIPC(c, hipMemcpyAsync(sl.d_geo, sl.geo.data(), (size_t)n * sizeof(PreGeo), hipMemcpyHostToDevice, st));
  } else if (!staged) {
    // the feed of a synchronous serve(): the runtime's own pageable-memory path (it stages and pipelines in chunks: 94 MB in
    // ~2 ms; gathering into the pinned buffer first costs a host memcpy that nothing hides here: 4.6 ms)
    if (ragged) {
      for (int i = 0; i < n; ++i)
        HIPC(c, hipMemcpyAsync(sl.d_img + sl.geo[i].off, ragged[i], (size_t)sl.geo[i].h * sl.geo[i].w * 3, hipMemcpyHostToDevice, st));
    } else {
      HIPC(c, hipMemcpyAsync(sl.d_img, images, total, hipMemcpyHostToDevice, st));
    }
    HIPC(c, hipMemcpyAsync(sl.d_geo, sl.geo.data(), (size_t)n * sizeof(PreGeo), hipMemcpyHostToDevice, st));
  } else {
    // prefetch: the host bytes are gathered into the slot's pinned buffer (this memcpy runs while the GPU computes the
    // current batch) and leave by ONE DMA on the copy stream that never blocks on pageable memory
    if (hdr + total > sl.pcap) {
      if (sl.pinned) HIPC(c, hipHostFree(sl.pinned));
      sl.pinned = nullptr; sl.pcap = 0;
      HIPC(c, hipHostMalloc((void**)&sl.pinned, hdr + total, hipHostMallocDefault));
      sl.pcap = hdr + total;
    }
    memcpy(sl.pinned, sl.geo.data(), (size_t)n * sizeof(PreGeo));
    if (ragged) {
      for (int i = 0; i < n; ++i) memcpy(sl.pinned + hdr + sl.geo[i].off, ragged[i], (size_t)sl.geo[i].h * sl.geo[i].w * 3);
    } else {
      memcpy(sl.pinned + hdr, images, total);
    }
    HIPC(c, hipMemcpyAsync(sl.d, sl.pinned, hdr + total, hipMemcpyHostToDevice, st));
  }
  HIPC(c, hipEventRecord(sl.ev, st));
  sl.n = n;
  sl.valid = true;
  ++sl.gen;
  return 0;
}

// Makes slot `si` the input of the next uda_run: the main stream waits for its upload, the image scales follow.
static int make_current(uda_ctx* c, int si) {
  uda_ctx::U8Slot& sl = c->u8[si];
  if (!sl.valid) return fail(c, "no uint8 batch in the input slot");
  HIPC(c, hipStreamWaitEvent(c->stream, sl.ev, 0));
  c->cur = si;
  c->n_images = sl.n;
  c->have_u8 = true;
  for (int i = 0; i < sl.n; ++i) c->h_scales[i] = sl.scales[i];
  HIPC(c, hipMemcpyAsync(c->d_scales, c->h_scales.data(), sl.n * sizeof(float), hipMemcpyHostToDevice, c->stream));
  return 0;
}

// entry points that rewrite what a pipelined run's post-process reads (head buffers, image count) refuse to run beside it
static int no_async(uda_ctx* c, const char* what) {
  if (c->as[0].open || c->as[1].open) return fail(c, "%s: a pipelined run (uda_run_async) is in flight - uda_collect it first", what);
  return 0;
}

extern "C" int uda_set_images_u8(uda_ctx_t* c, const uint8_t* images, int32_t n, int32_t h, int32_t w) {
  if (!c || !images) return c ? fail(c, "set_images_u8: NULL images") : 1;
  const int si = c->cur;
  const int rc = fill_slot(c, si, images, nullptr, n, &h, &w, hipMemcpyHostToDevice, c->stream, false);
  return rc ? rc : make_current(c, si);
}

extern "C" int uda_set_images_u8_device(uda_ctx_t* c, const void* images_dev, int32_t n, int32_t h, int32_t w) {
  if (!c || !images_dev) return c ? fail(c, "set_images_u8_device: NULL images") : 1;
  const int si = c->cur;
  const int rc = fill_slot(c, si, images_dev, nullptr, n, &h, &w, hipMemcpyDeviceToDevice, c->stream, false);
  return rc ? rc : make_current(c, si);
}

extern "C" int uda_set_images_u8_ragged(uda_ctx_t* c, const uint8_t* const* images, int32_t n, const int32_t* h, const int32_t* w) {
  if (!c || !images || !h || !w) return c ? fail(c, "set_images_u8_ragged: NULL argument") : 1;
  const int si = c->cur;
  const int rc = fill_slot(c, si, nullptr, images, n, h, w, hipMemcpyHostToDevice, c->stream, false);
  return rc ? rc : make_current(c, si);
}

extern "C" int uda_prefetch_images_u8(uda_ctx_t* c, const uint8_t* images, int32_t n, int32_t h, int32_t w) {
  if (!c || !images) return c ? fail(c, "prefetch_images_u8: NULL images") : 1;
  if (ensure_input_streams(c)) return 1;
  return fill_slot(c, c->cur ^ 1, images, nullptr, n, &h, &w, hipMemcpyHostToDevice, c->copy_stream, true);
}

extern "C" int uda_prefetch_images_u8_ragged(uda_ctx_t* c, const uint8_t* const* images, int32_t n, const int32_t* h, const int32_t* w) {
  if (!c || !images || !h || !w) return c ? fail(c, "prefetch_images_u8_ragged: NULL argument") : 1;
  if (ensure_input_streams(c)) return 1;
  return fill_slot(c, c->cur ^ 1, nullptr, images, n, h, w, hipMemcpyHostToDevice, c->copy_stream, true);
}

extern "C" int uda_input_u8_device(uda_ctx_t* c, const void** images_dev, int32_t* n, int32_t* h, int32_t* w) {
  if (!c || !images_dev || !n || !h || !w) return c ? fail(c, "input_u8_device: NULL argument") : 1;
  const uda_ctx::U8Slot& sl = c->u8[c->cur];
  if (!c->have_u8 || !sl.valid) return fail(c, "input_u8_device: no uint8 batch is set");
  for (int i = 1; i < sl.n; ++i)
    if (sl.geo[i].h != sl.geo[0].h || sl.geo[i].w != sl.geo[0].w) return fail(c, "input_u8_device: the batch has several raw sizes");
  HIPC(c, hipEventSynchronize(sl.ev));      // the upload has landed: another handle's stream may read the buffer now
  *images_dev = sl.d_img; *n = sl.n; *h = sl.geo[0].h; *w = sl.geo[0].w;
  return 0;
}

extern "C" int uda_swap_prefetched(uda_ctx_t* c) {
  if (!c) return 1;
  if (!c->u8[c->cur ^ 1].valid) return fail(c, "swap_prefetched: nothing was prefetched");
  HIPC(c, hipSetDevice(c->device));
  c->u8[c->cur].valid = false;
  return make_current(c, c->cur ^ 1);
}

extern "C" int uda_set_images_f32(uda_ctx_t* c, const float* images, int32_t n, const float* image_scales) {
  if (!c || !images) return c ? fail(c, "set_images_f32: NULL images") : 1;
  if (n < 1 || n > c->model.max_images) return fail(c, "set_images_f32: n=%d outside [1, %d]", n, c->model.max_images);
  HIPC(c, hipSetDevice(c->device));
  const size_t fl = (size_t)n * c->model.image_h * c->model.image_w * 3;
  HIPC(c, hipMemcpyAsync(c->d_images, images, fl * sizeof(float), hipMemcpyHostToDevice, c->stream));
  for (int i = 0; i < n; ++i) c->h_scales[i] = image_scales ? image_scales[i] : 1.0f;
  HIPC(c, hipMemcpyAsync(c->d_scales, c->h_scales.data(), n * sizeof(float), hipMemcpyHostToDevice, c->stream));
  c->n_images = n;
  c->have_u8 = false;
  c->stem_from_u8 = false;
  ++c->f32_gen;
  return 0;
}

extern "C" int uda_set_dropout_seed(uda_ctx_t* c, uint64_t seed) {
  if (!c) return 1;
  c->seed = seed;
  c->masks_injected = false;
  return 0;
}

extern "C" int uda_set_dropout_image_offset(uda_ctx_t* c, int64_t first_image) {
  if (!c) return 1;
  if (first_image < 0) return fail(c, "set_dropout_image_offset: negative offset");
  c->image_offset = first_image;
  return 0;
}

extern "C" int uda_set_dropout_sample_shard(uda_ctx_t* c, int32_t t_first, int32_t t_stride, int32_t t_total) {
  if (!c) return 1;
  const int T = c->model.mc_samples;
  if (t_total == 0) { c->t_first = 0; c->t_stride = 1; c->t_total = 0; return 0; }
  if (t_first < 0 || t_stride < 1 || t_total < T || t_first + (int64_t)(T - 1) * t_stride >= t_total)
    return fail(c, "set_dropout_sample_shard: samples %d + j * %d (j < %d) do not lie inside %d", t_first, t_stride, T, t_total);
  c->t_first = t_first; c->t_stride = t_stride; c->t_total = t_total;
  return 0;
}

extern "C" int uda_set_dropout_masks(uda_ctx_t* c, const float* masks, int64_t n_floats) {
  if (!c || !masks) return c ? fail(c, "set_dropout_masks: NULL") : 1;
  if (n_floats > c->mask_cap || (c->sum_site_ch && n_floats % c->sum_site_ch))
    return fail(c, "set_dropout_masks: %lld floats is not rows*%lld (capacity %lld)", (long long)n_floats,
                (long long)c->sum_site_ch, (long long)c->mask_cap);
  HIPC(c, hipSetDevice(c->device));
  if (n_floats) HIPC(c, hipMemcpyAsync(c->d_masks, masks, n_floats * sizeof(float), hipMemcpyHostToDevice, c->stream));
  c->masks_injected = true;
  ++c->masks_gen;
  c->masks_rows = c->sum_site_ch ? (int)(n_floats / c->sum_site_ch) : 0;
  return 0;
}

extern "C" int uda_get_dropout_masks(uda_ctx_t* c, float* masks, int64_t n_floats) {
  if (!c || !masks) return 1;
  if (n_floats > c->mask_cap) return fail(c, "get_dropout_masks: too many floats");
  HIPC(c, hipSetDevice(c->device));
  HIPC(c, hipStreamSynchronize(c->stream));
  if (n_floats) HIPC(c, hipMemcpy(masks, c->d_masks, n_floats * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

// ------------------------------------------------------------------------------------ op execution
struct ChunkView {
  uda_ctx* c;
  int i0, nc;
  int lane = 0;
  hipStream_t stream() const { return c->lane_stream[lane]; }
  int rows(const uda_buf_desc_t& b) const { return nc * (b.per_sample ? c->model.mc_samples : 1); }
  float* ptr(int id) const {
    const uda_buf_desc_t& b = c->bufs[id];
    const size_t per = (size_t)b.H * b.W * b.C;
    const int T = c->model.mc_samples;
    switch (b.kind) {
      case 1: return c->d_images + (size_t)i0 * per;
      case 2: return c->d_cls[b.level] + (size_t)i0 * (b.per_sample ? T : 1) * per;
      case 3: return c->d_box[b.level] + (size_t)i0 * (b.per_sample ? T : 1) * per;
      default: return c->lane_arena[lane] + b.offset;
    }
  }
  const float* wt(int64_t off) const { return off < 0 ? nullptr : c->d_weights + off; }
  const float* mask(int site) const {
    if (site < 0) return nullptr;
    return c->d_masks + c->site_off[site] + (size_t)i0 * c->model.mc_samples * c->sites[site].channels;
  }
  int div(const uda_buf_desc_t& in, const uda_buf_desc_t& out) const {
    return (out.per_sample && !in.per_sample) ? c->model.mc_samples : 1;
  }
};

// packed 1x1 weights / parameter block of op oi: its slice of d_wsplit, or its re-packed copy after a range demotion
static inline const uint16_t* wsplit_of(const uda_ctx* c, int oi) { return c->wovr[oi] ? c->wovr[oi] : c->d_wsplit + c->wsplit_off[oi]; }
static inline bool has_wpar(const uda_ctx* c, int oi) { return c->wovr[oi] ? c->wovr_par[oi] >= 0 : c->wpar_off[oi] >= 0; }
static inline const uint16_t* wpar_of(const uda_ctx* c, int oi) {
  return c->wovr[oi] ? c->wovr[oi] + c->wovr_par[oi] : c->d_wsplit + c->wpar_off[oi];
}

// FuseArgs of op oi's inputs (FUSE / POOL, and SEP with fuse_in: the BiFPN fusion in front of the node's separable conv)
static int fill_fuse_args(uda_ctx* c, const ChunkView& v, int oi, FuseArgs& a) {
  const uda_op_t& o = c->ops[oi];
  const uda_buf_desc_t& ob = c->bufs[o.out];
  a.n_in = o.n_in;
  a.H = ob.H; a.W = ob.W;
  if (o.n_in < 1 || o.n_in > UDA_MAX_FUSE_INPUTS) return fail(c, "op %d: %d fusion inputs", oi, o.n_in);
  a.C = c->bufs[o.in[0]].C;
  if (a.C % 4) return fail(c, "op %d: fuse needs channels %% 4 == 0", oi);
  for (int i = 0; i < o.n_in; ++i) {
    const uda_buf_desc_t& ib = c->bufs[o.in[i]];
    if (ib.C != a.C) return fail(c, "op %d: fuse input %d has %d channels, input 0 %d", oi, i, ib.C, a.C);
    a.in[i] = v.ptr(o.in[i]);
    a.wgt[i] = (o.kind == UDA_OP_POOL) ? 1.0f : o.fuse_w[i];
    a.mode[i] = o.resample[i];
    a.Hi[i] = ib.H; a.Wi[i] = ib.W;
    a.in_div[i] = v.div(ib, ob);
    if (a.mode[i] == UDA_RS_NONE) {
      if (ib.H != ob.H || ib.W != ob.W) return fail(c, "op %d: fuse input %d size mismatch", oi, i);
    } else if (a.mode[i] == UDA_RS_NEAREST_UP) {
      if (ib.H > ob.H || ib.W > ob.W) return fail(c, "op %d: nearest-up input %d larger than output", oi, i);
      a.sy[i] = (float)ib.H / (float)ob.H;
      a.sx[i] = (float)ib.W / (float)ob.W;
    } else if (a.mode[i] == UDA_RS_MAXPOOL) {
      const int sh_ = (ib.H - 1) / ob.H + 1, sw_ = (ib.W - 1) / ob.W + 1;
      if (sh_ != sw_) return fail(c, "op %d: non-square pooling window", oi);
      a.ps[i] = sh_;
      a.pk[i] = sh_ + 1;
      if ((ib.H + sh_ - 1) / sh_ != ob.H || (ib.W + sh_ - 1) / sh_ != ob.W)
        return fail(c, "op %d: pooled size mismatch", oi);
      a.ppt[i] = same_pad_before(ib.H, ob.H, a.pk[i], a.ps[i]);
      a.ppl[i] = same_pad_before(ib.W, ob.W, a.pk[i], a.ps[i]);
    } else {
      return fail(c, "op %d: unknown resample mode %d of input %d", oi, a.mode[i], i);
    }
  }
  return 0;
}

static int run_op(uda_ctx* c, const ChunkView& v, int oi) {
  const uda_op_t& o = c->ops[oi];
  const uda_buf_desc_t& ob = c->bufs[o.out];
  const int rows = v.rows(ob);
  ProfScope ps(c, o.kind, v.stream());
  switch (o.kind) {
    case UDA_OP_STEM: {
      const uda_buf_desc_t& ib = c->bufs[o.in[0]];
      if (ib.C != 3 || ob.C % 4) return fail(c, "op %d: stem needs 3 -> 4k channels", oi);
      if (ib.per_sample != ob.per_sample) return fail(c, "op %d: stem cannot change the sample axis", oi);
      StemArgs a{};
      a.in = v.ptr(o.in[0]);
      a.out = v.ptr(o.out);
      a.w = v.wt(o.w_off);
      a.bn_scale = v.wt(o.bn_scale_off);
      a.bn_shift = v.wt(o.bn_shift_off);
      a.H = ib.H; a.W = ib.W; a.Ho = ob.H; a.Wo = ob.W; a.Co = ob.C;
      a.pad_t = same_pad_before(ib.H, ob.H, 3, 2);
      a.pad_l = same_pad_before(ib.W, ob.W, 3, 2);
      a.rows = rows;
      a.act = o.act;
      if (c->stem_from_u8) {
        if (ib.kind != 1 || ib.per_sample) return fail(c, "op %d: the uint8 stem reads the image buffer", oi);
        const uda_ctx::U8Slot& sl = c->u8[c->cur];
        a.u8 = sl.d_img; a.geo = sl.d_geo; a.img0 = v.i0;
        for (int k = 0; k < 3; ++k) { a.mean[k] = c->model.mean_rgb[k]; a.stdv[k] = c->model.stddev_rgb[k]; }
      }
      launch_stem(a, v.stream());
      break;
    }
    case UDA_OP_PW: {
      const uda_buf_desc_t& ib = c->bufs[o.in[0]];
      if (ib.H != ob.H || ib.W != ob.W) return fail(c, "op %d: 1x1 conv changes the spatial size", oi);
      PwArgs a{};
      a.in = v.ptr(o.in[0]);
      a.out = v.ptr(o.out);
      a.w = v.wt(o.w_off);
      a.bias = v.wt(o.bias_off);
      a.bn_scale = v.wt(o.bn_scale_off);
      a.bn_shift = v.wt(o.bn_shift_off);
      a.se = o.se_scale >= 0 ? v.ptr(o.se_scale) : nullptr;
      a.se_div = 1;
      if (o.se_scale >= 0) {
        const uda_buf_desc_t& sb = c->bufs[o.se_scale];
        if (sb.per_sample && !ob.per_sample) return fail(c, "op %d: per-sample SE gate on a per-image output", oi);
        if (!sb.per_sample && ib.per_sample) return fail(c, "op %d: per-image SE gate on a per-sample input", oi);
        a.se_div = v.div(sb, ob);
      }
      a.mask = v.mask(o.drop_site);
      a.res = o.residual >= 0 ? v.ptr(o.residual) : nullptr;
      a.HW = ob.H * ob.W; a.Cin = ib.C; a.Cout = ob.C;
      a.in_div = v.div(ib, ob);
      a.res_div = o.residual >= 0 ? v.div(c->bufs[o.residual], ob) : 1;
      a.act = o.act;
      if (c->wsplit_off[oi] >= 0) {
        a.wsplit = wsplit_of(c, oi);
        a.wparts = c->wscheme[oi];
        a.wunscale = c->wunscale[oi];
        a.oor = c->oor_cur + oi;
        if (a.wparts == UDA_SPLIT_F16X2) c->oor_armed = true;
        launch_pwb(a, rows, v.stream());
      } else {
        launch_pw(a, rows, v.stream());
      }
      break;
    }
    case UDA_OP_DW: {
      const uda_buf_desc_t& ib = c->bufs[o.in[0]];
      if (ib.C != ob.C) return fail(c, "op %d: depthwise changes channels", oi);
      if (!((o.k == 3 || o.k == 5) && (o.stride == 1 || o.stride == 2)))
        return fail(c, "op %d: depthwise k=%d s=%d unsupported", oi, o.k, o.stride);
      DwArgs a{};
      a.in = v.ptr(o.in[0]);
      a.out = v.ptr(o.out);
      a.w = v.wt(o.w_off);
      a.bn_scale = v.wt(o.bn_scale_off);
      a.bn_shift = v.wt(o.bn_shift_off);
      a.mask = v.mask(o.drop_site);
      a.H = ib.H; a.W = ib.W; a.Ho = ob.H; a.Wo = ob.W; a.C = ob.C;
      a.pad_t = same_pad_before(ib.H, ob.H, o.k, o.stride);
      a.pad_l = same_pad_before(ib.W, ob.W, o.k, o.stride);
      a.in_div = v.div(ib, ob);
      a.act = o.act;
      if (o.se_partial >= 0) {
        const int nt = dw_tiles(ob.C, ob.H, ob.W, o.k, o.stride);
        const uda_buf_desc_t& pb = c->bufs[o.se_partial];
        if ((int64_t)pb.H * pb.W != (int64_t)nt || pb.C != ob.C || pb.per_sample != ob.per_sample)
          return fail(c, "op %d: SE partial buffer [%d,%d,%d] does not match %d tiles", oi, pb.H, pb.W, pb.C, nt);
        a.se_partial = v.ptr(o.se_partial);
      }
      launch_dw(a, rows, o.k, o.stride, v.stream());
      break;
    }
    case UDA_OP_MBX: {
      const uda_buf_desc_t& ib = c->bufs[o.in[0]];
      const bool fuse0 = o.se_scale >= 0;
      const int cin = fuse0 ? o.se_mid : ib.C;     // input channels of the expand
      const bool deep = cin > 48;
      if (fuse0 && c->wsplit_off[oi] < 0) return fail(c, "op %d: the fused projection needs the split-bf16 path", oi);
      if (deep ? !(mbxd_supported(cin, ob.C, o.k, o.stride) && c->wsplit_off[oi] >= 0) : !mbx_supported(cin, ob.C, o.k, o.stride))
        return fail(c, "op %d: fused MBConv %d->%d k%d s%d unsupported", oi, cin, ob.C, o.k, o.stride);
      if (o.drop_site2 >= c->model.n_drop_sites || (o.drop_site2 >= 0 && c->sites[o.drop_site2].channels != ob.C))
        return fail(c, "op %d: bad second dropout site", oi);
      MbxArgs a{};
      a.in = v.ptr(o.in[0]);
      a.out = v.ptr(o.out);
      a.we = v.wt(o.w_off); a.sc0 = v.wt(o.bn_scale_off); a.sh0 = v.wt(o.bn_shift_off);
      a.wd = v.wt(o.w2_off); a.sc1 = v.wt(o.bn2_scale_off); a.sh1 = v.wt(o.bn2_shift_off);
      if (!a.we || !a.sc0 || !a.wd || !a.sc1) return fail(c, "op %d: fused MBConv needs both kernels and both BNs", oi);
      a.mask0 = v.mask(o.drop_site);
      a.mask1 = v.mask(o.drop_site2);
      a.H = ib.H; a.W = ib.W; a.Ho = ob.H; a.Wo = ob.W; a.Cin = cin; a.Cmid = ob.C;
      a.pad_t = same_pad_before(ib.H, ob.H, o.k, o.stride);
      a.pad_l = same_pad_before(ib.W, ob.W, o.k, o.stride);
      a.in_div = v.div(ib, ob);
      a.n_tiles = deep ? mbxd_tiles(ob.H, ob.W, o.k, o.stride)
                       : (c->wsplit_off[oi] >= 0 ? mbxb_tiles(ob.H, ob.W, o.k, o.stride) : mbx_tiles(ob.H, ob.W, o.k, o.stride));
      if (o.se_partial >= 0) {
        const uda_buf_desc_t& pb = c->bufs[o.se_partial];
        if ((int64_t)pb.H * pb.W != (int64_t)a.n_tiles || pb.C != ob.C || pb.per_sample != ob.per_sample)
          return fail(c, "op %d: SE partial buffer [%d,%d,%d] does not match %d tiles", oi, pb.H, pb.W, pb.C, a.n_tiles);
        a.se_partial = v.ptr(o.se_partial);
      }
      if (c->wsplit_off[oi] >= 0) {
        a.wsplit = wsplit_of(c, oi);
        a.wparts = c->wscheme[oi];
        a.oor = c->oor_cur + oi;
        if (a.wparts == UDA_SPLIT_F16X2) c->oor_armed = true;
        a.wpar = (const float*)wpar_of(c, oi);
        if (fuse0) {
          const uda_buf_desc_t& gb = c->bufs[o.se_scale];
          if (gb.per_sample && !ob.per_sample) return fail(c, "op %d: per-sample gate on a per-image output", oi);
          a.gate = v.ptr(o.se_scale);
          a.g_div = v.div(gb, ob);
          a.c0 = ib.C;
          a.w0t = a.wpar + mbx_par_floats(ob.C, o.k);
          a.sh0f = a.w0t + 32 * 32;
          static const bool pre = !(getenv("UDA_W0GATE") && atoi(getenv("UDA_W0GATE")) == 0);   // 0: every block redoes the prep
          if (pre && ib.C == 32) {
            // one buffer per chunk lane: with UDA_LANES=2 consecutive chunks run concurrently on two streams, each with its
            // own gates - lane 1's prep must not overwrite the fragments lane 0's block-1 kernel is still reading
            const int gate_rows = v.rows(gb);
            const int ln = v.lane & 1;
            const size_t need = mbxb_w0frag_elems(gate_rows, a.wparts);
            if (need > c->w0frag_cap[ln]) {
              HIPC(c, hipStreamSynchronize(v.stream()));     // this lane's stream is the only user of this lane's buffer
              if (c->d_w0frag[ln]) HIPC(c, hipFree(c->d_w0frag[ln]));
              c->d_w0frag[ln] = nullptr; c->w0frag_cap[ln] = 0;
              HIPC(c, hipMalloc((void**)&c->d_w0frag[ln], need * sizeof(uint4)));
              c->w0frag_cap[ln] = need;
            }
            launch_w0gate(a.gate, a.w0t, a.c0, gate_rows, a.wparts, c->d_w0frag[ln], c->oor_cur + oi, v.stream());
            a.w0frag = c->d_w0frag[ln];
          }
        }
        if (deep) launch_mbxd(a, rows, o.k, o.stride, v.stream());
        else launch_mbxb(a, rows, o.k, o.stride, v.stream());
      } else {
        launch_mbx(a, rows, o.k, o.stride, v.stream());
      }
      break;
    }
    case UDA_OP_SEP: {
      const uda_buf_desc_t& ib = c->bufs[o.in[0]];
      if (c->wsplit_off[oi] < 0) return fail(c, "op %d: fused separable conv needs the split-bf16 path (UDA_PW_TERMS != 0)", oi);
      SepArgs a{};
      a.in = v.ptr(o.in[0]);
      a.out = v.ptr(o.out);
      a.wd = has_wpar(c, oi) ? (const float*)wpar_of(c, oi) : v.wt(o.w2_off);    // (fp16 pieces: pre-scaled taps)
      a.wsplit = wsplit_of(c, oi);
      a.wparts = c->wscheme[oi];
      a.wunscale = c->wunscale[oi];
      a.oor = c->oor_cur + oi;
      if (a.wparts == UDA_SPLIT_F16X2) c->oor_armed = true;
      a.bias = v.wt(o.bias_off);
      a.bn_scale = v.wt(o.bn_scale_off);
      a.bn_shift = v.wt(o.bn_shift_off);
      a.mask = v.mask(o.drop_site);
      a.mask_in = (!o.fuse_in && o.drop_site2 >= 0) ? v.mask(o.drop_site2) : nullptr;
      a.H = ob.H; a.W = ob.W; a.C = ib.C; a.Cout = ob.C;
      a.in_div = v.div(ib, ob);
      a.act = o.act;
      if (a.mask_in && !sep_tin_supported(a.C, a.Cout, a.wparts))
        return fail(c, "op %d: separable conv %d -> %d has no deferred-input mode (planner: plan.sep_tin_supported)", oi, a.C, a.Cout);
      if (o.fuse_in) {
        // the node's BiFPN fusion is this conv's input, computed on the fly (act_type, or none under conv_bn_act_pattern:
        // efficientdet_keras.py:229-236)
        FuseArgs f{};
        if (int rc = fill_fuse_args(c, v, oi, f)) return rc;
        f.act = o.fuse_act;
        if (!sepf_supported(a.C, a.Cout, a.wparts)) return fail(c, "op %d: fused-input separable conv %d -> %d not supported", oi, a.C, a.Cout);
        a.in = nullptr;
        launch_sepf(a, &f, rows, v.stream());
        break;
      }
      if (ib.H != ob.H || ib.W != ob.W) return fail(c, "op %d: separable conv changes the map size", oi);
      static const int sepf_all = getenv("UDA_SEPF_ALL") ? atoi(getenv("UDA_SEPF_ALL")) : 0;     // A/B: the tile kernel for every conv
      if (sepf_all && sepf_supported(a.C, a.Cout, a.wparts)) launch_sepf(a, nullptr, rows, v.stream());
      else launch_sep(a, rows, v.stream());
      break;
    }
    case UDA_OP_SE: {
      const uda_buf_desc_t& pb = c->bufs[o.in[0]];   // partial sums written by the DW op
      const uda_buf_desc_t& src = c->bufs[o.in[1]];  // the DW output (for H*W and geometry)
      SeArgs a{};
      a.partial = v.ptr(o.in[0]);
      a.scale = v.ptr(o.out);
      a.w1 = v.wt(o.se_w1_off); a.b1 = v.wt(o.se_b1_off);
      a.w2 = v.wt(o.se_w2_off); a.b2 = v.wt(o.se_b2_off);
      a.C = src.C; a.mid = o.se_mid; a.n_tiles = pb.H * pb.W;   // the producer (DW / MBX) validated this count
      a.inv_hw = 1.0f / (float)(src.H * src.W);
      if (pb.C != src.C || ob.C != src.C) return fail(c, "op %d: SE channel mismatch", oi);
      if (a.mid < 1 || a.mid > 1024) return fail(c, "op %d: SE hidden width %d outside [1, 1024]", oi, a.mid);
      // deferred dropout site (plan.py): the squeezed tensor is per image, its keep-scale per sample row
      a.mask = v.mask(o.drop_site);
      a.in_div = v.div(pb, ob);
      a.act = o.act;
      if (o.drop_site >= 0 && !ob.per_sample && c->model.mc_samples > 1) return fail(c, "op %d: deferred dropout needs a per-sample gate", oi);
      if (pb.per_sample != src.per_sample) return fail(c, "op %d: SE sums and source disagree on the sample axis", oi);
      launch_se(a, rows, v.stream());
      break;
    }
    case UDA_OP_FUSE:
    case UDA_OP_POOL: {
      FuseArgs a{};
      if (int rc = fill_fuse_args(c, v, oi, a)) return rc;
      if (a.C != ob.C) return fail(c, "op %d: fuse inputs have %d channels, output %d", oi, a.C, ob.C);
      a.out = v.ptr(o.out);
      a.act = o.act;
      a.total = (int64_t)rows * ob.H * ob.W * (ob.C / 4);
      launch_fuse(a, v.stream());
      break;
    }
    default:
      return fail(c, "op %d: unknown kind %d", oi, o.kind);
  }
  return 0;
}

// ops[oi .. oi + n): the separable convs of one head layer on all pyramid levels (uda_op_t.launch_group) as ONE launch.
// Returns -1 when the run does not qualify (the caller then executes the ops one by one), 0 on success, > 0 on failure.
static int run_sep_group(uda_ctx* c, const ChunkView& v, int oi, int n) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("UDA_SEP_MULTI"); on = e ? atoi(e) : 1; }
  if (!on || n < 2 || n > UDA_SEP_MAX_LV || oi + n > (int)c->ops.size()) return -1;
  const uda_op_t& o0 = c->ops[oi];
  const uda_buf_desc_t& ib0 = c->bufs[o0.in[0]];
  const uda_buf_desc_t& ob0 = c->bufs[o0.out];
  SepLevel lv[UDA_SEP_MAX_LV];
  for (int j = 0; j < n; ++j) {
    const uda_op_t& o = c->ops[oi + j];
    if (o.kind != UDA_OP_SEP || c->wsplit_off[oi + j] < 0) return -1;
    const uda_buf_desc_t& ib = c->bufs[o.in[0]];
    const uda_buf_desc_t& ob = c->bufs[o.out];
    if (ib.C != ib0.C || ob.C != ob0.C || o.act != o0.act || ib.per_sample != ib0.per_sample || ob.per_sample != ob0.per_sample ||
        ib.H != ob.H || ib.W != ob.W)
      return -1;
    for (int k = 0; k < n; ++k)
      if (k != j && (c->ops[oi + k].out == o.in[0] || c->ops[oi + k].out == o.out)) return -1;    // not independent
    lv[j].in = v.ptr(o.in[0]);
    lv[j].out = v.ptr(o.out);
    lv[j].wd = has_wpar(c, oi + j) ? (const float*)wpar_of(c, oi + j) : v.wt(o.w2_off);
    lv[j].wsplit = wsplit_of(c, oi + j);
    lv[j].bias = v.wt(o.bias_off);
    lv[j].bn_scale = v.wt(o.bn_scale_off);
    lv[j].bn_shift = v.wt(o.bn_shift_off);
    lv[j].mask = v.mask(o.drop_site);
    lv[j].mask_in = o.drop_site2 >= 0 ? v.mask(o.drop_site2) : nullptr;
    if ((o.drop_site2 >= 0) != (o0.drop_site2 >= 0)) return -1;
    lv[j].H = ob.H; lv[j].W = ob.W;
    lv[j].wunscale = c->wunscale[oi + j];
    if (c->wscheme[oi + j] != c->wscheme[oi]) return -1;
  }
  // The grouped ops run concurrently inside one grid: no op's OUTPUT byte range may touch another op's input or output
  // (the planner keeps every buffer of such a run alive to its end; this is the executor's own check of that promise).
  for (int j = 0; j < n; ++j) {
    const uda_buf_desc_t& obj = c->bufs[c->ops[oi + j].out];
    const char* o0p = (const char*)lv[j].out;
    const char* o1p = o0p + (size_t)v.rows(obj) * obj.H * obj.W * obj.C * sizeof(float);
    for (int k = 0; k < n; ++k) {
      if (k == j) continue;
      const uda_buf_desc_t& ibk = c->bufs[c->ops[oi + k].in[0]];
      const uda_buf_desc_t& obk = c->bufs[c->ops[oi + k].out];
      const char* i0p = (const char*)lv[k].in;
      const char* i1p = i0p + (size_t)v.rows(ibk) * ibk.H * ibk.W * ibk.C * sizeof(float);
      const char* q0p = (const char*)lv[k].out;
      const char* q1p = q0p + (size_t)v.rows(obk) * obk.H * obk.W * obk.C * sizeof(float);
      if ((o0p < i1p && i0p < o1p) || (o0p < q1p && q0p < o1p))
        return fail(c, "ops %d and %d of a launch group overlap in memory (arena plan broken)", oi + j, oi + k);
    }
  }
  ProfScope ps(c, UDA_OP_SEP, v.stream(), n);      // one launch for n planned ops
  SepArgs a{};
  a.C = ib0.C; a.Cout = ob0.C;
  a.in_div = v.div(ib0, ob0);
  a.act = o0.act;
  a.mask_in = o0.drop_site2 >= 0 ? v.mask(o0.drop_site2) : nullptr;     // (non-null = the deferred-input mode; the levels carry their own)
  if (a.mask_in && !sep_tin_supported(a.C, a.Cout, c->wscheme[oi]))
    return fail(c, "ops %d..: separable conv %d -> %d has no deferred-input mode (planner: plan.sep_tin_supported)", oi, a.C, a.Cout);
  a.wparts = c->wscheme[oi];
  a.wunscale = c->wunscale[oi];
  a.oor = c->oor_cur + oi;            // (one word for the layer's launch: its levels share the 1x1 kernel and are re-packed together)
  if (a.wparts == UDA_SPLIT_F16X2) c->oor_armed = true;
  launch_sep_multi(a, lv, n, v.rows(ob0), v.stream());
  return 0;
}

static int run_post_range(uda_ctx* c, int i0, int n, int post_mode, hipStream_t st);

static int run_preprocess(uda_ctx* c) {
  const uda_model_t& m = c->model;
  ProfScope ps(c, 18);
  PreprocArgs a{};
  const uda_ctx::U8Slot& sl = c->u8[c->cur];
  a.in = sl.d_img; a.out = c->d_images; a.geo = sl.d_geo;
  a.n = c->n_images; a.H = m.image_h; a.W = m.image_w;
  for (int k = 0; k < 3; ++k) { a.mean[k] = m.mean_rgb[k]; a.stdv[k] = m.stddev_rgb[k]; }
  launch_preprocess(a, c->stream);
  HIPC(c, hipGetLastError());
  c->pre_valid = true;
  return 0;
}

static int run_network(uda_ctx* c, int post_mode = 0, bool chunk_post = false, hipEvent_t defer_ev = nullptr) {
  const uda_model_t& m = c->model;
  const int n = c->n_images, T = m.mc_samples;
  // uint8 batch in which no image is resampled (scale 1: raw size within the network size - BASELINE configs[1]-[3]): the stem
  // reads the raw images itself (StemArgs::u8), the preprocess pass and its float32 image (378 MB written and read back per
  // 32-image batch) are skipped; UDA_STEM_U8=0 restores the separate pass
  bool stem_u8 = false;
  if (c->have_u8) {
    static const bool on = !(getenv("UDA_STEM_U8") && atoi(getenv("UDA_STEM_U8")) == 0);
    const uda_ctx::U8Slot& sl = c->u8[c->cur];
    stem_u8 = on && c->stem_co > 0 && stem_u8_supported(c->stem_co) && c->stem_act == UDA_ACT_SWISH;
    for (int i = 0; i < n && stem_u8; ++i) stem_u8 = sl.geo[i].sh == sl.geo[i].h && sl.geo[i].sw == sl.geo[i].w;
    c->pre_valid = false;
    if (!stem_u8) {
      if (int rc = run_preprocess(c)) return rc;
      HIPC(c, hipEventRecord(c->ev_pre_done[c->cur], c->stream));
    }
  }
  c->stem_from_u8 = stem_u8;
  if (!c->sites.empty()) {
    const int rows = n * T;
    if (c->masks_injected && c->masks_rows != rows)
      return fail(c, "injected dropout masks cover %d sample rows, the run has %d", c->masks_rows, rows);
    int64_t off = 0;
    for (size_t s = 0; s < c->sites.size(); ++s) {
      c->site_off[s] = off;
      off += (int64_t)rows * c->sites[s].channels;
    }
    if (!c->masks_injected) {
      HIPC(c, hipMemcpyAsync(c->d_site_off, c->site_off.data(), c->sites.size() * sizeof(int64_t),
                             hipMemcpyHostToDevice, c->stream));
      launch_philox_masks(c->d_masks, c->d_site_off, c->d_site_ch, c->d_site_rate, (int)c->sites.size(), rows,
                          (uint32_t)(c->image_offset * (c->t_total > 0 ? c->t_total : T)), c->max_c4, c->seed,
                          T, c->t_total > 0 ? c->t_total : T, c->t_first, c->t_stride, c->stream);
    }
  }
  const int lanes = c->n_lanes;
  if (lanes > 1) {
    HIPC(c, hipEventRecord(c->ev_start, c->stream));
    for (int l = 1; l < lanes; ++l) HIPC(c, hipStreamWaitEvent(c->lane_stream[l], c->ev_start, 0));
  }
  int ci = 0;
  for (int i0 = 0; i0 < n; i0 += m.chunk_images, ++ci) {
    ChunkView v{c, i0, (n - i0 < m.chunk_images) ? n - i0 : m.chunk_images};
    v.lane = ci % lanes;
    bool gated = false;
    for (int oi = 0; oi < (int)c->ops.size(); ++oi) {
      // pipelined runs: the previous run's post-process still reads the head buffers - the first op of this chunk that writes
      // one waits for it (everything before it, i.e. the whole backbone and BiFPN, runs beside that post-process)
      if (c->gate_ev && !gated && c->bufs[c->ops[oi].out].kind >= 2) {
        HIPC(c, hipStreamWaitEvent(v.stream(), c->gate_ev, 0));
        gated = true;
      }
      const int grp = c->ops[oi].launch_group;
      static const bool sepf_all_ = getenv("UDA_SEPF_ALL") && atoi(getenv("UDA_SEPF_ALL"));     // A/B: per-level tile-kernel launches
      if (grp > 1 && !sepf_all_) {
        const int rg = run_sep_group(c, v, oi, grp);
        if (rg > 0) return rg;
        if (rg == 0) {
          const hipError_t le = hipGetLastError();
          if (le != hipSuccess)
            return fail(c, "ops %d..%d (head layer, one launch for %d pyramid levels): launch failed: %s", oi, oi + grp - 1, grp, hipGetErrorString(le));
          oi += grp - 1;
          continue;
        }
      }
      const int rc = run_op(c, v, oi);
      if (rc) return rc;
      // debugging aid (UDA_SYNC_EACH=1): wait for every op and name it on stderr before the next one is queued, so that a
      // device fault is attributed to the op that raised it
      static const bool sync_each = getenv("UDA_SYNC_EACH") != nullptr;
      if (sync_each) {
        const uda_op_t& o = c->ops[oi];
        fprintf(stderr, "[uda] op %d kind %d k %d s %d C %d -> %d ... ", oi, o.kind, o.k, o.stride, c->bufs[o.in[0]].C, c->bufs[o.out].C);
        fflush(stderr);
        const hipError_t se = hipStreamSynchronize(v.stream());
        fprintf(stderr, "%s\n", hipGetErrorString(se));
      }
      // a refused launch (LDS over budget, bad grid) is named with its op: the query is host-only and costs nothing
      const hipError_t le = hipGetLastError();
      if (le != hipSuccess) {
        const uda_op_t& o = c->ops[oi];
        return fail(c, "op %d (kind %d, k %d, stride %d, in C %d, out C %d): launch failed: %s", oi, o.kind, o.k, o.stride,
                    c->bufs[o.in[0]].C, c->bufs[o.out].C, hipGetErrorString(le));
      }
    }
    c->last_chunk_i0 = i0;
    c->last_chunk_n = v.nc;
    c->last_lane = v.lane;
    if (chunk_post) {     // this chunk's head outputs are complete once its lane stream reaches the event
      HIPC(c, hipEventRecord(c->ev_chunk[ci], v.stream()));
      HIPC(c, hipStreamWaitEvent(c->post_stream, c->ev_chunk[ci], 0));
      const int rc = run_post_range(c, i0, v.nc, post_mode, c->post_stream);
      if (rc) return rc;
    }
  }
  if (chunk_post) {
    if (defer_ev) {
      HIPC(c, hipEventRecord(defer_ev, c->post_stream));      // pipelined: the main stream goes on without the post-process
    } else {
      HIPC(c, hipEventRecord(c->ev_post, c->post_stream));
      HIPC(c, hipStreamWaitEvent(c->stream, c->ev_post, 0));
    }
    c->last_post_mode = post_mode;
    c->last_n = n;
  }
  for (int l = 1; l < lanes; ++l) {       // the post-process on the main stream needs every lane's head outputs
    HIPC(c, hipEventRecord(c->ev_done[l], c->lane_stream[l]));
    HIPC(c, hipStreamWaitEvent(c->stream, c->ev_done[l], 0));
  }
  if (stem_u8) HIPC(c, hipEventRecord(c->ev_pre_done[c->cur], c->stream));      // the stem ops have consumed the uint8 slot
  HIPC(c, hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------ post-process
static NmsArgs nms_args_at(uda_ctx::NmsWs& w, size_t p0, int problems, int K, int M, const float* boxes);

// Returns true when the problems were solved on their score prefix (flags in pw->bad[p0 ..] say which ones have to be
// redone on the full set, see finish_post); `pw` null = never.
struct NmsCoop {           // scratch of the cooperative kernel; null members = never use it
  unsigned long long* bar = nullptr;    // exchange slots (per problem nms_coop_slot_words(M) words)
  int* err = nullptr;
  bool* used = nullptr;
  int64_t* not_launched = nullptr;      // counts the runs that wanted the single launch and fell through to the slower versions
};

static bool run_nms(const NmsArgs& na, const float* scores, int M, hipStream_t st, uda_ctx::PrefixWs* pw = nullptr, size_t p0 = 0,
                    NmsCoop coop = NmsCoop()) {
  // One launch for all epochs (one block per problem) when a problem is small - the top-k / per-class paths with a
  // few thousand candidates each; with the whole anchor set as candidates (184 k near-tied scores under random-init
  // weights) an epoch revisits 10-20 chunks one after the other inside the block and the grid version, which scans
  // all chunks in parallel, is 4x faster (measured: 35 vs 7.9 ms for 32 images).  UDA_NMS_SOLO = candidates per
  // problem up to which the single-launch kernel is used (0 = never).
  const int solo = solo_limit();
  static int reg = -1;
  if (reg < 0) { const char* e = getenv("UDA_NMS_REG"); reg = e ? atoi(e) : 1; }
  if (na.K <= solo && reg && nms_reg_supported(na)) {
    launch_nms_reg(na, scores, st);
    return false;
  }
  if (na.K <= solo && nms_solo_supported(na)) {
    launch_nms_solo(na, scores, st);
    return false;
  }
  // The whole set: all epochs in one launch of a co-resident grid when the device holds it (UDA_NMS_COOP=0: never) ...
  static int coop_on = -1;
  if (coop_on < 0) { const char* e = getenv("UDA_NMS_COOP"); coop_on = e ? atoi(e) : 1; }
  if (coop_on && coop.bar && coop.err) {
    const int lc = launch_nms_coop(na, scores, coop.bar + p0 * nms_coop_slot_words(M), coop.err, st);
    if (lc > 0) {
      if (coop.used) *coop.used = true;
      return false;
    }
    if (lc < 0 && coop.not_launched) ++*coop.not_launched;
  }
  // ... else (more than 64 x 32768 candidates per problem, or a grid the device cannot hold): the single-launch kernel on
  // the candidates that can be popped at all (score prefix), checked on the device, full set - two launches per epoch -
  // only for the problems the check rejects.
  if (pw && pw->Lcap > 0 && na.segs == 1 && na.K > pw->Lcap && M <= 128) {
    const size_t lc = (size_t)pw->Lcap;
    PrefixArgs pa{};
    pa.scores = scores; pa.boxes = na.boxes;
    pa.sub_idx = pw->sub_idx + p0 * lc; pa.sub_scores = pw->sub_scores + p0 * lc; pa.sub_boxes = pw->sub_boxes + p0 * lc * 4;
    pa.excl_key = pw->excl + p0; pa.bad = pw->bad + p0;
    pa.n_img = na.n_img; pa.K = na.K; pa.Lp = pw->Lcap / 2; pa.Lcap = pw->Lcap;
    launch_prefix_select(pa, st);
    NmsArgs sub = nms_args_at(pw->ws, p0, na.n_img, pw->Lcap, M, pa.sub_boxes);
    sub.iou_thr = na.iou_thr; sub.score_thr = na.score_thr; sub.soft = na.soft; sub.scale = na.scale;
    if (reg && nms_reg_supported(sub)) launch_nms_reg(sub, pa.sub_scores, st);
    else launch_nms_solo(sub, pa.sub_scores, st);
    PrefixCheckArgs ca{};
    ca.sub_sel_idx = sub.sel_idx; ca.sub_sel_score = sub.sel_score; ca.sub_nsel = sub.nsel;
    ca.sub_idx = pa.sub_idx; ca.excl_key = pa.excl_key;
    ca.sel_idx = na.sel_idx; ca.sel_score = na.sel_score; ca.nsel = na.nsel; ca.bad = pa.bad;
    ca.n_img = na.n_img; ca.M = M; ca.Lcap = pw->Lcap; ca.score_thr = na.score_thr;
    launch_prefix_check(ca, st);
    return true;
  }
  launch_nms_init(na, scores, st);
  for (int e = 0; e < M; ++e) launch_nms_epoch(na, e, st);
  return false;
}

static void nms_params(NmsArgs& a, float iou_thr, float score_thr, float soft_sigma) {
  a.iou_thr = iou_thr;
  a.score_thr = score_thr;
  a.soft = soft_sigma > 0.0f;
  a.scale = a.soft ? -0.5f / soft_sigma : 0.0f;
}

// a8-a14: class statistics, (top-k pre-selection), per-sample decode and MC aggregation -> candidates
// All post-process stages work on the image range [i0, i0 + n) and on stream `st`.
static int run_candidates(uda_ctx* c, int i0, int n, hipStream_t st) {
  const uda_model_t& m = c->model;
  ProfScope ps(c, 16, st);
  AggArgs a{};
  a.lv.num_levels = m.num_levels;
  const int Tc = m.cls_stacked ? m.mc_samples : 1, Tb = m.box_stacked ? m.mc_samples : 1;
  const size_t K = (size_t)c->Kc, C = (size_t)m.num_classes;
  const size_t uc = m.max_nms_inputs > 0 ? 1 : C;     // class-std values per candidate
  for (int l = 0; l < m.num_levels; ++l) {
    const size_t hw = (size_t)m.level_h[l] * m.level_w[l];
    a.lv.hw[l] = (int)hw;
    a.lv.a_off[l] = c->a_off[l];
    a.lv.cls[l] = c->d_cls[l] + (size_t)i0 * Tc * hw * c->cls_ch;
    a.lv.box[l] = c->d_box[l] + (size_t)i0 * Tb * hw * c->box_ch;
  }
  a.lv.a_off[m.num_levels] = c->A_tot;
  a.anchors = c->d_anchors;
  a.n_img = n; a.A_tot = c->A_tot; a.A = m.anchors_per_loc; a.C = m.num_classes;
  a.K = c->Kc;
  a.Tc = Tc;
  a.Tb = Tb;
  a.loss_att = m.loss_attenuation;
  a.decode = m.decode_method;
  a.decode_nsamples = m.decode_nsamples;
  a.decode_seed = c->seed ^ 0x5DEC0DE5A3B1E5ull;
  a.row_base = (uint32_t)((c->image_offset + i0) * Tb);
  a.boxes = c->d_cboxes + (size_t)i0 * K * 4; a.scores = c->d_cscores + (size_t)i0 * K;
  a.classes = c->d_cclasses + (size_t)i0 * K; a.logits = c->d_clogits + (size_t)i0 * K * C;
  a.u_cls = c->d_ucls ? c->d_ucls + (size_t)i0 * K * uc : nullptr;
  a.u_al = c->d_ual ? c->d_ual + (size_t)i0 * K * 4 : nullptr;
  a.u_ep = c->d_uep ? c->d_uep + (size_t)i0 * K * 4 : nullptr;
  a.cand_flat = nullptr;
  if (m.max_nms_inputs > 0) {
    float* cm = c->d_clsmean + (size_t)i0 * c->A_tot * C;
    int32_t* cf = c->d_cand_flat + (size_t)i0 * K;
    launch_class_mean(a, cm, st);
    launch_topk(cm, n, c->A_tot * m.num_classes, c->Kc, cf, c->d_topk_ws, st);
    a.cand_flat = cf;
  }
  launch_aggregate(a, st);
  return 0;
}

static NmsArgs nms_args_at(uda_ctx::NmsWs& w, size_t p0, int problems, int K, int M, const float* boxes) {
  NmsArgs a{};
  a.boxes = boxes;
  const size_t k = (size_t)K, mm = (size_t)M;
  a.stale = w.stale + p0 * k; a.begin = w.begin + p0 * k; a.tent = w.tent + p0 * k; a.ub = w.ub + p0 * k; a.ev = w.ev + p0 * k;
  a.sel_idx = w.sel_idx + p0 * mm; a.sel_score = w.sel_score + p0 * mm; a.sel_box = w.sel_box + p0 * mm * 4;
  a.bound_key = w.bound + p0 * mm; a.win_key = w.win + p0 * mm; a.nsel = w.nsel + p0; a.done = w.done + p0;
  a.n_img = problems; a.K = K; a.M = M;
  a.segs = 1; a.classes = nullptr;
  return a;
}

static int run_post_global(uda_ctx* c, int i0, int n, hipStream_t st) {
  const uda_model_t& m = c->model;
  int rc = c->pfx_off ? 0 : run_candidates(c, i0, n, st);    // a redo (finish_post) finds the candidates in place
  if (rc) return rc;
  const int K = c->Kc, M = m.max_output_size;
  const size_t k = (size_t)K, mm = (size_t)M, C = (size_t)m.num_classes;
  const size_t uc = m.max_nms_inputs > 0 ? 1 : C;
  {
    ProfScope ps(c, 17, st);
    NmsArgs na = nms_args_at(c->ws[0], (size_t)i0, n, K, M, c->d_cboxes + (size_t)i0 * k * 4);
    nms_params(na, m.nms_iou_thresh, m.nms_score_thresh, m.nms_soft_sigma);
    const bool try_prefix = !c->pfx_off && c->pfx_skip == 0;
    NmsCoop coop;
    if (!c->coop_off) { coop.bar = c->d_coop_bar; coop.err = c->d_coop_err; coop.used = &c->coop_used; coop.not_launched = &c->coop_not_launched; }
    if (run_nms(na, c->d_cscores + (size_t)i0 * k, M, st, try_prefix ? &c->pfx : nullptr, (size_t)i0, coop))
      c->pfx_pending.push_back({i0, n});
  }
  GatherArgs g{};
  g.box_cols = box_cols_of(m, UDA_POST_GLOBAL);
  g.cls_cols = cls_cols_of(m, UDA_POST_GLOBAL);
  g.ucls_cols = g.cls_cols - 1;
  g.sel_idx = c->ws[0].sel_idx + i0 * mm; g.sel_score = c->ws[0].sel_score + i0 * mm; g.nsel = c->ws[0].nsel + i0;
  g.boxes = c->d_cboxes + i0 * k * 4; g.classes = c->d_cclasses + i0 * k; g.logits = c->d_clogits + i0 * k * C;
  g.u_cls = c->d_ucls ? c->d_ucls + i0 * k * uc : nullptr;
  g.u_al = c->d_ual ? c->d_ual + i0 * k * 4 : nullptr;
  g.u_ep = c->d_uep ? c->d_uep + i0 * k * 4 : nullptr;
  g.scales = (c->d_scales_post ? c->d_scales_post : c->d_scales) + i0;
  g.out_boxes = c->d_oboxes + i0 * mm * g.box_cols; g.out_scores = c->d_oscores + i0 * mm;
  g.out_classes = c->d_oclasses + i0 * mm * g.cls_cols;
  g.out_valid = c->d_ovalid + i0; g.out_logits = c->d_ologits + i0 * mm * C;
  g.n_img = n; g.K = K; g.M = M; g.C = m.num_classes;
  g.clip_h = (float)m.image_h; g.clip_w = (float)m.image_w; g.clip = 1;
  launch_gather(g, st);
  HIPC(c, hipGetLastError());
  return 0;
}

// a17: one NMS problem per (image, class), then concat / pad / top-M (postprocess.py:624-740)
static int run_post_per_class(uda_ctx* c, int i0, int n, hipStream_t st) {
  const uda_model_t& m = c->model;
  int rc = run_candidates(c, i0, n, st);
  if (rc) return rc;
  const int K = c->Kc, M = m.max_output_size, C = m.num_classes;
  const size_t k = (size_t)K, mm = (size_t)M;
  const size_t p0 = (size_t)i0 * C;
  {
    ProfScope ps(c, 17, st);
    NmsArgs na = nms_args_at(c->ws[1], p0, n * C, K, M, c->d_cboxes + (size_t)i0 * k * 4);
    na.segs = C;
    na.classes = c->d_cclasses + (size_t)i0 * k;
    nms_params(na, m.nms_iou_thresh, m.nms_score_thresh, m.nms_soft_sigma);
    run_nms(na, c->d_cscores + (size_t)i0 * k, M, st);
  }
  MergeArgs g{};
  g.sel_idx = c->ws[1].sel_idx + p0 * mm; g.sel_score = c->ws[1].sel_score + p0 * mm; g.nsel = c->ws[1].nsel + p0;
  g.boxes = c->d_cboxes + (size_t)i0 * k * 4; g.scales = (c->d_scales_post ? c->d_scales_post : c->d_scales) + i0;
  g.keys = c->d_merge_keys + (size_t)i0 * ((size_t)C * M + M);
  g.out_boxes = c->d_oboxes + i0 * mm * 4; g.out_scores = c->d_oscores + i0 * mm;
  g.out_classes = c->d_oclasses + i0 * mm; g.out_valid = c->d_ovalid + i0;
  g.n_img = n; g.K = K; g.M = M; g.C = C;
  launch_merge_per_class(g, st);
  HIPC(c, hipGetLastError());
  return 0;
}

static int run_post_range(uda_ctx* c, int i0, int n, int pm, hipStream_t st) {
  if (pm == UDA_POST_GLOBAL) return run_post_global(c, i0, n, st);
  if (pm == UDA_POST_PER_CLASS) return run_post_per_class(c, i0, n, st);
  return fail(c, "unknown post mode %d", pm);
}

static int prepare_post(uda_ctx* c, int pm) {
  if (pm != UDA_POST_PER_CLASS) return 0;
  const uda_model_t& m = c->model;
  const size_t N = (size_t)m.max_images, C = (size_t)m.num_classes, M = (size_t)m.max_output_size;
  HIPC(c, alloc_nms_ws(c->ws[1], N * C, (size_t)c->Kc, M));
  if (!c->d_merge_keys) HIPC(c, dalloc(&c->d_merge_keys, N * (C * M + M)));
  return 0;
}

static int run_post(uda_ctx* c, int n, int post_mode) {
  const int pm = post_mode < 0 ? c->model.post_mode : post_mode;
  int rc = prepare_post(c, pm);
  if (rc) return rc;
  c->pfx_pending.clear();
  rc = run_post_range(c, 0, n, pm, c->stream);
  if (c->pfx_skip > 0 && c->pfx_pending.empty()) --c->pfx_skip;
  if (rc) return rc;
  c->last_post_mode = pm;
  c->last_n = n;
  return 0;
}

// ------------------------------------------------------------------------------------ fp16 range: demote and serve again
// fp16-piece contractions (UDA_SPLIT_F16X2): an activation above 65504 cannot be split and its products are infinite.
// Every kernel that splits operands reports that through its op's flag word; every reader of a run's results comes
// through check_split_range first.  A raised flag does not fail the run (the reference computes in float32 and always
// returns, infer_lib.py:337-343): the FIRST flagged op in op order - everything behind it only saw its infinities - is
// re-packed with three bf16 pieces (float32 exponent range; its kernels exist per op: `wscheme`), the run is served
// again from its unchanged inputs on the same handle, and the flags are read again, until the run is clean.  The op
// stays demoted for the life of the handle (`uda_range_demotions` counts them; one line on stderr each).
static int current_run_rec(const uda_ctx* c, uda_ctx::RunRec* r, int pm, bool do_post) {
  r->valid = true; r->pm = pm; r->do_post = do_post;
  r->have_u8 = c->have_u8; r->cur = c->cur; r->n = c->n_images;
  r->slot_gen = c->u8[c->cur].gen; r->f32_gen = c->f32_gen;
  r->masks_injected = c->masks_injected; r->masks_gen = c->masks_gen;
  r->seed = c->seed; r->image_offset = c->image_offset;
  return 0;
}

static bool can_replay(const uda_ctx* c, const uda_ctx::RunRec* r) {
  if (!r || !r->valid) return false;
  if (r->have_u8 ? c->u8[r->cur].gen != r->slot_gen : c->f32_gen != r->f32_gen) return false;
  if (r->masks_injected && c->masks_gen != r->masks_gen) return false;
  return true;
}

// Re-pack op `oi` (and the ops that share its launch: the pyramid levels of a head layer) with three bf16 pieces.
static int demote_ops(uda_ctx* c, int oi) {
  const int n_ops = (int)c->ops.size();
  if (oi < 0 || oi >= n_ops) return fail(c, "fp16 range flag outside the op list (word %d): re-create the handle with UDA_PW_SCHEME=bf16x3", oi);
  int g0 = oi, g1 = oi + 1;
  for (int i = 0; i < n_ops; ++i) {
    const int lg = c->ops[i].launch_group;
    if (lg > 1 && i <= oi && oi < i + lg) { g0 = i; g1 = i + lg; break; }
  }
  const float* weights = c->h_weights.data();
  int done = 0;      // (0: re-packed already - a pipelined run queued before that demotion raised the same flag)
  for (int i = g0; i < g1 && i < n_ops; ++i) {
    const uda_op_t& o = c->ops[i];
    if (c->wscheme[i] != UDA_SPLIT_F16X2 || c->wsplit_off[i] < 0) continue;
    const int K = c->bufs[o.in[0]].C, Nn = c->bufs[o.out].C;
    const int sch = UDA_SPLIT_BF16X3;
    std::vector<uint16_t> packed;
    int64_t par = -1;
    size_t lds = 0;
    if (o.kind == UDA_OP_PW || o.kind == UDA_OP_SEP) {
      packed.resize((pwb_packed_elems(K, Nn, sch) + 7) / 8 * 8);
      pwb_pack_weights(weights + o.w_off, K, Nn, sch, packed.data(), 1.0f);
      if (o.kind == UDA_OP_SEP) {
        lds = o.fuse_in ? sepf_lds_bytes(K, Nn, sch) : sep_lds_bytes(K, Nn, sch);
        if (o.fuse_in && !sepf_supported(K, Nn, sch))
          return fail(c, "op %d raised the fp16 range flag and has no three-piece kernel (fused-input separable conv %d -> %d): "
                         "re-create the handle with UDA_PW_SCHEME=bf16x3", i, K, Nn);
      }
    } else if (o.kind == UDA_OP_MBX) {
      const bool fuse0 = o.se_scale >= 0;
      const int Ke = fuse0 ? o.se_mid : K;
      const size_t we = (mbxb_packed_elems(Ke, Nn, sch) + 7) / 8 * 8;
      const size_t par_fl = mbx_par_floats(Nn, o.k), proj_fl = fuse0 ? 32 * 32 + 32 : 0;
      packed.resize(we + 2 * (par_fl + proj_fl));
      mbxb_pack_weights(weights + o.w_off, weights + o.bn_scale_off, weights + o.bn_shift_off, Ke, Nn, packed.data(), fuse0, sch);
      std::vector<float> pf(par_fl + proj_fl);
      mbx_pack_params(weights + o.w2_off, weights + o.bn2_scale_off, weights + o.bn2_shift_off, Nn, o.k, pf.data());
      if (fuse0) mbxb_pack_proj(weights + o.se_w1_off, weights + o.se_b1_off, weights + o.se_w2_off, K, Ke, pf.data() + par_fl);
      memcpy(packed.data() + we, pf.data(), pf.size() * sizeof(float));
      par = (int64_t)we;
      lds = mbx_lds_bytes(Ke, Nn, o.k, o.stride, sch, c->bufs[o.out].H, c->bufs[o.out].W);
    } else {
      continue;
    }
    if (lds > (size_t)160 * 1024)
      return fail(c, "op %d raised the fp16 range flag and its three-piece launch needs %zu bytes of LDS (a CU has 163840): "
                     "re-create the handle with UDA_PW_SCHEME=bf16x3", i, lds);
    uint16_t* d = nullptr;
    HIPC(c, hipMalloc((void**)&d, packed.size() * sizeof(uint16_t)));
    HIPC(c, hipMemcpy(d, packed.data(), packed.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    if (c->wovr[i]) hipFree(c->wovr[i]);
    c->wovr[i] = d;
    c->wovr_par[i] = par;
    c->wscheme[i] = sch;
    c->wunscale[i] = 1.0f;
    ++c->range_demotions;
    ++done;
    fprintf(stderr, "[uda] fp16 range: op %d (kind %d, %d -> %d channels) saw an operand above 65504 and now runs on three bf16 pieces "
                    "(demotion %lld of this handle); the run is served again\n", i, o.kind, K, Nn, (long long)c->range_demotions);
  }
  (void)done;      // serving the run again is all that is left to do in that case
  return 0;
}

// Serve run `r` again, synchronously, into whatever output set / scales the context currently points at.
static int replay_run(uda_ctx* c, const uda_ctx::RunRec& r) {
  const bool have_u8 = c->have_u8, inj = c->masks_injected;
  const int cur = c->cur, n = c->n_images;
  const uint64_t seed = c->seed;
  const int64_t off = c->image_offset;
  c->have_u8 = r.have_u8; c->cur = r.cur; c->n_images = r.n; c->seed = r.seed; c->image_offset = r.image_offset;
  c->masks_injected = r.masks_injected;
  c->pfx_pending.clear();
  const int pfx_skip = c->pfx_skip;
  c->pfx_skip = 1 << 20;           // no score-prefix NMS in the replay: nothing is left pending for the host behind it
  // The overflowed pass left infinities / NaNs in the arena.  Some kernels read a few floats past the end of their input
  // (the k-padding of the last pixel's matrix fragment, a dead lane's clamped window) and multiply them by zero weights:
  // harmless on the finite leftovers of ordinary runs, NaN on these - and one NaN pixel reaches the whole image through
  // the squeeze-excite mean.  The arenas are cleared before the run is served again (a few ms, once per demotion).
  for (int l = 0; l < c->n_lanes; ++l)
    if (c->lane_arena[l]) HIPC(c, hipMemsetAsync(c->lane_arena[l], 0, (size_t)c->model.arena_floats * sizeof(float), c->stream));
  int rc = run_network(c, 0, false, nullptr);
  if (!rc && r.do_post) rc = run_post(c, r.n, r.pm);
  c->pfx_skip = pfx_skip;
  if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(c, "replay_run: the stream failed");
  c->have_u8 = have_u8; c->cur = cur; c->n_images = n; c->seed = seed; c->image_offset = off; c->masks_injected = inj;
  return rc;
}

// first raised flag word of array `half` (or -1), cleared once read
static int read_range_flags(uda_ctx* c, int half, int* first) {
  const size_t nw = c->ops.size() + 1;
  std::vector<unsigned> f(nw);
  unsigned* base = c->d_oor + (size_t)half * nw;
  HIPC(c, hipMemcpy(f.data(), base, nw * sizeof(unsigned), hipMemcpyDeviceToHost));
  *first = -1;
  for (size_t i = 0; i < nw; ++i)
    if (f[i]) { *first = (int)i; break; }
  if (*first >= 0) HIPC(c, hipMemset(base, 0, nw * sizeof(unsigned)));
  return 0;
}

static int check_split_range(uda_ctx* c) {
  if (!c->oor_armed) return 0;
  c->oor_armed = false;
  HIPC(c, hipStreamSynchronize(c->stream));
  for (size_t round = 0; round <= c->ops.size(); ++round) {
    int first = -1;
    if (int rc = read_range_flags(c, c->oor_half, &first)) return rc;
    if (first < 0) return 0;
    if (int rc = demote_ops(c, first)) return rc;
    if (!can_replay(c, c->replay_rec))
      return fail(c, "an activation above 65504 reached op %d, whose operands were split into fp16 pieces: the results of this run are "
                     "invalid and its inputs have been replaced since, so it cannot be served again here.  The op now runs on three "
                     "bf16 pieces: run the batch again", first);
    unsigned* keep = c->oor_cur;
    c->oor_cur = c->d_oor + (size_t)c->oor_half * (c->ops.size() + 1);
    const int rc = replay_run(c, *c->replay_rec);
    c->oor_cur = keep;
    c->oor_armed = false;
    if (rc) return rc;
  }
  return fail(c, "fp16 range flags persist after every op was re-packed: internal error");
}

// Every reader of the post-process outputs comes through here: images whose score prefix turned out not to be
// sufficient (flag written by prefix_check_kernel) are redone on the full candidate set before anything is read.
// Pipelined runs whose post-process the main stream has not been made to wait for: every reader / writer of what they
// touch comes through here first (then the main stream is ordered behind them and everything below works as it always did).
static int join_async(uda_ctx* c) {
  for (int s = 0; s < 2; ++s) {
    uda_ctx::AsyncSlot& a = c->as[s];
    if (a.open && !a.joined) {
      HIPC(c, hipStreamWaitEvent(c->stream, a.ev, 0));
      a.joined = true;
    }
  }
  c->gate_ev = nullptr;
  return 0;
}

static int finish_post(uda_ctx* c) {
  if (int rc = join_async(c)) return rc;
  {
    // an ordinary reader (uda_get_detections, calibrators, ...) beside an uncollected pipelined run reads the NEWEST run: that
    // run's range / barrier / prefix flags are checked here, exactly as uda_collect would check them
    int s = -1;
    for (int i = 0; i < 2; ++i)
      if (c->as[i].open && (s < 0 || c->as[i].seq > c->as[s].seq)) s = i;
    if (s >= 0) {
      uda_ctx::AsyncSlot& a = c->as[s];
      if (a.coop_used || a.oor_armed || !a.pending.empty()) {
        c->d_oboxes = a.oboxes; c->d_oscores = a.oscores; c->d_oclasses = a.oclasses; c->d_ologits = a.ologits; c->d_ovalid = a.ovalid;
        c->d_scales_post = a.scales;
        c->last_n = a.n; c->last_post_mode = a.mode;
        c->coop_used = c->coop_used || a.coop_used;
        c->oor_armed = c->oor_armed || a.oor_armed;
        c->oor_half = s;
        c->replay_rec = &a.rec;
        c->pfx_pending.insert(c->pfx_pending.end(), a.pending.begin(), a.pending.end());
        a.coop_used = false; a.oor_armed = false; a.pending.clear();
      }
    }
  }
  if (int rc = check_split_range(c)) return rc;
  if (c->coop_used) {        // a barrier of the cooperative NMS that timed out leaves garbage: fail loudly
    c->coop_used = false;
    HIPC(c, hipStreamSynchronize(c->stream));
    int e = 0;
    HIPC(c, hipMemcpy(&e, c->d_coop_err, sizeof(int), hipMemcpyDeviceToHost));
    if (e) {
      // A barrier timed out: the blocks of a problem were not all resident (another process holding part of the GPU
      // with a grid of the same kind).  The outputs are garbage: redo the whole post-process with the two-launch
      // version, and keep this handle on it.
      hipMemset(c->d_coop_err, 0, sizeof(int));
      c->coop_off = true;
      ++c->coop_fallbacks;
      c->pfx_pending.clear();
      fprintf(stderr, "[uda] cooperative NMS: grid barrier timed out; falling back to two launches per epoch\n");
      if (c->last_post_mode == UDA_POST_GLOBAL) {
        c->pfx_off = true;
        const int rc = run_post_global(c, 0, c->last_n, c->stream);
        c->pfx_off = false;
        if (rc) return rc;
        HIPC(c, hipStreamSynchronize(c->stream));
      }
      return 0;
    }
  }
  if (c->pfx_pending.empty()) return 0;
  HIPC(c, hipStreamSynchronize(c->stream));
  std::vector<std::pair<int, int>> pend;
  pend.swap(c->pfx_pending);
  std::vector<int32_t> bad((size_t)c->model.max_images, 0);
  for (const auto& r : pend)
    HIPC(c, hipMemcpy(bad.data() + r.first, c->pfx.bad + r.first, (size_t)r.second * sizeof(int32_t), hipMemcpyDeviceToHost));
  int rc = 0, n_bad = 0, n_all = 0;
  c->pfx_off = true;
  for (const auto& r : pend) {
    n_all += r.second;
    for (int i = r.first; i < r.first + r.second && !rc;) {
      if (!bad[(size_t)i]) { ++i; continue; }
      int j = i;
      while (j < r.first + r.second && bad[(size_t)j]) ++j;    // a run of rejected images: one batched pass
      rc = run_post_global(c, i, j - i, c->stream);
      n_bad += j - i;
      i = j;
    }
  }
  c->pfx_off = false;
  c->pfx_fallbacks += n_bad;
  // Score distributions in which most of the anchor set stays in play (near-tied scores of an untrained head) gain
  // nothing from the prefix: pause it for 32, 64, ... 1024 runs, then probe again.
  if (2 * n_bad > n_all) {
    c->pfx_backoff = c->pfx_backoff ? (c->pfx_backoff < 1024 ? 2 * c->pfx_backoff : 1024) : 32;
    c->pfx_skip = c->pfx_backoff;
  } else {
    c->pfx_backoff = 0;
  }
  return rc;
}

static int use_output_set(uda_ctx* c, int s);

extern "C" int uda_run(uda_ctx_t* c, int32_t post_mode, int32_t do_post) {
  if (!c) return 1;
  if (c->n_images < 1) return fail(c, "uda_run: no images set");
  HIPC(c, hipSetDevice(c->device));
  if (c->as[0].open || c->as[1].open) return fail(c, "uda_run: a pipelined run (uda_run_async) is in flight - uda_collect it first");
  if (c->as_ready) { if (int rc = use_output_set(c, 0)) return rc; }
  c->d_scales_post = nullptr;
  c->pfx_pending.clear();
  if (c->oor_armed) {       // flags of a run nobody read: they are not this run's
    hipMemsetAsync(c->d_oor, 0, (c->ops.size() + 1) * sizeof(unsigned), c->stream);
    c->oor_armed = false;
  }
  c->oor_cur = c->d_oor;
  c->oor_half = 0;
  current_run_rec(c, &c->last_run, post_mode < 0 ? c->model.post_mode : post_mode, do_post != 0);
  c->replay_rec = &c->last_run;
  if (do_post && c->post_overlap) {
    const int pm = post_mode < 0 ? c->model.post_mode : post_mode;
    if (pm != UDA_POST_GLOBAL && pm != UDA_POST_PER_CLASS) return fail(c, "unknown post mode %d", pm);
    int rc = prepare_post(c, pm);
    if (rc) return rc;
    rc = run_network(c, pm, true);
    if (!rc && c->pfx_skip > 0 && c->pfx_pending.empty()) --c->pfx_skip;
    return rc;
  }
  int rc = run_network(c);
  if (rc) return rc;
  if (do_post) rc = run_post(c, c->n_images, post_mode);
  return rc;
}

// ---- pipelined runs
static int use_output_set(uda_ctx* c, int s) {
  const uda_ctx::AsyncSlot& a = c->as[s];
  c->d_oboxes = a.oboxes; c->d_oscores = a.oscores; c->d_oclasses = a.oclasses; c->d_ologits = a.ologits; c->d_ovalid = a.ovalid;
  return 0;
}

static int async_setup(uda_ctx* c) {
  if (c->as_ready) return 0;
  const uda_model_t& m = c->model;
  const size_t N = (size_t)m.max_images, M = (size_t)m.max_output_size, C = (size_t)m.num_classes;
  uda_ctx::AsyncSlot& a0 = c->as[0];
  uda_ctx::AsyncSlot& a1 = c->as[1];
  a0.oboxes = c->d_oboxes; a0.oscores = c->d_oscores; a0.oclasses = c->d_oclasses; a0.ologits = c->d_ologits; a0.ovalid = c->d_ovalid;
  HIPC(c, dalloc(&a1.oboxes, N * M * 12));
  HIPC(c, dalloc(&a1.oscores, N * M));
  HIPC(c, dalloc(&a1.oclasses, N * M * (1 + C)));
  HIPC(c, dalloc(&a1.ologits, N * M * C));
  HIPC(c, dalloc(&a1.ovalid, N));
  for (int s = 0; s < 2; ++s) {
    HIPC(c, dalloc(&c->as[s].scales, N));
    HIPC(c, hipEventCreateWithFlags(&c->as[s].ev, hipEventDisableTiming));
  }
  c->as_ready = true;
  return 0;
}

// The heavy path of a slot (range / barrier flags, prefix redo: finish_post): only when nothing newer is queued behind it.
static int resolve_slot_full(uda_ctx* c, int s) {
  uda_ctx::AsyncSlot& a = c->as[s];
  use_output_set(c, s);
  c->d_scales_post = a.scales;
  c->last_n = a.n; c->last_post_mode = a.mode;
  c->coop_used = a.coop_used; c->oor_armed = a.oor_armed;
  c->oor_half = s;
  c->replay_rec = &a.rec;
  if (a.cands_lost) {
    // an older run was served again after this one had been queued: the candidates / NMS state of this run are gone, so a
    // post-process that would have to be redone from them cannot be - say so instead of redoing it on another batch's data
    a.cands_lost = false;
    HIPC(c, hipStreamSynchronize(c->stream));
    int e = 0;
    if (a.coop_used) HIPC(c, hipMemcpy(&e, c->d_coop_err, sizeof(int), hipMemcpyDeviceToHost));
    if (e || !a.pending.empty()) {
      if (e) { hipMemset(c->d_coop_err, 0, sizeof(int)); c->coop_off = true; ++c->coop_fallbacks; }
      a.pending.clear(); a.coop_used = false; a.oor_armed = false;
      return fail(c, "a pipelined run needs its post-process redone, but an older run was served again in between (fp16 range "
                     "demotion) and replaced its candidates: run the batch again");
    }
    c->coop_used = false;
  }
  c->pfx_pending.swap(a.pending);
  a.pending.clear(); a.coop_used = false; a.oor_armed = false;
  return finish_post(c);         // joins, synchronises the main stream, redoes what has to be redone
}

extern "C" int uda_run_async(uda_ctx_t* c, int32_t post_mode, int32_t* ticket) {
  if (!c || !ticket) return c ? fail(c, "uda_run_async: NULL ticket") : 1;
  if (c->n_images < 1) return fail(c, "uda_run_async: no images set");
  HIPC(c, hipSetDevice(c->device));
  const int pm = post_mode < 0 ? c->model.post_mode : post_mode;
  if (pm != UDA_POST_GLOBAL && pm != UDA_POST_PER_CLASS) return fail(c, "unknown post mode %d", pm);
  if (int rc = async_setup(c)) return rc;
  const int s = c->as_next, o = s ^ 1;
  if (c->as[s].open) return fail(c, "uda_run_async: two runs are in flight already - collect ticket %d first", s);
  // leftovers of the run in flight that need the host (prefix-NMS ranges to check: only without the cooperative NMS) are
  // settled before anything new is queued behind it - its candidates are still in place now
  if (c->as[o].open && !c->as[o].pending.empty()) {
    if (int rc = resolve_slot_full(c, o)) return rc;
  }
  int rc = prepare_post(c, pm);
  if (rc) return rc;
  uda_ctx::AsyncSlot& a = c->as[s];
  use_output_set(c, s);
  if (c->oor_armed) {       // flags of a synchronous run nobody read: they are not this run's
    hipMemsetAsync(c->oor_cur, 0, (c->ops.size() + 1) * sizeof(unsigned), c->stream);
    c->oor_armed = false;
  }
  c->oor_cur = c->d_oor + (size_t)s * (c->ops.size() + 1);
  a.cands_lost = false;
  current_run_rec(c, &a.rec, pm, true);
  HIPC(c, hipMemcpyAsync(a.scales, c->d_scales, (size_t)c->n_images * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  c->d_scales_post = a.scales;
  c->pfx_pending.clear();
  c->coop_used = false;
  c->gate_ev = (c->as[o].open && !c->as[o].joined) ? c->as[o].ev : nullptr;
  if (c->post_overlap) {
    rc = run_network(c, pm, true, a.ev);
    a.joined = false;
  } else {                       // UDA_POST_OVERLAP=0: everything on the main stream, nothing to overlap
    rc = run_network(c);
    if (!rc) rc = run_post(c, c->n_images, pm);
    if (!rc) HIPC(c, hipEventRecord(a.ev, c->stream));
    a.joined = true;
  }
  c->gate_ev = nullptr;
  if (rc) return rc;
  if (c->pfx_skip > 0 && c->pfx_pending.empty()) --c->pfx_skip;
  a.open = true;
  a.seq = ++c->as_seq;
  a.n = c->n_images; a.mode = pm;
  a.coop_used = c->coop_used; c->coop_used = false;
  a.oor_armed = c->oor_armed; c->oor_armed = false;
  a.pending.swap(c->pfx_pending);
  c->pfx_pending.clear();
  c->as_next = o;
  *ticket = s;
  return 0;
}

// Waits for the post-process of the run behind `ticket` and checks its flags; the slot's output set is valid afterwards.
static int settle_ticket(uda_ctx* c, int ticket, const char* who) {
  if (ticket < 0 || ticket > 1 || !c->as[ticket].open) return fail(c, "%s: ticket %d is not in flight", who, ticket);
  HIPC(c, hipSetDevice(c->device));
  uda_ctx::AsyncSlot& a = c->as[ticket];
  const uda_ctx::AsyncSlot& other = c->as[ticket ^ 1];
  const bool newer_queued = other.open && other.seq > a.seq;
  if (!newer_queued) {
    // the newest run: the ordinary path (it may redo the post-process on the full candidate set)
    if (int rc = resolve_slot_full(c, ticket)) { a.open = false; return rc; }
    HIPC(c, hipStreamSynchronize(c->stream));
    return 0;
  }
  // a newer run is queued behind this one: wait for this run's post-process only, check its flags, never touch the streams
  HIPC(c, hipEventSynchronize(a.ev));
  if (a.oor_armed) {
    a.oor_armed = false;
    int first = -1;
    if (int rc = read_range_flags(c, ticket, &first)) { a.open = false; return rc; }
    if (first >= 0) {
      // This run split an operand above 65504.  Everything in flight is let finish (the newer run keeps its own flags and its
      // own outputs), the op is re-packed and THIS run is served again into its own output set; the newer run's candidates
      // are overwritten by that (cands_lost: its detections stay valid, a redo of its post-process would not be).
      if (int rc = join_async(c)) { a.open = false; return rc; }
      HIPC(c, hipStreamSynchronize(c->stream));
      c->as[ticket ^ 1].cands_lost = true;
      use_output_set(c, ticket);
      c->d_scales_post = a.scales;
      c->last_n = a.n; c->last_post_mode = a.mode;
      unsigned* keep = c->oor_cur;
      c->oor_cur = c->d_oor + (size_t)ticket * (c->ops.size() + 1);
      int rc = 0;
      for (size_t round = 0; round <= c->ops.size() && first >= 0 && !rc; ++round) {
        rc = demote_ops(c, first);
        if (!rc && !can_replay(c, &a.rec))
          rc = fail(c, "an activation above 65504 reached op %d (fp16 pieces) in a pipelined run whose inputs have been replaced since: "
                       "its results are invalid.  The op now runs on three bf16 pieces: run the batch again", first);
        if (!rc) rc = replay_run(c, a.rec);
        if (!rc) rc = read_range_flags(c, ticket, &first);
      }
      c->oor_cur = keep;
      c->oor_armed = false;
      a.coop_used = a.coop_used || c->coop_used;     // (a barrier time-out of the replay's NMS is looked at below)
      c->coop_used = false;
      if (rc) { a.open = false; return rc; }
    }
  }
  if (a.coop_used) {
    int e = 0;
    HIPC(c, hipMemcpy(&e, c->d_coop_err, sizeof(int), hipMemcpyDeviceToHost));
    if (e) {
      a.open = false;
      c->coop_off = true;
      ++c->coop_fallbacks;
      return fail(c, "cooperative NMS: a grid barrier timed out in a pipelined run whose candidates the next run has already "
                     "replaced - its detections are invalid.  The handle now uses the two-launch NMS; run the batch again");
    }
  }
  if (!a.pending.empty()) { a.open = false; return fail(c, "%s: internal: unsettled prefix-NMS ranges behind a newer run", who); }
  return 0;
}

extern "C" int uda_collect(uda_ctx_t* c, int32_t ticket, float* boxes, float* scores, float* classes, int32_t* valid, float* logits) {
  if (!c) return 1;
  if (int rc = settle_ticket(c, ticket, "uda_collect")) return rc;
  uda_ctx::AsyncSlot& a = c->as[ticket];
  const size_t n = a.n, M = c->model.max_output_size, C = c->model.num_classes;
  const int bc = box_cols_of(c->model, a.mode), cc = cls_cols_of(c->model, a.mode);
  if (boxes) HIPC(c, hipMemcpy(boxes, a.oboxes, n * M * bc * sizeof(float), hipMemcpyDeviceToHost));
  if (scores) HIPC(c, hipMemcpy(scores, a.oscores, n * M * sizeof(float), hipMemcpyDeviceToHost));
  if (classes) HIPC(c, hipMemcpy(classes, a.oclasses, n * M * cc * sizeof(float), hipMemcpyDeviceToHost));
  if (valid) HIPC(c, hipMemcpy(valid, a.ovalid, n * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (logits && a.mode == UDA_POST_GLOBAL) HIPC(c, hipMemcpy(logits, a.ologits, n * M * C * sizeof(float), hipMemcpyDeviceToHost));
  a.open = false;
  return 0;
}

extern "C" int uda_collect_device(uda_ctx_t* c, int32_t ticket, int32_t rows, int32_t with_logits, void** dev_ptr, int32_t* cols) {
  if (!c || !dev_ptr) return c ? fail(c, "uda_collect_device: NULL argument") : 1;
  if (ticket >= 0 && ticket <= 1 && c->as[ticket].open && (rows < c->as[ticket].n || rows > c->model.max_images))
    return fail(c, "uda_collect_device: rows %d outside [%d images of the run, max_images %d]", rows, c->as[ticket].n, c->model.max_images);
  if (int rc = settle_ticket(c, ticket, "uda_collect_device")) return rc;
  uda_ctx::AsyncSlot& a = c->as[ticket];
  const size_t M = c->model.max_output_size;
  const int bc = box_cols_of(c->model, a.mode), cc = cls_cols_of(c->model, a.mode);
  const int C = (with_logits && a.mode == UDA_POST_GLOBAL) ? c->model.num_classes : 0;
  const int nc = bc + 1 + cc + C + 1;
  if (!c->d_opacked) {
    const int widest = box_cols_of(c->model, UDA_POST_GLOBAL) + 1 + cls_cols_of(c->model, UDA_POST_GLOBAL) + c->model.num_classes + 1;
    HIPC(c, dalloc(&c->d_opacked, (size_t)c->model.max_images * M * widest));
  }
  if (!c->aux_stream) HIPC(c, hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
  PackDetArgs p{};
  p.boxes = a.oboxes; p.scores = a.oscores; p.classes = a.oclasses; p.logits = a.ologits; p.valid = a.ovalid;
  p.out = c->d_opacked;
  p.n = a.n; p.rows_out = rows; p.M = (int)M; p.bc = bc; p.cc = cc; p.C = C; p.cols = nc;
  launch_pack_det(p, c->aux_stream);     // (its own stream: the main and the post stream carry the newer run)
  HIPC(c, hipGetLastError());
  HIPC(c, hipStreamSynchronize(c->aux_stream));
  *dev_ptr = c->d_opacked;
  if (cols) *cols = nc;
  a.open = false;
  return 0;
}

// Abandon every pipelined run in flight: wait for the device, discard the results, close the tickets, clear what the runs
// left behind (range flags, a timed-out barrier) - the handle is as after the last collect.  A generator that is dropped
// half way (ServingDriver.serve_stream) or a collect that failed leaves runs open; every later synchronous entry point
// would refuse ("a pipelined run is in flight") until they are drained.
extern "C" int uda_drain(uda_ctx_t* c) {
  if (!c) return 1;
  if (!c->as[0].open && !c->as[1].open) return 0;
  HIPC(c, hipSetDevice(c->device));
  if (int rc = join_async(c)) return rc;
  HIPC(c, hipStreamSynchronize(c->stream));
  const size_t nw = c->ops.size() + 1;
  for (int s = 0; s < 2; ++s) {
    uda_ctx::AsyncSlot& a = c->as[s];
    if (!a.open) continue;
    if (a.oor_armed && c->d_oor) HIPC(c, hipMemset(c->d_oor + (size_t)s * nw, 0, nw * sizeof(unsigned)));
    if (a.coop_used) {
      int e = 0;
      HIPC(c, hipMemcpy(&e, c->d_coop_err, sizeof(int), hipMemcpyDeviceToHost));
      if (e) { HIPC(c, hipMemset(c->d_coop_err, 0, sizeof(int))); c->coop_off = true; ++c->coop_fallbacks; }
    }
    a.open = false; a.joined = true; a.coop_used = false; a.oor_armed = false; a.cands_lost = false;
    a.pending.clear();
  }
  c->pfx_pending.clear();
  c->coop_used = false;
  c->oor_armed = false;
  c->replay_rec = nullptr;
  c->d_scales_post = nullptr;
  return 0;
}

extern "C" int64_t uda_range_demotions(const uda_ctx_t* c) { return c ? c->range_demotions : -1; }
extern "C" int64_t uda_nms_prefix_fallbacks(const uda_ctx_t* c) { return c ? c->pfx_fallbacks : -1; }
extern "C" int64_t uda_nms_coop_fallbacks(const uda_ctx_t* c) { return c ? c->coop_fallbacks : -1; }
extern "C" int64_t uda_nms_coop_not_launched(const uda_ctx_t* c) { return c ? c->coop_not_launched : -1; }

extern "C" int uda_synchronize(uda_ctx_t* c) {
  if (!c) return 1;
  HIPC(c, hipSetDevice(c->device));
  if (int rc = finish_post(c)) return rc;
  HIPC(c, hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int uda_get_detections(uda_ctx_t* c, float* boxes, float* scores, float* classes,
                                  int32_t* valid, float* logits) {
  if (!c) return 1;
  HIPC(c, hipSetDevice(c->device));
  if (int rc = finish_post(c)) return rc;
  HIPC(c, hipStreamSynchronize(c->stream));
  const size_t n = c->last_n, M = c->model.max_output_size, C = c->model.num_classes;
  const int bc = box_cols_of(c->model, c->last_post_mode), cc = cls_cols_of(c->model, c->last_post_mode);
  if (boxes) HIPC(c, hipMemcpy(boxes, c->d_oboxes, n * M * bc * sizeof(float), hipMemcpyDeviceToHost));
  if (scores) HIPC(c, hipMemcpy(scores, c->d_oscores, n * M * sizeof(float), hipMemcpyDeviceToHost));
  if (classes) HIPC(c, hipMemcpy(classes, c->d_oclasses, n * M * cc * sizeof(float), hipMemcpyDeviceToHost));
  if (valid) HIPC(c, hipMemcpy(valid, c->d_ovalid, n * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (logits && c->last_post_mode == UDA_POST_GLOBAL)
    HIPC(c, hipMemcpy(logits, c->d_ologits, n * M * C * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int uda_detections_device(uda_ctx_t* c, int32_t rows, int32_t with_logits, void** dev_ptr, int32_t* cols) {
  if (!c || !dev_ptr) return c ? fail(c, "detections_device: NULL argument") : 1;
  const size_t n = c->last_n, M = c->model.max_output_size;
  if (rows < (int32_t)n || rows > c->model.max_images)
    return fail(c, "detections_device: rows %d outside [%d images of the last post-process, max_images %d]", rows, (int)n, c->model.max_images);
  HIPC(c, hipSetDevice(c->device));
  if (int rc = finish_post(c)) return rc;
  const int bc = box_cols_of(c->model, c->last_post_mode), cc = cls_cols_of(c->model, c->last_post_mode);
  const int C = (with_logits && c->last_post_mode == UDA_POST_GLOBAL) ? c->model.num_classes : 0;
  const int nc = bc + 1 + cc + C + 1;
  if (!c->d_opacked) {      // sized once for the widest record of this handle
    const int widest = box_cols_of(c->model, UDA_POST_GLOBAL) + 1 + cls_cols_of(c->model, UDA_POST_GLOBAL) + c->model.num_classes + 1;
    HIPC(c, dalloc(&c->d_opacked, (size_t)c->model.max_images * M * widest));
  }
  PackDetArgs a{};
  a.boxes = c->d_oboxes; a.scores = c->d_oscores; a.classes = c->d_oclasses; a.logits = c->d_ologits; a.valid = c->d_ovalid;
  a.out = c->d_opacked;
  a.n = (int)n; a.rows_out = rows; a.M = (int)M; a.bc = bc; a.cc = cc; a.C = C; a.cols = nc;
  launch_pack_det(a, c->stream);
  HIPC(c, hipGetLastError());
  HIPC(c, hipStreamSynchronize(c->stream));      // another stream (the process group's) reads the buffer next
  *dev_ptr = c->d_opacked;
  if (cols) *cols = nc;
  return 0;
}

extern "C" int uda_get_class_probs(uda_ctx_t* c, float* probs, float* entropy) {
  if (!c || !probs || !entropy) return c ? fail(c, "get_class_probs: NULL argument") : 1;
  if (c->last_post_mode != UDA_POST_GLOBAL) return fail(c, "get_class_probs: logits exist only after the global post-process");
  HIPC(c, hipSetDevice(c->device));
  if (int rc = finish_post(c)) return rc;
  const size_t n = c->last_n, M = c->model.max_output_size, C = c->model.num_classes;
  if (!c->d_oprobs) {
    const size_t N = (size_t)c->model.max_images;
    HIPC(c, dalloc(&c->d_oprobs, N * M * C));
    HIPC(c, dalloc(&c->d_oentropy, N * M));
  }
  launch_probs(c->d_ologits, c->d_oprobs, c->d_oentropy, (int)(n * M), (int)C, c->stream);
  HIPC(c, hipStreamSynchronize(c->stream));
  HIPC(c, hipGetLastError());
  HIPC(c, hipMemcpy(probs, c->d_oprobs, n * M * C * sizeof(float), hipMemcpyDeviceToHost));
  HIPC(c, hipMemcpy(entropy, c->d_oentropy, n * M * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int uda_calibrate_box(uda_ctx_t* c, int32_t col0, int32_t mode, int32_t relative, int32_t n_tables,
                                 const int32_t* tab_off, const double* xs, const double* ys, const float* temps, float* out) {
  if (!c || !out) return c ? fail(c, "calibrate_box: NULL out") : 1;
  if (c->last_post_mode != UDA_POST_GLOBAL) return fail(c, "calibrate_box: needs the global post-process (uncertainty columns)");
  const int bc = box_cols_of(c->model, UDA_POST_GLOBAL), cc = cls_cols_of(c->model, UDA_POST_GLOBAL);
  if (col0 < 4 || col0 + 4 > bc || (col0 & 3)) return fail(c, "calibrate_box: columns %d..%d outside the %d box columns", col0, col0 + 3, bc);
  const bool iso = mode >= UDA_CALIB_ISO_ALL;
  if (mode < 0 || mode > UDA_CALIB_ISO_PERCLSCOO) return fail(c, "calibrate_box: unknown mode %d", mode);
  if (!iso && !temps) return fail(c, "calibrate_box: temperature scaling needs temps");
  if (relative && mode != UDA_CALIB_ISO_PERCLSCOO) return fail(c, "calibrate_box: the relative variant exists per class and coordinate only");
  const int want = mode == UDA_CALIB_ISO_ALL ? 1 : (mode == UDA_CALIB_ISO_PERCOO ? 4 : 4 * c->model.num_classes);
  if (iso && (n_tables != want || !tab_off || !xs || !ys))
    return fail(c, "calibrate_box: mode %d needs %d tables, got %d", mode, want, n_tables);
  HIPC(c, hipSetDevice(c->device));
  if (int rc = finish_post(c)) return rc;
  CalibArgs a{};
  double *d_xs = nullptr, *d_ys = nullptr;
  int32_t* d_off = nullptr;
  float* d_out = nullptr;
  const size_t rows = (size_t)c->last_n * c->model.max_output_size;
  if (iso) {
    const size_t tot = (size_t)tab_off[n_tables];
    for (int t = 0; t < n_tables; ++t)
      if (tab_off[t + 1] < tab_off[t]) return fail(c, "calibrate_box: table offsets must be non-decreasing");
    HIPC(c, dalloc(&d_xs, tot)); HIPC(c, dalloc(&d_ys, tot)); HIPC(c, dalloc(&d_off, (size_t)n_tables + 1));
    HIPC(c, hipMemcpyAsync(d_xs, xs, tot * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPC(c, hipMemcpyAsync(d_ys, ys, tot * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPC(c, hipMemcpyAsync(d_off, tab_off, ((size_t)n_tables + 1) * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  }
  HIPC(c, dalloc(&d_out, rows * 4));
  a.boxes = c->d_oboxes; a.classes = c->d_oclasses; a.out = d_out;
  a.xs = d_xs; a.ys = d_ys; a.tab_off = d_off;
  for (int j = 0; j < 4; ++j) a.temps[j] = temps ? temps[mode == UDA_CALIB_TS_ALL ? 0 : j] : 1.f;
  a.rows = (int)rows; a.box_cols = bc; a.cls_cols = cc; a.col0 = col0;
  a.mode = mode; a.relative = relative; a.n_tables = n_tables;
  launch_calib(a, c->stream);
  hipError_t e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess) e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpy(out, d_out, rows * 4 * sizeof(float), hipMemcpyDeviceToHost);
  if (d_xs) hipFree(d_xs);
  if (d_ys) hipFree(d_ys);
  if (d_off) hipFree(d_off);
  hipFree(d_out);
  if (e != hipSuccess) return fail(c, "calibrate_box: %s", hipGetErrorString(e));
  return 0;
}

extern "C" int uda_calibrate_class(uda_ctx_t* c, int32_t mode, int32_t n_tables, const int32_t* tab_off, const double* xs,
                                   const double* ys, const float* temps, int32_t draws, uint64_t seed, float* probs,
                                   float* entropy, float* uncert) {
  if (!c || !probs || !entropy) return c ? fail(c, "calibrate_class: NULL output") : 1;
  const uda_model_t& m = c->model;
  if (c->last_post_mode != UDA_POST_GLOBAL || !m.enable_softmax)
    return fail(c, "calibrate_class: needs the logits of the global post-process (enable_softmax)");
  if (mode < UDA_CLS_TS || mode > UDA_CLS_ISO_PERCLS) return fail(c, "calibrate_class: unknown mode %d", mode);
  const int C = m.num_classes;
  if (C > 128) return fail(c, "calibrate_class: more than 128 classes");
  if (mode == UDA_CLS_TS && !temps) return fail(c, "calibrate_class: temperature scaling needs %d temperatures", C);
  const int want = mode == UDA_CLS_ISO_ALL ? 1 : C;
  if (mode != UDA_CLS_TS && (n_tables != want || !tab_off || !xs || !ys))
    return fail(c, "calibrate_class: mode %d needs %d isotonic tables, got %d", mode, want, n_tables);
  const int cc = cls_cols_of(m, UDA_POST_GLOBAL);
  if (draws < 0 || draws > 1000) return fail(c, "calibrate_class: draws %d outside [0, 1000]", draws);
  if (draws > 0 && cc != 1 + C)
    return fail(c, "calibrate_class: sampling needs the MC std of every class logit (MC dropout on the class head, max_nms_inputs = 0)");
  HIPC(c, hipSetDevice(c->device));
  if (int rc = finish_post(c)) return rc;
  const size_t rows = (size_t)c->last_n * m.max_output_size;
  double *d_xs = nullptr, *d_ys = nullptr;
  int32_t* d_off = nullptr;
  float *d_t = nullptr, *d_p = nullptr, *d_e = nullptr, *d_u = nullptr;
  hipError_t e = hipSuccess;
  auto up = [&](auto** dst, const void* src, size_t bytes) {
    if (e != hipSuccess) return;
    e = hipMalloc((void**)dst, bytes ? bytes : 1);
    if (e == hipSuccess && bytes) e = hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, c->stream);
  };
  if (mode != UDA_CLS_TS) {
    for (int t = 0; t < n_tables; ++t)
      if (tab_off[t + 1] <= tab_off[t]) return fail(c, "calibrate_class: every isotonic table needs at least one threshold");
    const size_t tot = (size_t)tab_off[n_tables];
    up(&d_xs, xs, tot * sizeof(double)); up(&d_ys, ys, tot * sizeof(double)); up(&d_off, tab_off, ((size_t)n_tables + 1) * sizeof(int32_t));
  } else {
    up(&d_t, temps, (size_t)C * sizeof(float));
  }
  if (e == hipSuccess) e = hipMalloc((void**)&d_p, rows * C * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void**)&d_e, rows * sizeof(float));
  if (e == hipSuccess && uncert) e = hipMalloc((void**)&d_u, rows * C * sizeof(float));
  if (e == hipSuccess) {
    ClsCalibArgs k{};
    k.logits = c->d_ologits; k.classes = c->d_oclasses; k.probs = d_p; k.entropy = d_e; k.uncert = d_u;
    k.xs = d_xs; k.ys = d_ys; k.tab_off = d_off; k.temps = d_t;
    k.rows = (int)rows; k.C = C; k.cls_cols = cc; k.mode = mode; k.draws = draws; k.seed = seed;
    launch_class_calib(k, c->stream);
    e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(probs, d_p, rows * C * sizeof(float), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(entropy, d_e, rows * sizeof(float), hipMemcpyDeviceToHost);
  if (e == hipSuccess && uncert) {
    if (draws > 0) e = hipMemcpy(uncert, d_u, rows * C * sizeof(float), hipMemcpyDeviceToHost);
    else memset(uncert, 0, rows * C * sizeof(float));
  }
  void* fr[] = {d_xs, d_ys, d_off, d_t, d_p, d_e, d_u};
  for (void* p : fr) if (p) hipFree(p);
  if (e != hipSuccess) return fail(c, "calibrate_class: %s", hipGetErrorString(e));
  return 0;
}

// CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), slicing-by-8 on the host: the per-tensor checksum of TensorFlow
// checkpoint bundles (ckpt_reader.py verifies every tensor it restores; a pure-Python table CRC manages ~1 MB/s).
extern "C" uint32_t uda_crc32c(const void* data, uint64_t n, uint32_t crc) {
  struct Tables {
    uint32_t t[8][256];
    Tables() {
      for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
        t[0][i] = c;
      }
      for (uint32_t i = 0; i < 256; ++i)
        for (int k = 1; k < 8; ++k) t[k][i] = (t[k - 1][i] >> 8) ^ t[0][t[k - 1][i] & 0xFFu];
    }
  };
  static const Tables tables;          // function-local static: initialised once, thread-safe (C++11) - ctypes releases the GIL
  const uint32_t (*T)[256] = tables.t;
  const uint8_t* p = (const uint8_t*)data;
  crc = ~crc;
  while (n >= 8) {
    uint64_t w;
    memcpy(&w, p, 8);
    w ^= crc;
    crc = T[7][w & 0xFF] ^ T[6][(w >> 8) & 0xFF] ^ T[5][(w >> 16) & 0xFF] ^ T[4][(w >> 24) & 0xFF] ^
          T[3][(w >> 32) & 0xFF] ^ T[2][(w >> 40) & 0xFF] ^ T[1][(w >> 48) & 0xFF] ^ T[0][(w >> 56) & 0xFF];
    p += 8; n -= 8;
  }
  while (n--) crc = T[0][(crc ^ *p++) & 0xFFu] ^ (crc >> 8);
  return ~crc;
}

extern "C" int uda_serve(uda_ctx_t* c, const uint8_t* images, int32_t n, int32_t h, int32_t w,
                         float* boxes, float* scores, float* classes, int32_t* valid, float* logits) {
  if (c) { if (int rc_ = no_async(c, "uda_serve")) return rc_; }
  int rc = uda_set_images_u8(c, images, n, h, w);
  if (rc) return rc;
  rc = uda_run(c, -1, 1);
  if (rc) return rc;
  return uda_get_detections(c, boxes, scores, classes, valid, logits);
}

// ------------------------------------------------------------------------------------ head outputs
// device layout [n, Tx, hw, ch]  <->  API layout [Tx, n, hw, ch]
extern "C" int uda_get_head_outputs(uda_ctx_t* c, int32_t level, float* cls, float* box) {
  if (!c) return 1;
  if (level < 0 || level >= c->model.num_levels) return fail(c, "get_head_outputs: bad level %d", level);
  HIPC(c, hipSetDevice(c->device));
  HIPC(c, hipStreamSynchronize(c->stream));
  if (int rc = check_split_range(c)) return rc;
  const uda_model_t& m = c->model;
  const size_t hw = (size_t)m.level_h[level] * m.level_w[level];
  const int n = c->n_images;
  struct { float* dst; const float* src; int T; size_t ch; } jobs[2] = {
      {cls, c->d_cls[level], m.cls_stacked ? m.mc_samples : 1, (size_t)c->cls_ch},
      {box, c->d_box[level], m.box_stacked ? m.mc_samples : 1, (size_t)c->box_ch}};
  for (auto& j : jobs) {
    if (!j.dst) continue;
    const size_t per = hw * j.ch;
    for (int i = 0; i < n; ++i)
      for (int t = 0; t < j.T; ++t)
        HIPC(c, hipMemcpy(j.dst + ((size_t)t * n + i) * per, j.src + ((size_t)i * j.T + t) * per,
                          per * sizeof(float), hipMemcpyDeviceToHost));
  }
  return 0;
}

extern "C" int uda_set_head_outputs(uda_ctx_t* c, int32_t level, int32_t n, const float* cls, int64_t cls_floats,
                                    const float* box, int64_t box_floats) {
  if (c) { if (int rc_ = no_async(c, "uda_set_head_outputs")) return rc_; }
  if (!c) return 1;
  if (level < 0 || level >= c->model.num_levels) return fail(c, "set_head_outputs: bad level %d", level);
  if (n < 1 || n > c->model.max_images) return fail(c, "set_head_outputs: n=%d", n);
  HIPC(c, hipSetDevice(c->device));
  const uda_model_t& m = c->model;
  const size_t hw = (size_t)m.level_h[level] * m.level_w[level];
  {   // the host buffers must hold exactly what is read from them: [T_x, n, h, w, ch] (T_x = 1 for an unstacked head)
    const int64_t want_c = (int64_t)(m.cls_stacked ? m.mc_samples : 1) * n * (int64_t)hw * c->cls_ch;
    const int64_t want_b = (int64_t)(m.box_stacked ? m.mc_samples : 1) * n * (int64_t)hw * c->box_ch;
    if (cls && cls_floats != want_c)
      return fail(c, "set_head_outputs: level %d class outputs hold %lld floats, the handle expects %lld ([%d, %d, %d, %d, %d])", level,
                  (long long)cls_floats, (long long)want_c, m.cls_stacked ? m.mc_samples : 1, n, m.level_h[level], m.level_w[level], c->cls_ch);
    if (box && box_floats != want_b)
      return fail(c, "set_head_outputs: level %d box outputs hold %lld floats, the handle expects %lld ([%d, %d, %d, %d, %d])", level,
                  (long long)box_floats, (long long)want_b, m.box_stacked ? m.mc_samples : 1, n, m.level_h[level], m.level_w[level], c->box_ch);
  }
  struct { const float* src; float* dst; int T; size_t ch; } jobs[2] = {
      {cls, c->d_cls[level], m.cls_stacked ? m.mc_samples : 1, (size_t)c->cls_ch},
      {box, c->d_box[level], m.box_stacked ? m.mc_samples : 1, (size_t)c->box_ch}};
  for (auto& j : jobs) {
    if (!j.src) continue;
    const size_t per = hw * j.ch;
    for (int i = 0; i < n; ++i)
      for (int t = 0; t < j.T; ++t)
        HIPC(c, hipMemcpy(j.dst + ((size_t)i * j.T + t) * per, j.src + ((size_t)t * n + i) * per,
                          per * sizeof(float), hipMemcpyHostToDevice));
  }
  c->n_images = n;
  return 0;
}

extern "C" int uda_head_outputs_device(uda_ctx_t* c, int32_t level, int32_t which, void** dev_ptr, int64_t* floats_per_row,
                                       int32_t* rows_per_image) {
  if (!c || !dev_ptr) return c ? fail(c, "head_outputs_device: NULL argument") : 1;
  if (level < 0 || level >= c->model.num_levels) return fail(c, "head_outputs_device: bad level %d", level);
  if (which != 0 && which != 1) return fail(c, "head_outputs_device: which must be 0 (class) or 1 (box)");
  const uda_model_t& m = c->model;
  const int64_t hw = (int64_t)m.level_h[level] * m.level_w[level];
  *dev_ptr = which ? (void*)c->d_box[level] : (void*)c->d_cls[level];
  if (floats_per_row) *floats_per_row = hw * (which ? c->box_ch : c->cls_ch);
  if (rows_per_image) *rows_per_image = (which ? m.box_stacked : m.cls_stacked) ? m.mc_samples : 1;
  return 0;
}

extern "C" int uda_set_num_images(uda_ctx_t* c, int32_t n) {
  if (c) { if (int rc_ = no_async(c, "uda_set_num_images")) return rc_; }
  if (!c) return 1;
  if (n < 1 || n > c->model.max_images) return fail(c, "set_num_images: n=%d outside [1, %d]", n, c->model.max_images);
  c->n_images = n;
  return 0;
}

extern "C" int uda_copy_heads(uda_ctx_t* dst, const uda_ctx_t* src, int32_t n, int32_t sample) {
  if (dst) { if (int rc_ = no_async(dst, "uda_copy_heads")) return rc_; }
  if (!dst || !src) return 1;
  const uda_model_t& md = dst->model;
  const uda_model_t& ms = src->model;
  if (dst->device != src->device) return fail(dst, "copy_heads: handles live on different devices");
  if (md.num_levels != ms.num_levels || dst->cls_ch != src->cls_ch || dst->box_ch != src->box_ch)
    return fail(dst, "copy_heads: head geometry differs");
  if (ms.cls_stacked || ms.box_stacked) return fail(dst, "copy_heads: the source must be a deterministic (unstacked) network");
  if (!md.cls_stacked || !md.box_stacked) return fail(dst, "copy_heads: the destination must stack both heads");
  if (sample < 0 || sample >= md.mc_samples) return fail(dst, "copy_heads: sample %d outside [0, %d)", sample, md.mc_samples);
  if (n < 1 || n > md.max_images || n > ms.max_images) return fail(dst, "copy_heads: n=%d", n);
  HIPC(dst, hipSetDevice(dst->device));
  HIPC(dst, hipStreamSynchronize(src->stream));       // the member's heads must be complete
  const int T = md.mc_samples;
  for (int l = 0; l < md.num_levels; ++l) {
    if (md.level_h[l] != ms.level_h[l] || md.level_w[l] != ms.level_w[l]) return fail(dst, "copy_heads: level %d size differs", l);
    const size_t hw = (size_t)md.level_h[l] * md.level_w[l];
    const size_t cb = hw * dst->cls_ch * sizeof(float), bb = hw * dst->box_ch * sizeof(float);
    // dst rows are [image][sample]: row pitch T * bytes; src rows are [image]: pitch bytes
    HIPC(dst, hipMemcpy2DAsync(dst->d_cls[l] + (size_t)sample * hw * dst->cls_ch, (size_t)T * cb, src->d_cls[l], cb, cb, n,
                               hipMemcpyDeviceToDevice, dst->stream));
    HIPC(dst, hipMemcpy2DAsync(dst->d_box[l] + (size_t)sample * hw * dst->box_ch, (size_t)T * bb, src->d_box[l], bb, bb, n,
                               hipMemcpyDeviceToDevice, dst->stream));
  }
  dst->n_images = n;
  return 0;
}

extern "C" int uda_postprocess_heads(uda_ctx_t* c, int32_t n, const float* image_scales, int32_t post_mode) {
  if (c) { if (int rc_ = no_async(c, "uda_postprocess_heads")) return rc_; }
  if (!c) return 1;
  if (n < 1 || n > c->model.max_images) return fail(c, "postprocess_heads: n=%d", n);
  HIPC(c, hipSetDevice(c->device));
  for (int i = 0; i < n; ++i) c->h_scales[i] = image_scales ? image_scales[i] : 1.0f;
  HIPC(c, hipMemcpyAsync(c->d_scales, c->h_scales.data(), n * sizeof(float), hipMemcpyHostToDevice, c->stream));
  c->n_images = n;
  return run_post(c, n, post_mode);
}

extern "C" int uda_predict(uda_ctx_t* c, const float* images, int32_t n) {
  if (c) { if (int rc_ = no_async(c, "uda_predict")) return rc_; }
  int rc = uda_set_images_f32(c, images, n, nullptr);
  if (rc) return rc;
  rc = uda_run(c, -1, 0);
  if (rc) return rc;
  return uda_synchronize(c);
}

extern "C" int32_t uda_num_candidates(const uda_ctx_t* c) { return c ? c->Kc : 0; }

extern "C" int uda_get_candidates(uda_ctx_t* c, float* boxes, float* scores, int32_t* classes,
                                  float* u_cls, float* u_al, float* u_ep) {
  if (!c) return 1;
  HIPC(c, hipSetDevice(c->device));
  HIPC(c, hipStreamSynchronize(c->stream));
  const size_t n = c->last_n, K = c->Kc, C = c->model.max_nms_inputs > 0 ? 1 : c->model.num_classes;
  if (boxes) HIPC(c, hipMemcpy(boxes, c->d_cboxes, n * K * 4 * sizeof(float), hipMemcpyDeviceToHost));
  if (scores) HIPC(c, hipMemcpy(scores, c->d_cscores, n * K * sizeof(float), hipMemcpyDeviceToHost));
  if (classes) HIPC(c, hipMemcpy(classes, c->d_cclasses, n * K * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (u_cls && c->d_ucls) HIPC(c, hipMemcpy(u_cls, c->d_ucls, n * K * C * sizeof(float), hipMemcpyDeviceToHost));
  if (u_al && c->d_ual) HIPC(c, hipMemcpy(u_al, c->d_ual, n * K * 4 * sizeof(float), hipMemcpyDeviceToHost));
  if (u_ep && c->d_uep) HIPC(c, hipMemcpy(u_ep, c->d_uep, n * K * 4 * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int uda_read_buffer(uda_ctx_t* c, int32_t buf, float* host, int64_t n_floats) {
  if (!c || !host) return 1;
  if (buf < 0 || buf >= (int)c->bufs.size()) return fail(c, "read_buffer: bad buffer %d", buf);
  HIPC(c, hipSetDevice(c->device));
  HIPC(c, hipStreamSynchronize(c->stream));
  ChunkView v{c, c->last_chunk_i0, c->last_chunk_n};
  v.lane = c->last_lane;
  const uda_buf_desc_t& b = c->bufs[buf];
  if (b.kind == 1 && c->have_u8 && !c->pre_valid) {
    if (int rc = run_preprocess(c)) return rc;
    HIPC(c, hipStreamSynchronize(c->stream));
  }
  const int64_t have = (int64_t)v.rows(b) * b.H * b.W * b.C;
  if (n_floats > have) return fail(c, "read_buffer: asked %lld floats, buffer holds %lld", (long long)n_floats, (long long)have);
  HIPC(c, hipMemcpy(host, v.ptr(buf), n_floats * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int uda_get_preprocessed(uda_ctx_t* c, float* images, float* scales) {
  if (!c) return 1;
  HIPC(c, hipSetDevice(c->device));
  if (images && c->have_u8 && !c->pre_valid)      // the stem read the uint8 batch itself: produce the float image on demand
    if (int rc = run_preprocess(c)) return rc;
  HIPC(c, hipStreamSynchronize(c->stream));
  const size_t n = c->n_images;
  if (images)
    HIPC(c, hipMemcpy(images, c->d_images, n * c->model.image_h * c->model.image_w * 3 * sizeof(float), hipMemcpyDeviceToHost));
  if (scales) memcpy(scales, c->h_scales.data(), n * sizeof(float));
  return 0;
}

// ------------------------------------------------------------------------------------ standalone NMS
extern "C" int uda_nms(uda_ctx_t* c, const float* boxes, const float* scores, int32_t n_img, int32_t k,
                       int32_t max_out, float iou_thresh, float score_thresh, float soft_sigma, int32_t pad,
                       int32_t* idx, float* out_scores, int32_t* valid) {
  if (!c || !boxes || !scores || !idx || !out_scores || !valid) return c ? fail(c, "uda_nms: NULL argument") : 1;
  if (n_img < 1 || k < 0 || max_out < 1 || max_out > 128) return fail(c, "uda_nms: bad sizes (max_out must be in [1, 128])");
  HIPC(c, hipSetDevice(c->device));
  const size_t NK = (size_t)n_img * (k ? k : 1), NM = (size_t)n_img * max_out;
  float *d_boxes, *d_scores, *d_stale, *d_tent, *d_ub, *d_ss, *d_sb;
  int32_t *d_begin, *d_ev, *d_si, *d_nsel, *d_done;
  unsigned long long *d_bound, *d_win;
  HIPC(c, dalloc(&d_boxes, NK * 4)); HIPC(c, dalloc(&d_scores, NK)); HIPC(c, dalloc(&d_stale, NK));
  HIPC(c, dalloc(&d_tent, NK)); HIPC(c, dalloc(&d_ub, NK)); HIPC(c, dalloc(&d_ev, NK)); HIPC(c, dalloc(&d_begin, NK)); HIPC(c, dalloc(&d_si, NM));
  HIPC(c, dalloc(&d_ss, NM)); HIPC(c, dalloc(&d_sb, NM * 4)); HIPC(c, dalloc(&d_bound, NM));
  HIPC(c, dalloc(&d_win, NM)); HIPC(c, dalloc(&d_nsel, (size_t)n_img)); HIPC(c, dalloc(&d_done, (size_t)n_img));
  if (k) {
    HIPC(c, hipMemcpyAsync(d_boxes, boxes, NK * 4 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPC(c, hipMemcpyAsync(d_scores, scores, NK * sizeof(float), hipMemcpyHostToDevice, c->stream));
  }
  NmsArgs a{};
  a.boxes = d_boxes; a.stale = d_stale; a.begin = d_begin; a.tent = d_tent; a.ub = d_ub; a.ev = d_ev;
  a.sel_idx = d_si; a.sel_score = d_ss; a.sel_box = d_sb; a.bound_key = d_bound; a.win_key = d_win;
  a.nsel = d_nsel; a.done = d_done; a.n_img = n_img; a.K = k; a.M = max_out;
  a.segs = 1; a.classes = nullptr;
  nms_params(a, iou_thresh, score_thresh, soft_sigma);
  uda_ctx::PrefixWs pw;
  const int lp = prefix_target();
  if (lp > 0 && k > solo_limit() && k > 2 * lp) HIPC(c, alloc_prefix_ws(pw, (size_t)n_img, 2 * lp, (size_t)max_out));
  bool prefix = false;
  {
    ProfScope ps(c, 17);
    NmsCoop coop;
    if ((size_t)n_img * nms_coop_slot_words(max_out) <= (size_t)c->model.max_images * nms_coop_slot_words(c->model.max_output_size) && !c->coop_off) { coop.bar = c->d_coop_bar; coop.err = c->d_coop_err; coop.used = &c->coop_used; coop.not_launched = &c->coop_not_launched; }
    if (k > 0) prefix = run_nms(a, d_scores, max_out, c->stream, pw.Lcap ? &pw : nullptr, 0, coop);
    else launch_nms_init(a, d_scores, c->stream);
  }
  HIPC(c, hipStreamSynchronize(c->stream));
  HIPC(c, hipGetLastError());
  if (prefix) {            // problems whose prefix was not sufficient: the full candidate set, one problem at a time
    std::vector<int32_t> bad((size_t)n_img);
    HIPC(c, hipMemcpy(bad.data(), pw.bad, (size_t)n_img * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int p = 0; p < n_img; ++p) {
      if (!bad[(size_t)p]) continue;
      const size_t pk = (size_t)p * k, pm = (size_t)p * max_out;
      NmsArgs f = a;
      f.boxes = d_boxes + pk * 4; f.stale = d_stale + pk; f.begin = d_begin + pk; f.tent = d_tent + pk; f.ub = d_ub + pk; f.ev = d_ev + pk;
      f.sel_idx = d_si + pm; f.sel_score = d_ss + pm; f.sel_box = d_sb + pm * 4; f.bound_key = d_bound + pm; f.win_key = d_win + pm;
      f.nsel = d_nsel + p; f.done = d_done + p; f.n_img = 1;
      ProfScope ps(c, 17);
      NmsCoop coop;
      if (!c->coop_off) { coop.bar = c->d_coop_bar; coop.err = c->d_coop_err; coop.used = &c->coop_used; coop.not_launched = &c->coop_not_launched; }
      run_nms(f, d_scores + pk, max_out, c->stream, nullptr, 0, coop);
      ++c->pfx_fallbacks;
    }
    HIPC(c, hipStreamSynchronize(c->stream));
    HIPC(c, hipGetLastError());
  }
  free_prefix_ws(pw);
  if (c->coop_used) {
    c->coop_used = false;
    int e = 0;
    hipMemcpy(&e, c->d_coop_err, sizeof(int), hipMemcpyDeviceToHost);
    if (e) {           // barrier time-out: redo with the two-launch version (see finish_post)
      hipMemset(c->d_coop_err, 0, sizeof(int));
      c->coop_off = true;
      ++c->coop_fallbacks;
      fprintf(stderr, "[uda] cooperative NMS: grid barrier timed out; falling back to two launches per epoch\n");
      run_nms(a, d_scores, max_out, c->stream);
      HIPC(c, hipStreamSynchronize(c->stream));
    }
  }
  HIPC(c, hipMemcpy(valid, d_nsel, n_img * sizeof(int32_t), hipMemcpyDeviceToHost));
  HIPC(c, hipMemcpy(idx, d_si, NM * sizeof(int32_t), hipMemcpyDeviceToHost));
  HIPC(c, hipMemcpy(out_scores, d_ss, NM * sizeof(float), hipMemcpyDeviceToHost));
  (void)pad;  // slots >= valid already hold index 0 / score 0.0 (the padded form); callers slice when pad == 0
  void* frees[] = {d_boxes, d_scores, d_stale, d_tent, d_ub, d_ev, d_begin, d_si, d_ss, d_sb, d_bound, d_win, d_nsel, d_done};
  for (void* p : frees) hipFree(p);
  return 0;
}

// ------------------------------------------------------------------------------------ profiling
extern "C" int uda_profile_enable(uda_ctx_t* c, uint32_t kind_mask) {
  if (!c) return 1;
  c->prof_mask = kind_mask;
  return 0;
}

extern "C" int uda_profile_read(uda_ctx_t* c, int32_t kind, double* total_ms, int64_t* launches, int32_t reset) {
  if (!c || kind < 0 || kind >= 32) return 1;
  hipSetDevice(c->device);
  prof_collect(c, kind);
  if (total_ms) *total_ms = c->prof[kind].total_ms;
  if (launches) *launches = c->prof[kind].launches;
  if (reset) { c->prof[kind].total_ms = 0; c->prof[kind].launches = 0; }
  return 0;
}

// ------------------------------------------------------------------------------------ standalone 1x1 conv
extern "C" int uda_debug_pw(int32_t device, const float* in, const float* w, const float* bias, const float* bn_scale,
                            const float* bn_shift, const float* se, const float* mask, const float* res,
                            int32_t rows, int32_t in_div, int32_t hw, int32_t cin, int32_t cout, int32_t act,
                            int32_t terms, int32_t reps, float* out, float* avg_ms) {
  if (!in || !w || !out || rows < 1 || in_div < 1 || rows % in_div || hw < 1 || cin < 4 || cin % 4 || cout < 1)
    return fail(nullptr, "uda_debug_pw: bad argument");
  if (terms != 0 && terms != 3 && terms != 6 && terms != 16)
    return fail(nullptr, "uda_debug_pw: terms must be 0 (f32 MFMA), 3 (bf16 x2), 6 (bf16 x3) or 16 (fp16 x2)");
  HIPC(nullptr, hipSetDevice(device));
  const size_t rows_in = rows / in_div;
  std::vector<void*> owned;
  auto up = [&](const float* h, size_t n) -> float* {
    if (!h) return nullptr;
    float* d = nullptr;
    if (hipMalloc((void**)&d, n * sizeof(float)) != hipSuccess) return nullptr;
    owned.push_back(d);
    hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice);
    return d;
  };
  PwArgs a{};
  a.in = up(in, rows_in * hw * cin);
  a.w = up(w, (size_t)cin * cout);
  a.bias = up(bias, cout);
  a.bn_scale = up(bn_scale, cout);
  a.bn_shift = up(bn_shift, cout);
  a.se = up(se, rows_in * cin);
  a.mask = up(mask, (size_t)rows * cout);
  a.res = up(res, (size_t)rows * hw * cout);
  float* d_out = nullptr;
  HIPC(nullptr, hipMalloc((void**)&d_out, (size_t)rows * hw * cout * sizeof(float)));
  owned.push_back(d_out);
  a.out = d_out;
  a.HW = hw; a.Cin = cin; a.Cout = cout; a.in_div = in_div; a.res_div = 1; a.se_div = in_div; a.act = act;
  uint16_t* d_ws = nullptr;
  unsigned* d_oor = nullptr;
  if (terms) {
    const int scheme = terms == 6 ? UDA_SPLIT_BF16X3 : (terms == 16 ? UDA_SPLIT_F16X2 : UDA_SPLIT_BF16X2);
    const float scale = scheme == UDA_SPLIT_F16X2 ? split_weight_scale(w, (size_t)cin * cout) : 1.0f;
    std::vector<uint16_t> packed(pwb_packed_elems(cin, cout, scheme));
    pwb_pack_weights(w, cin, cout, scheme, packed.data(), scale);
    HIPC(nullptr, hipMalloc((void**)&d_ws, packed.size() * sizeof(uint16_t)));
    owned.push_back(d_ws);
    HIPC(nullptr, hipMemcpy(d_ws, packed.data(), packed.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    HIPC(nullptr, hipMalloc((void**)&d_oor, sizeof(unsigned)));
    owned.push_back(d_oor);
    HIPC(nullptr, hipMemset(d_oor, 0, sizeof(unsigned)));
    a.wsplit = d_ws;
    a.wparts = scheme;
    a.wunscale = 1.0f / scale;
    a.oor = d_oor;
  }
  hipStream_t st;
  HIPC(nullptr, hipStreamCreate(&st));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto go = [&]() { if (terms) launch_pwb(a, rows, st); else launch_pw(a, rows, st); };
  go();                                   // warm-up (and the result that is read back)
  hipEventRecord(e0, st);
  for (int i = 0; i < reps; ++i) go();
  hipEventRecord(e1, st);
  hipError_t err = hipStreamSynchronize(st);
  if (err == hipSuccess) err = hipGetLastError();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  if (avg_ms) *avg_ms = reps > 0 ? ms / reps : 0.f;
  if (err == hipSuccess) err = hipMemcpy(out, d_out, (size_t)rows * hw * cout * sizeof(float), hipMemcpyDeviceToHost);
  unsigned oor = 0;
  if (err == hipSuccess && d_oor) err = hipMemcpy(&oor, d_oor, sizeof(unsigned), hipMemcpyDeviceToHost);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  hipStreamDestroy(st);
  for (void* p : owned) hipFree(p);
  if (err != hipSuccess) return fail(nullptr, "uda_debug_pw: %s", hipGetErrorString(err));
  if (oor) return fail(nullptr, "uda_debug_pw: an input above 65504 cannot be split into fp16 pieces (terms = 16)");
  return 0;
}

// ------------------------------------------------------------------------------------ numpy NMS family (a18)
template <typename T>
static int run_nmsnp(int device, const std::vector<T>& dets, const std::vector<int32_t>& off, int method, double iou_thr,
                     double sigma, double score_thr, std::vector<T>& out, std::vector<int32_t>& n_out) {
  const int problems = (int)off.size() - 1;
  const size_t total = (size_t)off.back();
  out.assign(total * 5, (T)0);
  n_out.assign(problems > 0 ? problems : 0, 0);
  if (problems <= 0 || total == 0) return 0;
  HIPC(nullptr, hipSetDevice(device));
  T *d_dets = nullptr, *d_score = nullptr, *d_out = nullptr;
  int32_t *d_off = nullptr, *d_state = nullptr, *d_nout = nullptr;
  HIPC(nullptr, dalloc(&d_dets, total * 5)); HIPC(nullptr, dalloc(&d_score, total)); HIPC(nullptr, dalloc(&d_out, total * 5));
  HIPC(nullptr, dalloc(&d_off, off.size())); HIPC(nullptr, dalloc(&d_state, total)); HIPC(nullptr, dalloc(&d_nout, (size_t)problems));
  HIPC(nullptr, hipMemcpy(d_dets, dets.data(), total * 5 * sizeof(T), hipMemcpyHostToDevice));
  HIPC(nullptr, hipMemcpy(d_off, off.data(), off.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  NmsNpArgs<T> a{};
  a.dets = d_dets; a.off = d_off; a.score = d_score; a.state = d_state; a.out = d_out; a.n_out = d_nout;
  a.method = method; a.iou_thr = (T)iou_thr; a.sigma = (T)sigma; a.score_thr = (T)score_thr;
  launch_nmsnp<T>(a, problems, nullptr);
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpy(out.data(), d_out, total * 5 * sizeof(T), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(n_out.data(), d_nout, (size_t)problems * sizeof(int32_t), hipMemcpyDeviceToHost);
  void* fr[] = {d_dets, d_score, d_out, d_off, d_state, d_nout};
  for (void* p : fr) hipFree(p);
  if (e != hipSuccess) return fail(nullptr, "nms_np: %s", hipGetErrorString(e));
  return 0;
}

// rows sorted by score, descending (what `dets[:, 4].argsort()[::-1]` yields for distinct scores; ties: later index first)
template <typename T>
static void sort_desc(std::vector<T>& dets, int begin, int n) {
  std::vector<int> idx(n);
  std::iota(idx.begin(), idx.end(), 0);
  std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) {
    const T sx = dets[(size_t)(begin + x) * 5 + 4], sy = dets[(size_t)(begin + y) * 5 + 4];
    return sx > sy || (sx == sy && x > y);
  });
  std::vector<T> tmp((size_t)n * 5);
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < 5; ++k) tmp[(size_t)i * 5 + k] = dets[(size_t)(begin + idx[i]) * 5 + k];
  std::copy(tmp.begin(), tmp.end(), dets.begin() + (size_t)begin * 5);
}

extern "C" int uda_nms_np(int32_t device, const double* dets, int32_t n, int32_t method, double iou_thresh, double sigma,
                          double score_thresh, double* out, int32_t* n_out) {
  if (!dets || !out || !n_out || n < 0 || method < 0 || method > 3) return fail(nullptr, "uda_nms_np: bad argument");
  std::vector<double> d(dets, dets + (size_t)n * 5), o;
  std::vector<int32_t> off = {0, n}, no;
  if (method <= 1) sort_desc(d, 0, n);
  const int rc = run_nmsnp<double>(device, d, off, method, iou_thresh, sigma, score_thresh, o, no);
  if (rc) return rc;
  *n_out = n ? no[0] : 0;
  std::copy(o.begin(), o.begin() + (size_t)*n_out * 5, out);
  return 0;
}

extern "C" int uda_per_class_nms_np(int32_t device, const float* boxes, const float* scores, const int32_t* classes, int32_t k,
                                    float image_id, float image_scale, int32_t num_classes, int32_t max_boxes, int32_t method,
                                    float iou_thresh, float sigma, float score_thresh, float* out) {
  if (!boxes || !scores || !classes || !out || k < 0 || num_classes < 1 || max_boxes < 1 || method < 0 || method > 3)
    return fail(nullptr, "uda_per_class_nms_np: bad argument");
  std::vector<float> d;
  std::vector<int32_t> off = {0}, cls_of;
  for (int c = 0; c < num_classes; ++c) {
    const int begin = off.back();
    int n = 0;
    for (int i = 0; i < k; ++i)
      if (classes[i] == c) {           // boxes arrive y1,x1,y2,x2 -> x1,y1,x2,y2 (nms_np.py:234)
        d.insert(d.end(), {boxes[i * 4 + 1], boxes[i * 4 + 0], boxes[i * 4 + 3], boxes[i * 4 + 2], scores[i]});
        ++n;
      }
    if (!n) continue;
    if (method <= 1) sort_desc(d, begin, n);
    off.push_back(begin + n);
    cls_of.push_back(c);
  }
  std::vector<float> o;
  std::vector<int32_t> no;
  const int rc = run_nmsnp<float>(device, d, off, method, iou_thresh, sigma, score_thresh, o, no);
  if (rc) return rc;
  struct Row { float v[7]; };
  std::vector<Row> rows;
  for (size_t p = 0; p + 1 < off.size(); ++p)
    for (int i = 0; i < no[p]; ++i) {
      const float* r = o.data() + ((size_t)off[p] + i) * 5;
      rows.push_back(Row{{image_id, r[0], r[1], r[2], r[3], r[4], (float)(cls_of[p] + 1)}});
    }
  std::stable_sort(rows.begin(), rows.end(), [](const Row& x, const Row& y) { return x.v[5] > y.v[5]; });
  for (int i = 0; i < max_boxes; ++i) {
    float* dst = out + (size_t)i * 7;
    if (i < (int)rows.size()) {
      for (int j = 0; j < 7; ++j) dst[j] = rows[i].v[j];
    } else {                           // dummy rows: score -1e5 (nms_np.py:256-274)
      for (int j = 0; j < 7; ++j) dst[j] = 0.f;
      dst[0] = image_id;
      dst[5] = -1e5f;
    }
    for (int j = 1; j < 5; ++j) dst[j] *= image_scale;
  }
  return 0;
}
