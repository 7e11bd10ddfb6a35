// Internal declarations shared by the host executor and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/uda_hip.h"

namespace uda {

// ---------------------------------------------------------------- counter-based random stream (shared with oracle/philox_ref.py)
#ifdef __HIPCC__
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// One standard-normal draw per (i0, i1, i2) of stream `tag`: words of Philox4x32-10(counter = (i0, i1, i2, tag), key = seed),
// u1 = ((w0 >> 8) + 0.5) 2^-24 in (0, 1), u2 = (w1 >> 8) 2^-24, z = sqrt(-2 ln u1) cos(2 pi u2) in float64 (Box-Muller).
// Used where the reference draws from TFP distributions (class-calibration logit samples, utils_class.py:121-123; the
// `sample` decode, utils_box.py:162-184): TF's stream cannot be reproduced, so the build defines this one.
__device__ __forceinline__ double philox_normal(uint64_t seed, uint32_t i0, uint32_t i1, uint32_t i2, uint32_t tag) {
  uint32_t w[4];
  philox4x32_10(i0, i1, i2, tag, (uint32_t)seed, (uint32_t)(seed >> 32), w);
  const double u1 = ((double)(w[0] >> 8) + 0.5) * 5.9604644775390625e-08;
  const double u2 = (double)(w[1] >> 8) * 5.9604644775390625e-08;
  return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
}
// both Box-Muller values of one Philox call: z0 = r cos(2 pi u2), z1 = r sin(2 pi u2)
__device__ __forceinline__ void philox_normal2(uint64_t seed, uint32_t i0, uint32_t i1, uint32_t i2, uint32_t tag, double& z0, double& z1) {
  uint32_t w[4];
  philox4x32_10(i0, i1, i2, tag, (uint32_t)seed, (uint32_t)(seed >> 32), w);
  const double u1 = ((double)(w[0] >> 8) + 0.5) * 5.9604644775390625e-08;
  const double u2 = (double)(w[1] >> 8) * 5.9604644775390625e-08;
  const double r = sqrt(-2.0 * log(u1)), th = 6.283185307179586476925286766559 * u2;
  z0 = r * cos(th);
  z1 = r * sin(th);
}

// The activations of utils.activation_fn (reference utils.py:42-59) other than swish, which every kernel keeps as its own
// first branch (`act == UDA_ACT_SWISH`): relu = max(x, 0), relu6 = min(max(x, 0), 6), hswish = x * relu6(x + 3) / 6.
// One clamp with two wave-uniform constants (+ a multiply for hswish; the division by six is a multiplication by its
// float32 reciprocal, 1 ulp like the hardware rcp / exp2 of the swish path): four vector instructions and no live register
// beyond the value, so the epilogues of the tuned kernels keep their register counts (tools/codeobj.py resources).
__device__ __forceinline__ float act_relu_family(float v, int act) {
  if (act == UDA_ACT_MISH) {
    // x tanh(softplus(x)) = x n / (n + 2) with n = e^x (e^x + 2): one exp, one rcp (the hardware ones, as in the swish path);
    // beyond x = 20 the quotient is 1 to float32 and e^2x would overflow
    const float e = __expf(fminf(v, 20.0f));
    const float n = e * (e + 2.0f);
    return v * (n * __builtin_amdgcn_rcpf(n + 2.0f));
  }
  const float off = act == UDA_ACT_HSWISH ? 3.0f : 0.0f;
  const float hi = act == UDA_ACT_RELU ? __builtin_inff() : 6.0f;
  const float r = __builtin_amdgcn_fmed3f(v + off, 0.0f, hi);
  return act == UDA_ACT_HSWISH ? (v * r) * 0.16666667163372040f : r;
}
#endif

// ---------------------------------------------------------------- split-precision schemes of the 1x1 contractions
// (`wparts` of the argument blocks below = PARTS of the kernels; see mfma_common.h)
enum {
  UDA_SPLIT_NONE = 0,     // exact f32-input MFMA kernels (kernels_conv.hip)
  UDA_SPLIT_BF16X2 = 2,   // two bf16 pieces per operand, three cross terms (~2^-17 per product)
  UDA_SPLIT_BF16X3 = 3,   // three bf16 pieces, six cross terms (~2^-24)
  UDA_SPLIT_F16X2 = 4,    // two fp16 pieces, three cross terms (~2^-22; operands must stay below 65504)
};
inline int uda_split_pieces(int scheme) { return scheme == UDA_SPLIT_BF16X3 ? 3 : 2; }

// ---------------------------------------------------------------- kernel argument blocks
struct PreGeo;
struct StemArgs {
  const float* in;    // [rows, H, W, 3]
  float* out;         // [rows, Ho, Wo, Co]
  const float* w;     // [27, Co]
  const float* bn_scale;
  const float* bn_shift;
  int H, W, Ho, Wo, Co;
  int pad_t, pad_l;
  int rows;
  // uint8 input (every image of the batch at scale 1: nothing to resample): the stem normalises on the fly through a 768-entry
  // table of ((float)v - mean[c]) / std[c] - the preprocess kernel's own expression, so the convolution sees the same values bit
  // for bit - and reads a quarter of the bytes; the separate preprocess pass and its float32 image are skipped
  const uint8_t* u8;  // packed raw images (image i at u8 + geo[i].off, [geo[i].h, geo[i].w, 3]) or null
  const PreGeo* geo;  // [images] (device)
  int img0;           // first image of this launch in geo
  float mean[3], stdv[3];
  int act;            // uda_act (the specialised stems are swish kernels: any other activation runs stem_kernel)
};

struct PwArgs {
  const float* in;       // [rows_in, HW, Cin]
  float* out;            // [rows, HW, Cout]
  const float* w;        // [Cin, Cout]
  const float* bias;     // [Cout] or null
  const float* bn_scale; // [Cout] or null
  const float* bn_shift;
  const float* se;       // [rows / se_div, Cin] gate on the input or null
  const float* mask;     // [rows, Cout] dropout keep-scale or null
  const float* res;      // [rows, HW, Cout] residual or null
  int HW, Cin, Cout;
  int in_div;            // rows_in = rows / in_div (input shared by the MC samples of an image)
  int res_div;
  int se_div;            // 1: one gate per output row (per sample); in_div: one per input row
  int act;
  const void* wsplit;    // split weights in MFMA fragment order (kernels_pwb.hip) or null
  int wparts;            // split scheme of wsplit (UDA_SPLIT_*)
  float wunscale;        // the packed weights are the kernel times 1 / wunscale (a power of two; 1 unless fp16 pieces)
  unsigned* oor;         // fp16 pieces: word that receives bit 0 when an operand above 65504 was split (or null)
};
void launch_pwb(const PwArgs& a, int rows, hipStream_t s);
size_t pwb_packed_elems(int K, int N, int scheme);
// scale: every weight is multiplied by it before the split (a power of two; fp16 pieces only)
void pwb_pack_weights(const float* w, int K, int N, int scheme, uint16_t* out, float scale = 1.0f);
// power of two that brings the largest magnitude of w[0 .. n) into [2^13, 2^14) (1 for an all-zero tensor)
float split_weight_scale(const float* w, size_t n);

struct SepArgs {          // fused depthwise 3x3 (stride 1, SAME) + 1x1 (kernels_pwb.hip)
  const float* in;        // [rows / in_div, H, W, C]
  float* out;             // [rows, H, W, Cout]
  const float* wd;        // depthwise kernel [9, C]
  const void* wsplit;     // 1x1 kernel [C, Cout] as split-bf16 fragments
  const float* bias;      // [Cout] or null
  const float* bn_scale;  // [Cout] or null
  const float* bn_shift;
  const float* mask;      // [rows, Cout] dropout keep-scale or null
  const float* mask_in;   // [rows, C] DEFERRED keep-scale of the input channels (the producer left its dropout site to this op:
                          // input shared by the in_div samples of an image; sep_kernel's TIN mode) or null
  int H, W, C, Cout;
  int in_div;
  int act;
  int wparts;             // split scheme (UDA_SPLIT_*)
  float wunscale;         // see PwArgs
  unsigned* oor;
};
// rows = sample rows of the output; with mask_in the grid runs over rows / in_div images and each block serves the in_div samples
void launch_sep(const SepArgs& a, int rows, hipStream_t s);
bool sep_tin_supported(int C, int Cout, int scheme);      // the deferred-input mode exists for this shape (plan.py mirrors it)
// Several independent separable convs of ONE shape class (same C, Cout, activation, sample-axis relation, row count) in one
// launch: the five pyramid levels of a head layer.  The grid covers the tiles of all problems; a block finds its problem
// from the tile prefix sums.  Everything that differs between the problems lives in SepLevel.
constexpr int UDA_SEP_MAX_LV = 8;
struct SepLevel {
  const float* in; float* out; const float* wd; const void* wsplit;
  const float* bias; const float* bn_scale; const float* bn_shift; const float* mask; const float* mask_in;
  int H, W;
  float wunscale;
};
struct SepMulti {
  SepArgs one;                         // the single problem (n_lv == 0) / the fields shared by all problems
  int n_lv;
  int tiles, rows;                     // tiles of one sample row (all problems), sample rows: filled by the launcher
  int tile0[UDA_SEP_MAX_LV + 1];       // first tile of every problem, total in [n_lv]
  SepLevel lv[UDA_SEP_MAX_LV];
};
void launch_sep_multi(const SepArgs& common, const SepLevel* lv, int n_lv, int rows, hipStream_t s);
bool sep_supported(int C, int Cout);
size_t sep_lds_bytes(int C, int Cout, int scheme);      // dynamic LDS of the launch an op of this shape gets
struct FuseArgs;
// The same conv on an LDS-staged 16 x 16 tile (kernels_sep.hip: sepf_kernel); with `fused` the input is the BiFPN fusion
// described there, computed on the fly (a.in unused).  Outputs are bit-identical to launch_sep's.
void launch_sepf(const SepArgs& a, const FuseArgs* fused, int rows, hipStream_t s);
bool sepf_supported(int C, int Cout, int scheme);
size_t sepf_lds_bytes(int C, int Cout, int scheme);

struct DwArgs {
  const float* in;       // [rows_in, H, W, C]
  float* out;            // [rows, Ho, Wo, C]
  const float* w;        // [k*k, C]
  const float* bn_scale; // or null
  const float* bn_shift;
  const float* mask;     // [rows, C] or null
  float* se_partial;     // [rows, n_tiles, C] or null
  int H, W, Ho, Wo, C;
  int pad_t, pad_l;
  int in_div;
  int act;
  int tc;                // threads along channel quads
  int pxb;               // x-groups per block
  int n_cchunk;          // channel chunks (grid.z = rows * n_cchunk)
  int n_tiles;           // Ho * gridDim.x
};

struct MbxArgs {
  const float* in;        // [rows_in, H, W, Cin]
  float* out;             // [rows, Ho, Wo, Cmid]
  const float* we;        // expand kernel [Cin, Cmid]
  const float* sc0;       // BN after expand
  const float* sh0;
  const float* mask0;     // [rows, Cmid] or null
  const float* wd;        // depthwise kernel [k*k, Cmid]
  const float* sc1;       // BN after depthwise
  const float* sh1;
  const float* mask1;     // [rows, Cmid] or null
  float* se_partial;      // [rows, n_tiles, Cmid] or null
  int H, W, Ho, Wo, Cin, Cmid;
  int pad_t, pad_l;
  int in_div;
  int n_tiles;
  unsigned long long* stamps;  // diagnostic phase stamps (UDA_MBX_STAMPS); null in production
  const void* wsplit;     // expand kernel * BN scale (+ BN shift row) as split fragments (kernels_pwb.hip) or null
  int wparts;             // split scheme of wsplit (UDA_SPLIT_*)
  unsigned* oor;          // fp16 pieces: out-of-range flag word (see PwArgs) or null
  const float* wpar;      // per-slab depthwise taps + BN scale / shift block (mbx_pack_params) or null
  // fused projection of the previous block (mbxb_kernel FUSE0): in = D [rows / in_div, H, W, c0]
  const float* gate;      // [rows / g_div, c0] per-sample gate on D (SE gate x deferred dropout), null = not fused
  const float* w0t;       // [32][32] projection kernel^T x BN scale (mbxb_pack_proj)
  const float* sh0f;      // [32] projection BN shift
  const uint4* w0frag;    // [rows / g_div][2 k-steps][wparts][64 lanes]: W0^T x gate, split, in A-fragment order (launch_w0gate) or null
  int c0, g_div;
  // block order of launches whose input is shared by the in_div samples of an image (mbxb_kernel): 1-D grid, the
  // in_div blocks of one tile adjacent and on ONE XCD (ids = tile slot + 8 t), so that the shared tile is fetched into
  // that XCD's L2 once instead of once per sample; 0 = plain (tiles_x, tiles_y, rows) grid
  int remap_T, tiles_x, tiles_y;
  // deep variants (mbxd / mbxp) on small grids (a few sample rows): the 32-channel slabs of a tile are divided among
  // ch_groups blocks (grid z = rows x ch_groups), so that a launch of fewer blocks than the device holds still fills it
  int ch_groups;
};
void mbxb_pack_proj(const float* w0, const float* sc, const float* sh, int c0, int cout, float* out);
size_t mbx_par_floats(int Cmid, int k);
void mbx_pack_params(const float* wd, const float* sc1, const float* sh1, int Cmid, int k, float* out);
void launch_mbxb(const MbxArgs& a, int rows, int k, int stride, hipStream_t s);
void launch_mbxd(const MbxArgs& a, int rows, int k, int stride, hipStream_t s);     // deep blocks (Cin > 48), stride 1 or 2
bool mbxd_supported(int Cin, int Cmid, int k, int stride);
size_t mbx_lds_bytes(int Cin, int Cmid, int k, int stride, int scheme, int Ho, int Wo);   // dynamic LDS of the fused launch of this op
int mbxd_tiles(int Ho, int Wo, int k, int stride = 1);
bool mbxd_wide(int Ho, int Wo, int k, int stride);     // 20-column tiles for this map (mirror: plan.mbx_tile)
bool mbxb_supported(int Cin, int Cmid, int k, int stride);
int mbxb_tiles(int Ho, int Wo, int k, int stride);
size_t mbxb_packed_elems(int Cin, int Cmid, int scheme = UDA_SPLIT_BF16X2);
size_t mbxb_w0frag_elems(int gate_rows, int scheme);
void launch_w0gate(const float* gate, const float* w0t, int c0, int gate_rows, int scheme, uint4* out, unsigned* oor, hipStream_t s);
// stats (optional, 2 floats): largest magnitude and rms of the matrix that was packed (expand kernel x BN scale, shift row)
void mbxb_pack_weights(const float* we, const float* sc0, const float* sh0, int Cin, int Cmid, uint16_t* out, bool perm16 = false,
                       int scheme = UDA_SPLIT_BF16X2, float* stats = nullptr);
void launch_mbx(const MbxArgs& a, int rows, int k, int stride, hipStream_t s);
int mbx_tiles(int Ho, int Wo, int k, int stride);
bool mbx_supported(int Cin, int Cmid, int k, int stride);

struct SeArgs {
  const float* partial;  // [rows, n_tiles, C]
  float* scale;          // [rows, C]
  const float* w1;       // [C, mid]
  const float* b1;       // [mid]
  const float* w2;       // [mid, C]
  const float* b2;       // [C]
  int C, mid, n_tiles;
  float inv_hw;
  const float* mask;     // [rows, C] deferred dropout keep-scale of the squeezed tensor, or null
  int in_div;            // partial sums are per input row: rows / in_div
  int act;               // uda_act of the reduce layer (relu_fn, efficientnet_model.py:225)
};

struct FuseArgs {
  const float* in[UDA_MAX_FUSE_INPUTS];
  float* out;
  float wgt[UDA_MAX_FUSE_INPUTS];
  int mode[UDA_MAX_FUSE_INPUTS];   // uda_resample
  int Hi[UDA_MAX_FUSE_INPUTS], Wi[UDA_MAX_FUSE_INPUTS];
  int in_div[UDA_MAX_FUSE_INPUTS];
  float sy[UDA_MAX_FUSE_INPUTS], sx[UDA_MAX_FUSE_INPUTS];       // nearest: in/out
  int pk[UDA_MAX_FUSE_INPUTS], ps[UDA_MAX_FUSE_INPUTS];         // pool size / stride
  int ppt[UDA_MAX_FUSE_INPUTS], ppl[UDA_MAX_FUSE_INPUTS];       // pool pad before
  int n_in;
  int H, W, C;
  int act;
  int64_t total;  // rows*H*W*C/4
};

// ---------------------------------------------------------------- launchers (kernels_conv.hip)
void launch_stem(const StemArgs& a, hipStream_t s);
bool stem_u8_supported(int Co);      // the uint8-input variant of the stem exists for this width
void launch_pw(const PwArgs& a, int rows, hipStream_t s);
void launch_dw(DwArgs a, int rows, int k, int stride, hipStream_t s);
void dw_geometry(int C, int Wo, int k, int stride, int* tc, int* pxb, int* n_cchunk, int* grid_x, int* xb);
int dw_tiles(int C, int Ho, int Wo, int k, int stride);  // SE tile sums one depthwise launch leaves per sample row
void launch_se(const SeArgs& a, int rows, hipStream_t s);
void launch_fuse(const FuseArgs& a, hipStream_t s);
void launch_philox_masks(float* masks, const int64_t* site_off_dev, const int32_t* site_ch_dev,
                         const float* site_rate_dev, int n_sites, int rows, uint32_t row_base, int max_c4,
                         uint64_t seed, int t_local, int t_total, int t_first, int t_stride, hipStream_t s);

// ---------------------------------------------------------------- post-process (kernels_post.hip)
struct LevelTable {
  int num_levels;
  int hw[UDA_MAX_LEVELS];
  int a_off[UDA_MAX_LEVELS + 1];   // anchor offset of each level; a_off[num_levels] = A_tot
  const float* cls[UDA_MAX_LEVELS]; // [n*Tc, hw, A*C]
  const float* box[UDA_MAX_LEVELS]; // [n*Tb, hw, boxch]
};

struct AggArgs {
  LevelTable lv;
  const float* anchors;  // [A_tot, 4]
  int n_img, A_tot, A, C;
  int K;                 // candidates per image: A_tot (argmax path) or max_nms_inputs (top-k path)
  const int32_t* cand_flat;  // [n, K] flat (anchor*C + class) indices of the top-k path, or null
  int Tc, Tb;            // samples carried by the class / box head outputs (1 = not stacked)
  int loss_att;
  int decode;            // uda_decode
  float* boxes;          // [n, A_tot, 4]
  float* scores;         // [n, A_tot]
  int32_t* classes;      // [n, A_tot]
  float* logits;         // [n, A_tot, C]  mean logits
  float* u_cls;          // [n, K, C] (argmax path) / [n, K, 1] (top-k path) or null
  float* u_al;           // [n, A_tot, 4] or null
  float* u_ep;           // [n, A_tot, 4] or null
  int park_all, cls_slots;   // set by launch_aggregate: LDS parking layout of the class logits
  int decode_nsamples;       // UDA_DECODE_SAMPLE: draws per (candidate, MC sample)
  uint64_t decode_seed;      //   seed of the Philox normal stream
  uint32_t row_base;         //   global sample-row index of image 0 of this range: (image offset + i0) * Tb
};
void launch_aggregate(const AggArgs& a, hipStream_t s);
// mean logits of every (anchor, class): out [n, A_tot*C]   (input of the top-k pre-selection)
void launch_class_mean(const AggArgs& a, float* out, hipStream_t s);
// per image: the k largest of vals[n][0..L) -> flat indices, value descending, ties -> lower index
// ws: scratch of topk_workspace_bytes(n_img, k) bytes for the multi-block selection (null: one block per image)
void launch_topk(const float* vals, int n_img, int L, int k, int32_t* out_idx, void* ws, hipStream_t s);
size_t topk_workspace_bytes(int n_img, int k);

struct PreGeo {          // one raw image of a batch (raw sizes may differ between images: KITTI has four)
  unsigned long long off;     // byte offset of the image in the packed uint8 buffer
  int h, w;                   // raw size
  int sh, sw;                 // scaled size inside the H x W network input (the rest is zero padding)
  float scale_y, scale_x;     // raw / scaled ratios of the bilinear sampler
  float inv_scale;            // image_scale handed to the post-process
  int pad_;
};
struct PreprocArgs {
  const uint8_t* in;   // images back to back, image i = [geo[i].h, geo[i].w, 3] at geo[i].off
  float* out;          // [n, H, W, 3]
  const PreGeo* geo;   // [n] (device)
  int n, H, W;
  float mean[3], stdv[3];
};
void launch_preprocess(const PreprocArgs& a, hipStream_t s);

struct NmsArgs {
  const float* boxes;    // [n, K, 4]
  float* stale;          // [n, K]  working scores (dead = -inf)
  int32_t* begin;        // [n, K]
  float* tent;           // [n, K]  updated score computed in epoch ev[i]
  float* ub;             // [n, K]  upper bound of the candidate's updated score in every later epoch
  int32_t* ev;           // [n, K]  epoch of the last evaluation (-1 = never)
  int32_t* sel_idx;      // [n, M]
  float* sel_score;      // [n, M]
  float* sel_box;        // [n, M, 4]
  unsigned long long* bound_key;  // [n, M]
  unsigned long long* win_key;    // [n, M]
  int32_t* nsel;         // [n]
  int32_t* done;         // [n]
  int n_img, K, M;       // n_img = number of NMS problems (images, or images*classes in per-class mode)
  float iou_thr, score_thr, scale;  // scale = soft ? -0.5/sigma : 0
  int soft;
  int segs;              // problems per image: 1 (global) or num_classes (per-class mode)
  const int32_t* classes;  // [images, K] candidate classes (per-class mode) or null
};
void launch_nms_init(const NmsArgs& a, const float* scores, hipStream_t s);
void launch_nms_epoch(const NmsArgs& a, int epoch, hipStream_t s);
void launch_nms_finish(const NmsArgs& a, int pad, hipStream_t s);
// all epochs in one launch, one block per problem (kernels_post.hip "NMS, one launch")
bool nms_solo_supported(const NmsArgs& a);
void launch_nms_solo(const NmsArgs& a, const float* scores, hipStream_t s);
// the same with the per-candidate state in registers, problems of up to 8192 candidates
bool nms_reg_supported(const NmsArgs& a);
void launch_nms_reg(const NmsArgs& a, const float* scores, hipStream_t s);

// all epochs of large problems in one launch of a co-resident grid; slots [n_img x nms_coop_slot_words(M)] / err [1] are scratch
size_t nms_coop_slot_words(int M);
int launch_nms_coop(const NmsArgs& a, const float* scores, unsigned long long* slots, int* err, hipStream_t s);   // 1 launched, 0 not its domain, -1 wanted but not launched

// NMS on the top-scoring prefix of a large candidate set (kernels_post.hip "NMS on a score prefix")
struct PrefixArgs {
  const float* scores;   // [n, K]
  const float* boxes;    // [n, K, 4]
  int32_t* sub_idx;      // [n, Lcap]  candidate index of every prefix entry, ascending
  float* sub_scores;     // [n, Lcap]  (-inf padding)
  float* sub_boxes;      // [n, Lcap, 4]
  uint32_t* excl_key;    // [n]  order-preserving key of the largest excluded score (0 = nothing excluded)
  int32_t* bad;          // [n]  1 = the prefix cannot stand in for the full problem
  int n_img, K, Lp, Lcap;
};
void launch_prefix_select(const PrefixArgs& a, hipStream_t s);
struct PrefixCheckArgs {
  const int32_t* sub_sel_idx;   // [n, M] positions in the prefix
  const float* sub_sel_score;   // [n, M]
  const int32_t* sub_nsel;      // [n]
  const int32_t* sub_idx;       // [n, Lcap]
  const uint32_t* excl_key;     // [n]
  int32_t* sel_idx;             // [n, M] candidate indices (what the gather stage reads)
  float* sel_score;             // [n, M]
  int32_t* nsel;                // [n]
  int32_t* bad;                 // [n]
  int n_img, M, Lcap;
  float score_thr;
};
void launch_prefix_check(const PrefixCheckArgs& a, hipStream_t s);

struct GatherArgs {
  const int32_t* sel_idx;   // [n, M]
  const float* sel_score;   // [n, M]
  const int32_t* nsel;      // [n]
  const float* boxes;       // [n, K, 4]
  const int32_t* classes;   // [n, K]
  const float* logits;      // [n, K, C]
  const float* u_cls;       // or null
  const float* u_al;
  const float* u_ep;
  const float* scales;      // [n] image scales or null
  float* out_boxes;         // [n, M, box_cols]
  float* out_scores;        // [n, M]
  float* out_classes;       // [n, M, cls_cols]
  int32_t* out_valid;       // [n]
  float* out_logits;        // [n, M, C] or null
  int n_img, K, M, C, box_cols, cls_cols;
  int ucls_cols;            // class-std values per candidate: C (argmax path) or 1 (top-k path)
  float clip_h, clip_w;
  int clip;
};
void launch_gather(const GatherArgs& a, hipStream_t s);

struct MergeArgs {         // per-class mode: concat per-class selections, pad, top-M by score
  const int32_t* sel_idx;  // [images*C, M]
  const float* sel_score;  // [images*C, M]
  const int32_t* nsel;     // [images*C]
  const float* boxes;      // [images, K, 4]
  const float* scales;     // [images] or null
  unsigned long long* keys;  // workspace [images, C*M + M]
  float* out_boxes;        // [images, M, 4]
  float* out_scores;       // [images, M]
  float* out_classes;      // [images, M]
  int32_t* out_valid;      // [images]
  int n_img, K, M, C;
};
void launch_merge_per_class(const MergeArgs& a, hipStream_t s);
template <typename T>
struct NmsNpArgs {         // numpy NMS family (kernels_post.hip): problems = consecutive slices of dets
  const T* dets;           // [total, 5] x1 y1 x2 y2 score (sorted by score, descending, for hard / diou)
  const int32_t* off;      // [problems + 1]
  T* score;                // [total] workspace
  int32_t* state;          // [total] workspace
  T* out;                  // [total, 5] kept rows of each problem at its offset
  int32_t* n_out;          // [problems]
  int method;              // 0 hard, 1 diou, 2 gaussian, 3 linear
  T iou_thr, sigma, score_thr;
};
template <typename T>
void launch_nmsnp(const NmsNpArgs<T>& a, int n_problems, hipStream_t s);

struct CalibArgs {
  const float* boxes;     // [rows, box_cols] output boxes (+ uncertainty columns)
  const float* classes;   // [rows, cls_cols] class id in column 0
  float* out;             // [rows, 4]
  const double* xs;       // concatenated table thresholds
  const double* ys;
  const int32_t* tab_off; // [n_tables + 1]
  float temps[4];
  int rows, box_cols, cls_cols, col0;
  int mode, relative, n_tables;
};
void launch_calib(const CalibArgs& a, hipStream_t s);
void launch_probs(const float* logits, float* probs, float* entropy, int rows, int C, hipStream_t s);
struct PackDetArgs {
  const float *boxes, *scores, *classes, *logits;   // post-process outputs [n, M, bc] / [n, M] / [n, M, cc] / [n, M, C] (C = 0: no logits)
  const int32_t* valid;                             // [n]
  float* out;                                       // [rows_out, M, cols], cols = bc + 1 + cc + C + 1
  int n, rows_out, M, bc, cc, C, cols;
};
void launch_pack_det(const PackDetArgs& a, hipStream_t s);

struct ClsCalibArgs {
  const float* logits;    // [rows, C] mean logits of the selected rows
  const float* classes;   // [rows, cls_cols]: column 0 class id, columns 1..C the MC std of the logits (draws > 0)
  float* probs;           // [rows, C]
  float* entropy;         // [rows]
  float* uncert;          // [rows, C] std of the calibrated probabilities over the draws, or null
  const double* xs;       // isotonic tables (concatenated thresholds), 1 (all) or C (per class)
  const double* ys;
  const int32_t* tab_off; // [n_tables + 1]
  const float* temps;     // [C] temperature divisors (device)
  int rows, C, cls_cols;
  int mode;               // uda_class_calib_mode
  int draws;              // 0: calibrate the mean logits; > 0: that many normal draws per logit (the reference uses 10)
  uint64_t seed;
};
void launch_class_calib(const ClsCalibArgs& a, hipStream_t s);

}  // namespace uda
