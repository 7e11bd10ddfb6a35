// Pre/post-processing kernels (gfx950).  Compiled with -ffp-contract=off: every float
// operation here is an individually rounded IEEE op in the order written, so results are
// comparable bit for bit with the CPU oracle's statement of the same arithmetic.
//
//   preprocess_kernel  normalise + bilinear resize (half-pixel centres) + zero pad
//                      (reference dataloader.py:69-75,123-152; efficientdet_keras.py:1076-1100)
//   aggregate_kernel   MC mean / population std of the class logits, argmax + sigmoid,
//                      per-sample anchor decode (plain f32 or variance-propagating f64),
//                      mean / std over samples of the decoded corners, mean of decoded sigma
//                      (utils_extra.py:220-244; postprocess.py:75-141,284-331; anchors.py:41-75;
//                       utils_box.py:105-276)
//   nms_*              tf.raw_ops.NonMaxSuppressionV5 as an epoch-synchronous data-parallel
//                      algorithm (postprocess.py:342-420; see DESIGN.md "NMS")
//   gather_kernel      gathers by selected index, clips, rescales, packs the output tuple
//                      (postprocess.py:402-413,599-621)
//
// Pinned-down numerics shared with the oracle (oracle/post_ref.py header):
//   exp32(x) = float(exp(double(x))), sigmoid(x) = float(1/(1+exp(-double(x)))),
//   MC reductions are sequential float32 sums over t = 0..T-1.
#include <type_traits>
#include <hip/hip_fp16.h>

#include "uda_internal.h"

namespace uda {

// ------------------------------------------------------------------------------------ preprocess
__device__ __forceinline__ float norm_px(const uint8_t* p, int c, const PreprocArgs& a) {
  return ((float)p[c] - a.mean[c]) / a.stdv[c];
}

__global__ __launch_bounds__(256) void preprocess_kernel(PreprocArgs a) {
  const int64_t total = (int64_t)a.n * a.H * a.W;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  const int x = (int)(gid % a.W);
  const int y = (int)((gid / a.W) % a.H);
  const int n = (int)(gid / ((int64_t)a.W * a.H));
  float* o = a.out + (size_t)gid * 3;
  const PreGeo g = a.geo[n];                    // (uniform per image: one 40-byte scalar load for most waves)
  if (y >= g.sh || x >= g.sw) {
    o[0] = 0.f; o[1] = 0.f; o[2] = 0.f;
    return;
  }
  const uint8_t* img = a.in + g.off;
  if (g.sh == g.h && g.sw == g.w) {
    const uint8_t* p = img + ((size_t)y * g.w + x) * 3;
    for (int c = 0; c < 3; ++c) o[c] = norm_px(p, c, a);
    return;
  }
  const float fy = ((float)y + 0.5f) * g.scale_y - 0.5f;
  const float fx = ((float)x + 0.5f) * g.scale_x - 0.5f;
  const float fly = floorf(fy), flx = floorf(fx);
  const int ylo = (int)fmaxf(fly, 0.f), yhi = min((int)ceilf(fy), g.h - 1);
  const int xlo = (int)fmaxf(flx, 0.f), xhi = min((int)ceilf(fx), g.w - 1);
  const float ly = fy - fly, lx = fx - flx;
  const uint8_t* ptl = img + ((size_t)ylo * g.w + xlo) * 3;
  const uint8_t* ptr = img + ((size_t)ylo * g.w + xhi) * 3;
  const uint8_t* pbl = img + ((size_t)yhi * g.w + xlo) * 3;
  const uint8_t* pbr = img + ((size_t)yhi * g.w + xhi) * 3;
  for (int c = 0; c < 3; ++c) {
    const float tl = norm_px(ptl, c, a), tr = norm_px(ptr, c, a);
    const float bl = norm_px(pbl, c, a), br = norm_px(pbr, c, a);
    const float top = tl + (tr - tl) * lx;
    const float bot = bl + (br - bl) * lx;
    o[c] = top + (bot - top) * ly;
  }
}

void launch_preprocess(const PreprocArgs& a, hipStream_t s) {
  const int64_t total = (int64_t)a.n * a.H * a.W;
  hipLaunchKernelGGL(preprocess_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
}

// ------------------------------------------------------------------------------------ aggregate + decode
__device__ __forceinline__ float exp32(float x) { return (float)exp((double)x); }

struct Dec {
  float box[4];
  float sig[4];
};

__device__ __forceinline__ void decode_plain(const float* t, const float* an, Dec& d) {
  const float ya = (an[0] + an[2]) / 2.0f, xa = (an[1] + an[3]) / 2.0f;
  const float ha = an[2] - an[0], wa = an[3] - an[1];
  const float w = exp32(t[3]) * wa;
  const float h = exp32(t[2]) * ha;
  const float yc = t[0] * ha + ya;
  const float xc = t[1] * wa + xa;
  d.box[0] = yc - h / 2.0f;
  d.box[1] = xc - w / 2.0f;
  d.box[2] = yc + h / 2.0f;
  d.box[3] = xc + w / 2.0f;
}

// method "sample" (utils_box.py:162-184): S draws from Normal(loc = (ty, tx, th, tw), scale = sqrt(sigma^2)), each decoded like
// a plain box, mean and population variance (tf.nn.moments) of the four corners over the draws, all in float64.  TFP's
// stream cannot be reproduced: the draws come from the build's Philox normal stream (counter (anchor, sample row, 2 s + {0, 1}),
// tag 0xD5: (y, x) from the first call, (h, w) from the second), shared with the oracle.  Welford's update keeps the variance
// accurate at box coordinates ~1e3 with sigma ~1e-1.
__device__ __forceinline__ void decode_sample(const float* t, const float* sg, const float* an, int S, uint64_t seed,
                                              uint32_t id0, uint32_t id1, Dec& d) {
  const double a0 = an[0], a1 = an[1], a2 = an[2], a3 = an[3];
  const double ya = (a0 + a2) / 2, xa = (a1 + a3) / 2;
  const double ha = a2 - a0, wa = a3 - a1;
  const double ty = t[0], tx = t[1], th = t[2], tw = t[3];
  double sc[4];
  for (int k = 0; k < 4; ++k) { const double s = sg[k]; sc[k] = sqrt(s * s); }
  double mean[4] = {0, 0, 0, 0}, m2[4] = {0, 0, 0, 0};
  for (int s = 0; s < S; ++s) {
    double zy, zx, zh, zw;
    philox_normal2(seed, id0, id1, (uint32_t)(2 * s), 0xD5u, zy, zx);
    philox_normal2(seed, id0, id1, (uint32_t)(2 * s + 1), 0xD5u, zh, zw);
    const double sy = ty + sc[0] * zy, sx = tx + sc[1] * zx, sh = th + sc[2] * zh, sw = tw + sc[3] * zw;
    const double w = exp(sw) * wa, h = exp(sh) * ha;
    const double yc = sy * ha + ya, xc = sx * wa + xa;
    const double v[4] = {yc - h / 2.0, xc - w / 2.0, yc + h / 2.0, xc + w / 2.0};
    const double n = (double)(s + 1);
    for (int k = 0; k < 4; ++k) {
      const double dl = v[k] - mean[k];
      mean[k] += dl / n;
      m2[k] += dl * (v[k] - mean[k]);
    }
  }
  for (int k = 0; k < 4; ++k) {
    d.box[k] = (float)mean[k];
    d.sig[k] = (float)sqrt(m2[k] / (double)S);
  }
}

__device__ __forceinline__ void decode_uncert(const float* t, const float* sg, const float* an,
                                              int method, Dec& d) {
  const double a0 = an[0], a1 = an[1], a2 = an[2], a3 = an[3];
  const double ya = (a0 + a2) / 2, xa = (a1 + a3) / 2;
  const double ha = a2 - a0, wa = a3 - a1;
  const double ty = t[0], tx = t[1], th = t[2], tw = t[3];
  const double s0 = sg[0], s1 = sg[1], s2 = sg[2], s3 = sg[3];
  const double dty = s0 * s0, dtx = s1 * s1, dth = s2 * s2, dtw = s3 * s3;
  double w, h, yc, xc, dymin, dxmin, dymax, dxmax;
  if (method == UDA_DECODE_LNORM) {
    w = exp(tw + dtw / 2) * wa;
    h = exp(th + dth / 2) * ha;
    yc = ty * ha + ya;
    xc = tx * wa + xa;
    const double dw = (exp(dtw) - 1) * exp(2 * tw + dtw) * (wa * wa);
    const double dh = (exp(dth) - 1) * exp(2 * th + dth) * (ha * ha);
    const double dyc = dty * (ha * ha);
    const double dxc = dtx * (wa * wa);
    dymin = dyc + dh / 4.0;
    dxmin = dxc + dw / 4.0;
    dymax = dymin;
    dxmax = dxmin;
  } else {  // falsedec
    w = exp(tw) * wa;
    h = exp(th) * ha;
    yc = ty * ha + ya;
    xc = tx * wa + xa;
    const double dw = exp(dtw) * wa;
    const double dh = exp(dth) * ha;
    const double dyc = dty * ha + ya;
    const double dxc = dtx * wa + xa;
    dymin = fabs(dyc - dh / 2.0);
    dxmin = fabs(dxc - dw / 2.0);
    dymax = dyc + dh / 2.0;
    dxmax = dxc + dw / 2.0;
  }
  d.box[0] = (float)(yc - h / 2.0);
  d.box[1] = (float)(xc - w / 2.0);
  d.box[2] = (float)(yc + h / 2.0);
  d.box[3] = (float)(xc + w / 2.0);
  d.sig[0] = (float)sqrt(dymin);
  d.sig[1] = (float)sqrt(dxmin);
  d.sig[2] = (float)sqrt(dymax);
  d.sig[3] = (float)sqrt(dxmax);
}

// SAMPLE is a compile-time flag: the Philox / Box-Muller code of the "sample" method would otherwise cost the common
// kernels registers (measured: aggregate_reg_kernel 0.94 -> 1.25 ms with the branch inside).
template <bool SAMPLE = false>
__device__ __forceinline__ void decode_one(const AggArgs& a, const float* bp, int A, const float* an,
                                           Dec& d, uint32_t id0 = 0, uint32_t id1 = 0) {
  float t[4] = {bp[0], bp[1], bp[2], bp[3]};
  if (a.loss_att) {
    const float* sp = bp + 4 * A;
    float sg[4] = {sp[0], sp[1], sp[2], sp[3]};
    if constexpr (SAMPLE) decode_sample(t, sg, an, a.decode_nsamples, a.decode_seed, id0, id1, d);
    else decode_uncert(t, sg, an, a.decode, d);
  } else {
    decode_plain(t, an, d);
    d.sig[0] = d.sig[1] = d.sig[2] = d.sig[3] = 0.f;
  }
}

// One thread per candidate, 64-thread blocks.  Every head value is read from HBM exactly ONCE: the T class
// logits and the T decoded boxes of the candidate are parked in LDS ([slot][lane], conflict-free) between
// the mean pass and the deviation pass, so the two-pass population std keeps the oracle's arithmetic order
// (bit-exact) without a second trip to memory and without decoding twice.
constexpr int AGG_BLOCK = 64;

template <bool SAMPLE>
__global__ __launch_bounds__(AGG_BLOCK) void aggregate_kernel(AggArgs a) {
  extern __shared__ float park[];          // [Tc * C + Tb * 4][AGG_BLOCK]
  const int lane = threadIdx.x;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)a.n_img * a.K) return;
  const int n = (int)(gid / a.K);
  const int C = a.C;
  int ai = (int)(gid % a.K), fixed_c = -1;
  if (a.cand_flat) {            // top-k path: this candidate is one (anchor, class) pair
    const int flat = a.cand_flat[gid];
    ai = flat / C;
    fixed_c = flat % C;
  }
  int lvl = 0;
  while (lvl + 1 < a.lv.num_levels && ai >= a.lv.a_off[lvl + 1]) ++lvl;
  const int loc = ai - a.lv.a_off[lvl];
  const int p = loc / a.A, al = loc % a.A;
  const int hw = a.lv.hw[lvl];

  // ---- class logits: mean / population std over the T axis; first-max argmax or the given class
  const int cch = a.A * C;
  const float* cbase = a.lv.cls[lvl] + ((size_t)n * a.Tc * hw + p) * cch + al * C;
  const size_t cstride = (size_t)hw * cch;
  const float fT = (float)a.Tc;
  // park_all: slot (t * C + c), every logit parked up front (sample-major reads: the C logits of a sample
  // are contiguous); otherwise slot t, one class at a time (large T * C that would not fit LDS)
  float* pc = park + lane;
  const int cs = a.park_all ? C : 0;                        // slot of (t, c) = t * cs' + c' below
  if (a.Tc > 1 && a.park_all)
    for (int t = 0; t < a.Tc; ++t)
      for (int c = 0; c < C; ++c) pc[(t * C + c) * AGG_BLOCK] = cbase[t * cstride + c];
  float best = -INFINITY;
  int best_c = 0;
  for (int c = 0; c < C; ++c) {
    float m, sd = 0.f;
    if (a.Tc > 1) {
      const int c0 = a.park_all ? c : 0, ts = a.park_all ? cs : 1;
      if (!a.park_all)
        for (int t = 0; t < a.Tc; ++t) pc[t * AGG_BLOCK] = cbase[t * cstride + c];
      m = pc[c0 * AGG_BLOCK];
      for (int t = 1; t < a.Tc; ++t) m = m + pc[(t * ts + c0) * AGG_BLOCK];
      m = m / fT;
      if (a.u_cls && (fixed_c < 0 || fixed_c == c)) {
        float v = 0.f;
        for (int t = 0; t < a.Tc; ++t) {
          const float dlt = pc[(t * ts + c0) * AGG_BLOCK] - m;
          v = v + dlt * dlt;
        }
        sd = sqrtf(v / fT);
      }
    } else {
      m = cbase[c];
    }
    a.logits[(size_t)gid * C + c] = m;
    if (fixed_c < 0) {
      if (a.u_cls) a.u_cls[(size_t)gid * C + c] = sd;
      if (m > best) {
        best = m;
        best_c = c;
      }
    } else if (c == fixed_c) {
      if (a.u_cls) a.u_cls[gid] = sd;
      best = m;
      best_c = c;
    }
  }
  a.scores[gid] = (float)(1.0 / (1.0 + exp(-(double)best)));
  a.classes[gid] = best_c;

  // ---- boxes: per-sample decode, then mean / std over samples
  const int bch = a.A * (a.loss_att ? 8 : 4);
  const float* bbase = a.lv.box[lvl] + ((size_t)n * a.Tb * hw + p) * bch + al * 4;
  const size_t bstride = (size_t)hw * bch;
  const float an[4] = {a.anchors[ai * 4 + 0], a.anchors[ai * 4 + 1], a.anchors[ai * 4 + 2],
                       a.anchors[ai * 4 + 3]};
  Dec d;
  decode_one<SAMPLE>(a, bbase, a.A, an, d, (uint32_t)ai, a.row_base + (uint32_t)n * (uint32_t)a.Tb);
  if (a.Tb == 1) {
    for (int k = 0; k < 4; ++k) a.boxes[(size_t)gid * 4 + k] = d.box[k];
    if (a.u_al) for (int k = 0; k < 4; ++k) a.u_al[(size_t)gid * 4 + k] = d.sig[k];
    if (a.u_ep) for (int k = 0; k < 4; ++k) a.u_ep[(size_t)gid * 4 + k] = 0.f;
    return;
  }
  float* pb = park + (size_t)a.cls_slots * AGG_BLOCK + lane;   // slot (t * 4 + k)
  const float fTb = (float)a.Tb;
  float sb[4] = {d.box[0], d.box[1], d.box[2], d.box[3]};
  float ss[4] = {d.sig[0], d.sig[1], d.sig[2], d.sig[3]};
  for (int k = 0; k < 4; ++k) pb[k * AGG_BLOCK] = d.box[k];
  for (int t = 1; t < a.Tb; ++t) {
    decode_one<SAMPLE>(a, bbase + t * bstride, a.A, an, d, (uint32_t)ai, a.row_base + (uint32_t)n * (uint32_t)a.Tb + (uint32_t)t);
    for (int k = 0; k < 4; ++k) {
      sb[k] = sb[k] + d.box[k];
      ss[k] = ss[k] + d.sig[k];
      pb[(t * 4 + k) * AGG_BLOCK] = d.box[k];
    }
  }
  float mb[4];
  for (int k = 0; k < 4; ++k) {
    mb[k] = sb[k] / fTb;
    a.boxes[(size_t)gid * 4 + k] = mb[k];
    if (a.u_al) a.u_al[(size_t)gid * 4 + k] = ss[k] / fTb;
  }
  if (a.u_ep) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < a.Tb; ++t) {
      for (int k = 0; k < 4; ++k) {
        const float dlt = pb[(t * 4 + k) * AGG_BLOCK] - mb[k];
        v[k] = v[k] + dlt * dlt;
      }
    }
    for (int k = 0; k < 4; ++k) a.u_ep[(size_t)gid * 4 + k] = sqrtf(v[k] / fTb);
  }
}

// The same for the common small case (T samples of both heads, C classes, T * (C + 4) <= 110 values per candidate):
// the logits and decoded boxes of the candidate stay in REGISTERS between the mean pass and the deviation pass.  The
// LDS-parked version holds 28 KB per 64 threads (5 waves per CU: latency-bound, 2.3 ms); this one has no LDS, all
// T * C logit loads of a thread are in flight at once, and 16 waves fit a CU.  Same arithmetic order (bit-exact).
template <int T, int CT>     // CT = 0: any number of classes, one class at a time (T logits in registers per class)
__global__ __launch_bounds__(128) void aggregate_reg_kernel(AggArgs a) {
  const int C = CT ? CT : a.C;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)a.n_img * a.K) return;
  const int n = (int)(gid / a.K);
  int ai = (int)(gid % a.K), fixed_c = -1;
  if (a.cand_flat) {
    const int flat = a.cand_flat[gid];
    ai = flat / C;
    fixed_c = flat % C;
  }
  int lvl = 0;
  while (lvl + 1 < a.lv.num_levels && ai >= a.lv.a_off[lvl + 1]) ++lvl;
  const int loc = ai - a.lv.a_off[lvl];
  const int p = loc / a.A, al = loc % a.A;
  const int hw = a.lv.hw[lvl];
  const int cch = a.A * C;
  const float* cbase = a.lv.cls[lvl] + ((size_t)n * T * hw + p) * cch + al * C;
  const size_t cstride = (size_t)hw * cch;
  const float fT = (float)T;
  float best = -INFINITY;
  int best_c = 0;
  auto one_class = [&](int c, const float* x) {       // x[t] = logit of sample t
    float m = x[0], sd = 0.f;
#pragma unroll
    for (int t = 1; t < T; ++t) m = m + x[t];
    m = m / fT;
    if (a.u_cls && (fixed_c < 0 || fixed_c == c)) {
      float v = 0.f;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const float dlt = x[t] - m;
        v = v + dlt * dlt;
      }
      sd = sqrtf(v / fT);
    }
    a.logits[(size_t)gid * C + c] = m;
    if (fixed_c < 0) {
      if (a.u_cls) a.u_cls[(size_t)gid * C + c] = sd;
      if (m > best) {
        best = m;
        best_c = c;
      }
    } else if (c == fixed_c) {
      if (a.u_cls) a.u_cls[gid] = sd;
      best = m;
      best_c = c;
    }
  };
  if constexpr (CT > 0) {
    float cl[T * (CT ? CT : 1)];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int c = 0; c < CT; ++c) cl[t * CT + c] = cbase[t * cstride + c];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      float x[T];
#pragma unroll
      for (int t = 0; t < T; ++t) x[t] = cl[t * CT + c];
      one_class(c, x);
    }
  } else {
    for (int c = 0; c < C; ++c) {
      float x[T];
#pragma unroll
      for (int t = 0; t < T; ++t) x[t] = cbase[t * cstride + c];
      one_class(c, x);
    }
  }
  a.scores[gid] = (float)(1.0 / (1.0 + exp(-(double)best)));
  a.classes[gid] = best_c;

  const int bch = a.A * (a.loss_att ? 8 : 4);
  const float* bbase = a.lv.box[lvl] + ((size_t)n * T * hw + p) * bch + al * 4;
  const size_t bstride = (size_t)hw * bch;
  const float an[4] = {a.anchors[ai * 4 + 0], a.anchors[ai * 4 + 1], a.anchors[ai * 4 + 2], a.anchors[ai * 4 + 3]};
  float bxs[T * 4];
  float sb[4], ss[4];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    Dec d;
    decode_one<false>(a, bbase + t * bstride, a.A, an, d);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      sb[k] = t ? sb[k] + d.box[k] : d.box[k];
      ss[k] = t ? ss[k] + d.sig[k] : d.sig[k];
      bxs[t * 4 + k] = d.box[k];
    }
  }
  const float fTb = (float)T;
  float mb[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    mb[k] = sb[k] / fTb;
    a.boxes[(size_t)gid * 4 + k] = mb[k];
    if (a.u_al) a.u_al[(size_t)gid * 4 + k] = ss[k] / fTb;
  }
  if (a.u_ep) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float dlt = bxs[t * 4 + k] - mb[k];
        v[k] = v[k] + dlt * dlt;
      }
#pragma unroll
    for (int k = 0; k < 4; ++k) a.u_ep[(size_t)gid * 4 + k] = sqrtf(v[k] / fTb);
  }
}

// The register kernel for the other common (T, C): the class statistics are computed in two STREAMING passes over the
// sample axis (sample-major: the C logits of an anchor are contiguous, so consecutive loads of a wave touch the same lines;
// the class-major loop of aggregate_reg_kernel<T, 0> re-fetched every line C times through a thrashing L1: 6.3 ms for
// T = 20 / C = 10 on 32 images) - pass 1 the sums, pass 2 the squared deviations - with 3 C registers instead of T C; the T
// decoded boxes stay in registers as before.  Same arithmetic order (bit-exact).
template <int T, int C>
__global__ __launch_bounds__(128) void aggregate_reg2_kernel(AggArgs a) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)a.n_img * a.K) return;
  const int n = (int)(gid / a.K);
  int ai = (int)(gid % a.K), fixed_c = -1;
  if (a.cand_flat) {
    const int flat = a.cand_flat[gid];
    ai = flat / C;
    fixed_c = flat % C;
  }
  int lvl = 0;
  while (lvl + 1 < a.lv.num_levels && ai >= a.lv.a_off[lvl + 1]) ++lvl;
  const int loc = ai - a.lv.a_off[lvl];
  const int p = loc / a.A, al = loc % a.A;
  const int hw = a.lv.hw[lvl];
  const int cch = a.A * C;
  const float* cbase = a.lv.cls[lvl] + ((size_t)n * T * hw + p) * cch + al * C;
  const size_t cstride = (size_t)hw * cch;
  const float fT = (float)T;
  float m[C], v[C];
#pragma unroll
  for (int c = 0; c < C; ++c) m[c] = cbase[c];
  for (int t = 1; t < T; ++t) {
    const float* x = cbase + t * cstride;
#pragma unroll
    for (int c = 0; c < C; ++c) m[c] = m[c] + x[c];
  }
#pragma unroll
  for (int c = 0; c < C; ++c) { m[c] = m[c] / fT; v[c] = 0.f; }
  if (a.u_cls) {
    for (int t = 0; t < T; ++t) {
      const float* x = cbase + t * cstride;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float dlt = x[c] - m[c];
        v[c] = v[c] + dlt * dlt;
      }
    }
  }
  float best = -INFINITY;
  int best_c = 0;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float sd = a.u_cls ? sqrtf(v[c] / fT) : 0.f;
    a.logits[(size_t)gid * C + c] = m[c];
    if (fixed_c < 0) {
      if (a.u_cls) a.u_cls[(size_t)gid * C + c] = sd;
      if (m[c] > best) { best = m[c]; best_c = c; }
    } else if (c == fixed_c) {
      if (a.u_cls) a.u_cls[gid] = sd;
      best = m[c];
      best_c = c;
    }
  }
  a.scores[gid] = (float)(1.0 / (1.0 + exp(-(double)best)));
  a.classes[gid] = best_c;

  const int bch = a.A * (a.loss_att ? 8 : 4);
  const float* bbase = a.lv.box[lvl] + ((size_t)n * T * hw + p) * bch + al * 4;
  const size_t bstride = (size_t)hw * bch;
  const float an[4] = {a.anchors[ai * 4 + 0], a.anchors[ai * 4 + 1], a.anchors[ai * 4 + 2], a.anchors[ai * 4 + 3]};
  float bxs[T * 4];
  float sb[4], ss[4];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    Dec d;
    decode_one<false>(a, bbase + t * bstride, a.A, an, d);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      sb[k] = t ? sb[k] + d.box[k] : d.box[k];
      ss[k] = t ? ss[k] + d.sig[k] : d.sig[k];
      bxs[t * 4 + k] = d.box[k];
    }
  }
  float mb[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    mb[k] = sb[k] / fT;
    a.boxes[(size_t)gid * 4 + k] = mb[k];
    if (a.u_al) a.u_al[(size_t)gid * 4 + k] = ss[k] / fT;
  }
  if (a.u_ep) {
    float vv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float dlt = bxs[t * 4 + k] - mb[k];
        vv[k] = vv[k] + dlt * dlt;
      }
#pragma unroll
    for (int k = 0; k < 4; ++k) a.u_ep[(size_t)gid * 4 + k] = sqrtf(vv[k] / fT);
  }
}

void launch_aggregate(const AggArgs& a0, hipStream_t s) {
  AggArgs a = a0;
  static int regs = -1;
  if (regs < 0) { const char* e = getenv("UDA_AGG_REG"); regs = e ? atoi(e) : 1; }
  const bool sample = a.loss_att && a.decode == UDA_DECODE_SAMPLE;      // the (slow, optional) sampling decode: LDS-parking kernel only
  if (regs && !sample && a.Tc == a.Tb && (a.Tc == 10 || a.Tc == 20)) {
    const int64_t tot = (int64_t)a.n_img * a.K;
    const dim3 grid((unsigned)((tot + 127) / 128)), block(128);
    if (a.Tc == 10 && a.C == 7) hipLaunchKernelGGL((aggregate_reg_kernel<10, 7>), grid, block, 0, s, a);
    else if (a.Tc == 10 && a.C == 10) hipLaunchKernelGGL((aggregate_reg2_kernel<10, 10>), grid, block, 0, s, a);
    else if (a.Tc == 20 && a.C == 7) hipLaunchKernelGGL((aggregate_reg2_kernel<20, 7>), grid, block, 0, s, a);
    else if (a.Tc == 20 && a.C == 10) hipLaunchKernelGGL((aggregate_reg2_kernel<20, 10>), grid, block, 0, s, a);
    else if (a.Tc == 10) hipLaunchKernelGGL((aggregate_reg_kernel<10, 0>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((aggregate_reg_kernel<20, 0>), grid, block, 0, s, a);
    return;
  }
  if (regs && !sample && a.Tc == a.Tb && a.Tc == 30 && (a.C == 7 || a.C == 10)) {      // configs[4]: T = 30
    const int64_t tot = (int64_t)a.n_img * a.K;
    const dim3 grid((unsigned)((tot + 127) / 128)), block(128);
    if (a.C == 7) hipLaunchKernelGGL((aggregate_reg2_kernel<30, 7>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((aggregate_reg2_kernel<30, 10>), grid, block, 0, s, a);
    return;
  }
  const int64_t total = (int64_t)a.n_img * a.K;
  const int box_slots = a.Tb > 1 ? a.Tb * 4 : 0;
  a.park_all = (a.Tc * a.C + box_slots) * AGG_BLOCK * (int)sizeof(float) <= 48 * 1024;   // >= 3 blocks per CU
  a.cls_slots = a.Tc > 1 ? (a.park_all ? a.Tc * a.C : a.Tc) : 0;
  const size_t lds = (size_t)(a.cls_slots + box_slots) * AGG_BLOCK * sizeof(float);
  static size_t attr_lds[2] = {64 * 1024, 64 * 1024};
  if (lds > attr_lds[sample]) {
    if (sample) hipFuncSetAttribute((const void*)aggregate_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    else hipFuncSetAttribute((const void*)aggregate_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_lds[sample] = lds;
  }
  const dim3 grid((unsigned)((total + AGG_BLOCK - 1) / AGG_BLOCK));
  if (sample) hipLaunchKernelGGL(aggregate_kernel<true>, grid, dim3(AGG_BLOCK), lds, s, a);
  else hipLaunchKernelGGL(aggregate_kernel<false>, grid, dim3(AGG_BLOCK), lds, s, a);
}

// ------------------------------------------------------------------------------------ top-k pre-selection
// postprocess.topk_class_boxes with max_nms_inputs > 0 (postprocess.py:96-121): the k largest mean
// logits over anchors*classes.  TF leaves the order of sorted=False open; the build fixes it to
// value descending, ties -> lower flat index, identically in oracle and kernel.
__global__ __launch_bounds__(256) void class_mean_kernel(AggArgs a, float* out) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // (n, anchor, class)
  const int64_t per = (int64_t)a.A_tot * a.C;
  if (gid >= (int64_t)a.n_img * per) return;
  const int n = (int)(gid / per);
  const int r = (int)(gid % per);
  const int ai = r / a.C, c = r % a.C;
  int lvl = 0;
  while (lvl + 1 < a.lv.num_levels && ai >= a.lv.a_off[lvl + 1]) ++lvl;
  const int loc = ai - a.lv.a_off[lvl];
  const int p = loc / a.A, al = loc % a.A;
  const int hw = a.lv.hw[lvl], cch = a.A * a.C;
  const float* cbase = a.lv.cls[lvl] + ((size_t)n * a.Tc * hw + p) * cch + al * a.C + c;
  const size_t cstride = (size_t)hw * cch;
  float m = cbase[0];
  if (a.Tc > 1) {
    for (int t = 1; t < a.Tc; ++t) m = m + cbase[t * cstride];
    m = m / (float)a.Tc;
  }
  out[gid] = m;
}

void launch_class_mean(const AggArgs& a, float* out, hipStream_t s) {
  const int64_t total = (int64_t)a.n_img * a.A_tot * a.C;
  hipLaunchKernelGGL(class_mean_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a, out);
}

__device__ __forceinline__ uint32_t ord32(float v) {
  const uint32_t b = __float_as_uint(v);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

constexpr int TOPK_THREADS = 1024;
constexpr int TOPK_MAX = 8192;

// block-wide exclusive scan of one int per thread (1024 threads); *total gets the sum
__device__ __forceinline__ int block_excl_scan(int v, int* wsum /*[17]*/, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_up(incl, off, 64);
    if (lane >= off) incl += o;
  }
  __syncthreads();
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int w = 0; w < TOPK_THREADS / 64; ++w) {
      const int t = wsum[w];
      wsum[w] = run;
      run += t;
    }
    wsum[16] = run;
  }
  __syncthreads();
  *total = wsum[16];
  return wsum[wave] + incl - v;
}

// radix-select step: among elements whose key matches `prefix` on the bits above `shift+bits`,
// histogram the next `bits` bits; pick the bin (from the top) where the running count reaches `need`.
__device__ __forceinline__ void topk_select_digit(const float* v, int L, uint32_t prefix, int hi_shift, int shift,
                                                  int bits, unsigned* hist, int* need, uint32_t* out_prefix) {
  const int nb = 1 << bits;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    const uint32_t key = ord32(v[i]);
    if (hi_shift >= 32 || (key >> hi_shift) == prefix) atomicAdd(&hist[(key >> shift) & (nb - 1)], 1u);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int rem = *need;
    int b = nb - 1;
    for (; b > 0; --b) {
      if ((int)hist[b] >= rem) break;
      rem -= (int)hist[b];
    }
    *need = rem;                                   // still to take inside bin b
    *out_prefix = (hi_shift >= 32 ? 0u : (prefix << bits)) | (uint32_t)b;
  }
  __syncthreads();
}

__global__ __launch_bounds__(TOPK_THREADS) void topk_kernel(const float* vals, int L, int k, int32_t* out_idx) {
  __shared__ unsigned long long keys[TOPK_MAX];
  __shared__ unsigned hist[2048];
  __shared__ int wsum[17];
  __shared__ int need;
  __shared__ uint32_t prefix;
  __shared__ int n_gt, n_eq_taken;
  const float* v = vals + (size_t)blockIdx.x * L;
  int32_t* out = out_idx + (size_t)blockIdx.x * k;
  if (threadIdx.x == 0) { need = k; prefix = 0; n_gt = 0; n_eq_taken = 0; }
  __syncthreads();
  topk_select_digit(v, L, 0u, 32, 21, 11, hist, &need, &prefix);
  topk_select_digit(v, L, prefix, 21, 10, 11, hist, &need, &prefix);
  topk_select_digit(v, L, prefix, 10, 0, 10, hist, &need, &prefix);
  const uint32_t T = prefix;            // key of the k-th largest value; `need` of the elements equal to it are taken
  const int take_eq = need;
  int P = 1;
  while (P < k) P <<= 1;
  for (int i = threadIdx.x; i < P; i += blockDim.x) keys[i] = 0ull;
  __syncthreads();
  // collect in index order: everything above T, and the first `take_eq` elements equal to T
  for (int base = 0; base < L; base += TOPK_THREADS) {
    const int i = base + threadIdx.x;
    uint32_t key = 0;
    int gt = 0, eq = 0;
    if (i < L) {
      key = ord32(v[i]);
      gt = key > T;
      eq = key == T;
    }
    int tot_gt, tot_eq;
    const int pg = block_excl_scan(gt, wsum, &tot_gt);
    const int pe = block_excl_scan(eq, wsum, &tot_eq);
    const int base_gt = n_gt, base_eq = n_eq_taken;
    const unsigned long long full = ((unsigned long long)key << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)i);
    if (gt) keys[base_gt + pg] = full;
    if (eq && base_eq + pe < take_eq) keys[(k - take_eq) + base_eq + pe] = full;
    __syncthreads();
    if (threadIdx.x == 0) { n_gt = base_gt + tot_gt; n_eq_taken = min(take_eq, base_eq + tot_eq); }
    __syncthreads();
  }
  // bitonic sort, descending (padding keys are 0 = smallest)
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < P / 2; t += blockDim.x) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool desc = ((lo & size) == 0);
        const unsigned long long x = keys[lo], y = keys[hi];
        if ((x < y) == desc) { keys[lo] = y; keys[hi] = x; }
      }
      __syncthreads();
    }
  }
  for (int j = threadIdx.x; j < k; j += blockDim.x) out[j] = (int32_t)(0xFFFFFFFFu - (uint32_t)keys[j]);
}

// ---- the same selection spread over the device (an image of D2 at 1024 x 1024 has 1.37 M (anchor, class) values and a
// batch share of 2 images: one block per image left 254 CUs idle for 4.4 ms).  Every value gets a DISTINCT 53-bit composite
// key (order-preserving value bits << 21 | 2^21 - 1 - index): the k largest composites are exactly "value descending, ties ->
// lower index".  Radix select, 5 digits (11, 11, 11, 10, 10 bits): per digit one grid-wide histogram pass (LDS histogram per
// block, merged with atomics) - every block first re-derives the digits chosen so far from the previous histograms, so there
// is no separate pick launch; then one compaction pass (everything >= the k-th composite, appended in any order) and the
// bitonic sort of the k keys, one block per image.
constexpr int TK2_LEVELS = 5;
__device__ __constant__ int TK2_BITS[TK2_LEVELS] = {11, 11, 11, 10, 10};
__device__ __forceinline__ unsigned long long topk_comp(float v, int i) {
  return ((unsigned long long)ord32(v) << 21) | (unsigned long long)(0x1FFFFFu - (uint32_t)i);
}
// digits chosen on levels < level (deterministic, identical in every block): prefix of the k-th largest composite and how many
// elements of the current prefix class are still to be taken
__device__ __forceinline__ void topk2_resolve(const unsigned* hist_img /*[levels][2048]*/, int level, int k, unsigned long long* prefix, int* need,
                                              unsigned* sh /*[2048] scratch*/) {
  unsigned long long pre = 0;
  int rem = k;
  for (int l = 0; l < level; ++l) {
    const int nb = 1 << TK2_BITS[l];
    for (int i = threadIdx.x; i < nb; i += blockDim.x) sh[i] = hist_img[l * 2048 + i];
    __syncthreads();
    int b = nb - 1;                      // every thread walks the (<= 2048) bins itself: no broadcast, no extra barrier
    for (; b > 0; --b) {
      const int c = (int)sh[b];
      if (c >= rem) break;
      rem -= c;
    }
    pre = (pre << TK2_BITS[l]) | (unsigned long long)b;
    __syncthreads();
  }
  *prefix = pre;
  *need = rem;
}

__global__ __launch_bounds__(256) void topk2_hist_kernel(const float* vals, int L, int k, unsigned* hist /*[n_img][levels][2048]*/, int level) {
  __shared__ unsigned sh[2048];
  const int img = blockIdx.y;
  const float* v = vals + (size_t)img * L;
  unsigned* hist_img = hist + (size_t)img * TK2_LEVELS * 2048;
  unsigned long long prefix;
  int need;
  topk2_resolve(hist_img, level, k, &prefix, &need, sh);
  int used = 0;
  for (int l = 0; l < level; ++l) used += TK2_BITS[l];
  const int bits = TK2_BITS[level], shift = 53 - used - bits, nb = 1 << bits;
  for (int i = threadIdx.x; i < nb; i += 256) sh[i] = 0;
  __syncthreads();
  const int per = (L + gridDim.x - 1) / gridDim.x;
  const int lo = blockIdx.x * per, hi = min(L, lo + per);
  for (int i = lo + threadIdx.x; i < hi; i += 256) {
    const unsigned long long key = topk_comp(v[i], i);
    if (level == 0 || (key >> (shift + bits)) == prefix) atomicAdd(&sh[(unsigned)(key >> shift) & (nb - 1)], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nb; i += 256)
    if (sh[i]) atomicAdd(&hist_img[level * 2048 + i], sh[i]);
}

__global__ __launch_bounds__(256) void topk2_collect_kernel(const float* vals, int L, int k, const unsigned* hist, unsigned long long* keys /*[n_img][P]*/,
                                                            int* count /*[n_img]*/, int P) {
  __shared__ unsigned sh[2048];
  const int img = blockIdx.y;
  const float* v = vals + (size_t)img * L;
  unsigned long long kth;
  int need;
  topk2_resolve(hist + (size_t)img * TK2_LEVELS * 2048, TK2_LEVELS, k, &kth, &need, sh);   // all 53 bits: the k-th largest composite itself
  const int per = (L + gridDim.x - 1) / gridDim.x;
  const int lo = blockIdx.x * per, hi = min(L, lo + per);
  for (int i = lo + threadIdx.x; i < hi; i += 256) {
    const unsigned long long key = topk_comp(v[i], i);
    if (key >= kth) {
      const int pos = atomicAdd(&count[img], 1);
      if (pos < P) keys[(size_t)img * P + pos] = key;
    }
  }
}

__global__ __launch_bounds__(TOPK_THREADS) void topk2_sort_kernel(const unsigned long long* keys_in, const int* count, int k, int P, int32_t* out_idx) {
  __shared__ unsigned long long keys[TOPK_MAX];
  const int img = blockIdx.x;
  const int n = min(count[img], P);
  for (int i = threadIdx.x; i < P; i += blockDim.x) keys[i] = i < n ? keys_in[(size_t)img * P + i] : 0ull;
  __syncthreads();
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < P / 2; t += blockDim.x) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool desc = ((lo & size) == 0);
        const unsigned long long x = keys[lo], y = keys[hi];
        if ((x < y) == desc) { keys[lo] = y; keys[hi] = x; }
      }
      __syncthreads();
    }
  }
  int32_t* out = out_idx + (size_t)img * k;
  for (int j = threadIdx.x; j < k; j += blockDim.x) out[j] = (int32_t)(0x1FFFFFu - (uint32_t)(keys[j] & 0x1FFFFFull));
}

size_t topk_workspace_bytes(int n_img, int k) {
  int P = 1;
  while (P < k) P <<= 1;
  return (size_t)n_img * (TK2_LEVELS * 2048 * sizeof(unsigned) + sizeof(int) * 4 + (size_t)P * sizeof(unsigned long long)) + 256;
}

void launch_topk(const float* vals, int n_img, int L, int k, int32_t* out_idx, void* ws, hipStream_t s) {
  static int multi = -1;
  if (multi < 0) { const char* e = getenv("UDA_TOPK_MULTI"); multi = e ? atoi(e) : 1; }
  // few images per launch and a long value list: spread each image over the device; many images: one block each is enough
  // (UDA_TOPK_MULTI=2: also for short lists - test hook; 0: never)
  if (!multi || !ws || L > (1 << 21) || (L < (1 << 16) && multi != 2) || n_img > 64 || k > TOPK_MAX) {
    hipLaunchKernelGGL(topk_kernel, dim3(n_img), dim3(TOPK_THREADS), 0, s, vals, L, k, out_idx);
    return;
  }
  int P = 1;
  while (P < k) P <<= 1;
  unsigned* hist = (unsigned*)ws;
  int* count = (int*)(hist + (size_t)n_img * TK2_LEVELS * 2048);
  unsigned long long* keys = (unsigned long long*)(((uintptr_t)(count + 4 * n_img) + 255) & ~(uintptr_t)255);
  hipMemsetAsync(ws, 0, (size_t)n_img * (TK2_LEVELS * 2048 * sizeof(unsigned) + sizeof(int) * 4), s);
  int nblk = (1024 + n_img - 1) / n_img;            // about 1024 blocks in all
  if (nblk > (L + 4095) / 4096) nblk = (L + 4095) / 4096;
  if (nblk < 1) nblk = 1;
  for (int level = 0; level < TK2_LEVELS; ++level)
    hipLaunchKernelGGL(topk2_hist_kernel, dim3(nblk, n_img), dim3(256), 0, s, vals, L, k, hist, level);
  hipLaunchKernelGGL(topk2_collect_kernel, dim3(nblk, n_img), dim3(256), 0, s, vals, L, k, hist, keys, count, P);
  hipLaunchKernelGGL(topk2_sort_kernel, dim3(n_img), dim3(TOPK_THREADS), 0, s, keys, count, k, P, out_idx);
}

// ------------------------------------------------------------------------------------ NMS (NonMaxSuppressionV5)
// Epoch k (k boxes already selected) of the reference's lazy max-heap algorithm is exactly:
//   winner_k  = argmax over live candidates of the UPDATED priority (score after the pending
//               suppression chain j = k-1 .. begin, ties -> smaller index)
//   commit    = every candidate whose STALE priority outranks winner_k is popped once in this
//               epoch: its score becomes the updated one (dropped if <= threshold / hard-suppressed)
//               and its begin index becomes k.  Nothing else is touched.
// (proof sketch in DESIGN.md).  Three data-parallel passes per epoch:
//   A bound : each 2048-candidate chunk evaluates its stale-best candidate -> lower bound on the winner
//   B eval  : evaluate every candidate whose stale priority >= bound; global max -> winner
//   C commit: apply the commit rule, record the winner.
#ifndef UDA_NMS_ITEMS
#define UDA_NMS_ITEMS 8
#endif
constexpr int NMS_ITEMS = UDA_NMS_ITEMS;
constexpr int NMS_CHUNK = 256 * NMS_ITEMS;

__device__ __forceinline__ unsigned long long nms_key(float s, int idx) {
  const uint32_t b = __float_as_uint(s);
  const uint32_t o = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
  return ((unsigned long long)o << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)idx);
}

__device__ __forceinline__ float nms_iou(const float* a, const float* b) {
  const float ymin_i = fminf(a[0], a[2]), xmin_i = fminf(a[1], a[3]);
  const float ymax_i = fmaxf(a[0], a[2]), xmax_i = fmaxf(a[1], a[3]);
  const float ymin_j = fminf(b[0], b[2]), xmin_j = fminf(b[1], b[3]);
  const float ymax_j = fmaxf(b[0], b[2]), xmax_j = fmaxf(b[1], b[3]);
  const float area_i = (ymax_i - ymin_i) * (xmax_i - xmin_i);
  const float area_j = (ymax_j - ymin_j) * (xmax_j - xmin_j);
  if (area_i <= 0.f || area_j <= 0.f) return 0.0f;
  const float iy0 = fmaxf(ymin_i, ymin_j), ix0 = fmaxf(xmin_i, xmin_j);
  const float iy1 = fminf(ymax_i, ymax_j), ix1 = fminf(xmax_i, xmax_j);
  const float inter = fmaxf(iy1 - iy0, 0.0f) * fmaxf(ix1 - ix0, 0.0f);
  return inter / (area_i + area_j - inter);
}

// LDS of the epoch kernels: the image's selected boxes + the block's compacted work list
struct NmsLds {
  float sel[4 * 128];        // up to 128 selected boxes (max_output_size <= 128 on this path)
  int list[NMS_CHUNK];       // candidates of this chunk whose pending chain must be evaluated
  float wgt[4][128];         // per wave: the chain's weights, computed 64 links at a time by the wave's lanes
  int count;
};

__device__ __forceinline__ void nms_load_sel(const NmsArgs& a, NmsLds& L, int n, int k) {
  for (int t = threadIdx.x; t < 4 * k; t += blockDim.x) L.sel[t] = a.sel_box[(size_t)n * a.M * 4 + t];
  if (threadIdx.x == 0) L.count = 0;
  __syncthreads();
}

// exact updated score of candidate i in epoch k: the reference's pending chain as written
// (j = k-1 .. begin, newest selected box first, with its early exits); -inf when dropped.
__device__ __forceinline__ float nms_chain_eval(const NmsArgs& a, const NmsLds& L, size_t base, int i, int k) {
  float score = a.stale[base + i];
  const int begin = a.begin[base + i];
  const size_t bbase = (size_t)(blockIdx.y / a.segs) * a.K;   // boxes are per image, state per problem
  const float4 b4 = *(const float4*)(a.boxes + (bbase + i) * 4);
  const float bx[4] = {b4.x, b4.y, b4.z, b4.w};
  for (int j = k - 1; j >= begin; --j) {
    const float sim = nms_iou(bx, L.sel + 4 * j);
    float w;
    if (a.soft || sim <= a.iou_thr) {
      const float e = a.scale * sim * sim;
      w = (e == 0.0f) ? 1.0f : (float)exp((double)e);
    } else {
      w = 0.0f;
    }
    score *= w;
    if (!a.soft && sim > a.iou_thr) return -INFINITY;
    if (score <= a.score_thr) return -INFINITY;
  }
  return score;
}

// The same chain for ONE candidate evaluated by a whole wave (the per-chunk bound candidate, which used to be a
// serial chain on one thread at the end of every block): the expensive part of a link (IoU + the float64 exp of
// the soft weight) does not depend on the running score, so the 64 lanes compute 64 links at a time; lane 0 then
// multiplies them into the score in exactly the reference's order with its early exits (bit-identical).
// All lanes of the wave must call it with the same arguments; every lane returns the score.
__device__ __forceinline__ float nms_chain_eval_wave(const NmsArgs& a, NmsLds& L, size_t base, int i, int k) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* wg = L.wgt[wave];
  const int begin = a.begin[base + i];
  const size_t bbase = (size_t)(blockIdx.y / a.segs) * a.K;   // boxes are per image, state per problem
  const float4 b4 = *(const float4*)(a.boxes + (bbase + i) * 4);
  const float bx[4] = {b4.x, b4.y, b4.z, b4.w};
  const int n = k - begin;                   // links j = k-1 .. begin -> slot s = k-1-j
  for (int s0 = 0; s0 < n; s0 += 64) {
    const int sl = s0 + lane;
    if (sl < n) {
      const int j = k - 1 - sl;
      const float sim = nms_iou(bx, L.sel + 4 * j);
      float w;
      if (a.soft || sim <= a.iou_thr) {
        const float e = a.scale * sim * sim;
        w = (e == 0.0f) ? 1.0f : (float)exp((double)e);
      } else {
        w = 0.0f;
      }
      if (!a.soft && sim > a.iou_thr) w = -2.0f;      // hard suppression marker (a weight is never negative)
      wg[sl] = w;
    }
  }
  __builtin_amdgcn_wave_barrier();
  float score = a.stale[base + i];
  if (lane == 0) {
    for (int sl = 0; sl < n; ++sl) {
      const float w = wg[sl];
      if (w == -2.0f) { score = -INFINITY; break; }   // reference: score *= 0, then the hard-suppression return
      score *= w;
      if (score <= a.score_thr) { score = -INFINITY; break; }
    }
  }
  __builtin_amdgcn_wave_barrier();
  return __shfl(score, 0, 64);
}

// Evaluate the block's work list, one candidate per thread round-robin (the flagged candidates
// of a chunk are spatial neighbours, so without the compaction a few waves would carry all the
// chains; lists are long and chains short, so a thread per candidate beats a wave per candidate: measured 5x).
// Records tent / ub / ev; returns the best key among the entries this thread evaluated.
__device__ __forceinline__ unsigned long long nms_run_list(const NmsArgs& a, NmsLds& L, size_t base, int k) {
  __syncthreads();
  unsigned long long best = 0ull;
  const int cnt = L.count;
  for (int e = threadIdx.x; e < cnt; e += blockDim.x) {
    const int i = L.list[e];
    const float s = nms_chain_eval(a, L, base, i, k);
    a.tent[base + i] = s;
    a.ub[base + i] = s;       // weights are <= 1: no later epoch can score this candidate higher
    a.ev[base + i] = k;
    if (s != -INFINITY) {
      const unsigned long long key = nms_key(s, i);
      best = key > best ? key : best;
    }
  }
  __syncthreads();
  return best;
}

__device__ __forceinline__ unsigned long long block_max_key(unsigned long long v) {
  __shared__ unsigned long long wmax[4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = v;
  __syncthreads();
  unsigned long long r = wmax[0];
#pragma unroll
  for (int i = 1; i < 4; ++i) r = wmax[i] > r ? wmax[i] : r;
  return r;
}

__global__ __launch_bounds__(256) void nms_init_kernel(NmsArgs a, const float* scores) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)a.n_img * a.K;
  if (gid < total) {
    const int seg = (int)(gid / a.K), i = (int)(gid % a.K);
    const size_t src = (size_t)(seg / a.segs) * a.K + i;
    const float s = scores[src];
    const bool member = (a.segs == 1) || (a.classes[src] == seg % a.segs);
    const float v = (member && s > a.score_thr) ? s : -INFINITY;
    a.stale[gid] = v;
    a.ub[gid] = v;
    a.ev[gid] = -1;
    a.begin[gid] = 0;
  }
  if (gid < (int64_t)a.n_img * a.M) {
    a.bound_key[gid] = 0ull;
    a.win_key[gid] = 0ull;
    a.sel_idx[gid] = 0;
    a.sel_score[gid] = 0.f;
  }
  if (gid < a.n_img) {
    a.nsel[gid] = 0;
    a.done[gid] = 0;
  }
}

// A bound: every chunk evaluates its best candidate by cached upper bound -> lower bound on the winner
__global__ __launch_bounds__(256) void nms_bound_kernel(NmsArgs a, int k) {
  __shared__ NmsLds L;
  const int n = blockIdx.y;
  if (a.done[n]) return;
  nms_load_sel(a, L, n, k);
  const size_t base = (size_t)n * a.K;
  const int i0 = blockIdx.x * NMS_CHUNK;
  unsigned long long best = 0ull;
#pragma unroll
  for (int it = 0; it < NMS_ITEMS; ++it) {
    const int i = i0 + it * 256 + threadIdx.x;
    if (i < a.K) {
      const float u = a.ub[base + i];
      if (u != -INFINITY && a.stale[base + i] != -INFINITY) {
        const unsigned long long key = nms_key(u, i);
        best = key > best ? key : best;
      }
    }
  }
  best = block_max_key(best);
  if (threadIdx.x < 64 && best != 0ull) {       // wave 0, cooperatively
    const int idx = (int)(0xFFFFFFFFu - (uint32_t)best);
    const float s = nms_chain_eval_wave(a, L, base, idx, k);
    if (threadIdx.x == 0) {
      a.tent[base + idx] = s;
      a.ub[base + idx] = s;
      a.ev[base + idx] = k;
      if (s != -INFINITY) atomicMax(&a.bound_key[(size_t)n * a.M + k], nms_key(s, idx));
    }
  }
}

// B eval: exact scores for every candidate whose upper bound can still reach the bound -> winner
__global__ __launch_bounds__(256) void nms_eval_kernel(NmsArgs a, int k) {
  __shared__ NmsLds L;
  const int n = blockIdx.y;
  if (a.done[n]) return;
  nms_load_sel(a, L, n, k);
  const size_t base = (size_t)n * a.K;
  const unsigned long long bound = a.bound_key[(size_t)n * a.M + k];
  const int i0 = blockIdx.x * NMS_CHUNK;
  unsigned long long best = 0ull;
#pragma unroll
  for (int it = 0; it < NMS_ITEMS; ++it) {
    const int i = i0 + it * 256 + threadIdx.x;
    if (i < a.K) {
      const float u = a.ub[base + i];
      if (u != -INFINITY && a.stale[base + i] != -INFINITY && nms_key(u, i) >= bound) {
        if (a.ev[base + i] == k) {                 // already exact (the chunk's bound candidate)
          const unsigned long long key = nms_key(a.tent[base + i], i);
          best = key > best ? key : best;
        } else {
          L.list[atomicAdd(&L.count, 1)] = i;
        }
      }
    }
  }
  const unsigned long long b2 = nms_run_list(a, L, base, k);
  best = b2 > best ? b2 : best;
  best = block_max_key(best);
  if (threadIdx.x == 0 && best != 0ull) atomicMax(&a.win_key[(size_t)n * a.M + k], best);
}

// C commit: the winner is selected; every candidate whose STALE priority outranks it is popped
// once in this epoch (score <- updated score, begin <- k), exactly the reference's heap traffic.
// The same pass then does the bound step of epoch k+1 (its chunk's best candidate by upper bound,
// evaluated against the k+1 selected boxes), so an epoch costs two launches instead of three.
__global__ __launch_bounds__(256) void nms_commit_kernel(NmsArgs a, int k) {
  __shared__ NmsLds L;
  const int n = blockIdx.y;
  if (a.done[n]) return;
  const size_t base = (size_t)n * a.K;
  const unsigned long long wk = a.win_key[(size_t)n * a.M + k];
  if (wk == 0ull) {
    if (blockIdx.x == 0 && threadIdx.x == 0) a.done[n] = 1;
    return;
  }
  nms_load_sel(a, L, n, k);
  const int widx = (int)(0xFFFFFFFFu - (uint32_t)wk);
  const int i0 = blockIdx.x * NMS_CHUNK;
  bool pop[NMS_ITEMS];
#pragma unroll
  for (int it = 0; it < NMS_ITEMS; ++it) {
    const int i = i0 + it * 256 + threadIdx.x;
    pop[it] = false;
    if (i < a.K && i != widx) {
      const float s = a.stale[base + i];
      if (s != -INFINITY && nms_key(s, i) > wk) {
        pop[it] = true;
        if (a.ev[base + i] != k) L.list[atomicAdd(&L.count, 1)] = i;   // popped without having been scored this epoch
      }
    }
  }
  nms_run_list(a, L, base, k);      // (ends with a barrier: tent[] of this chunk is visible below)
  unsigned long long best = 0ull;   // best upper bound left in this chunk, for the next epoch's bound
#pragma unroll
  for (int it = 0; it < NMS_ITEMS; ++it) {
    const int i = i0 + it * 256 + threadIdx.x;
    if (i >= a.K) continue;
    float u = a.ub[base + i];
    bool alive = a.stale[base + i] != -INFINITY;
    if (pop[it]) {
      u = a.tent[base + i];
      a.stale[base + i] = u;
      a.begin[base + i] = k;
      alive = (u != -INFINITY);
    }
    if (i == widx) {
      const size_t o = (size_t)n * a.M + k;
      a.sel_idx[o] = i;
      a.sel_score[o] = a.tent[base + i];
      const float* bx = a.boxes + ((size_t)(n / a.segs) * a.K + i) * 4;
      a.sel_box[o * 4 + 0] = bx[0];
      a.sel_box[o * 4 + 1] = bx[1];
      a.sel_box[o * 4 + 2] = bx[2];
      a.sel_box[o * 4 + 3] = bx[3];
      a.stale[base + i] = -INFINITY;
      a.nsel[n] = k + 1;
      alive = false;
    }
    if (alive && u != -INFINITY) {
      const unsigned long long key = nms_key(u, i);
      best = key > best ? key : best;
    }
  }
  if (k + 1 >= a.M) return;
  best = block_max_key(best);
  // bound of epoch k+1: selected box k is the winner's box (read from the candidate table: the
  // sel_box row may be written by another block of this launch)
  if (threadIdx.x < 4) L.sel[4 * k + threadIdx.x] = a.boxes[((size_t)(n / a.segs) * a.K + widx) * 4 + threadIdx.x];
  __syncthreads();
  if (threadIdx.x < 64 && best != 0ull) {       // wave 0, cooperatively
    const int idx = (int)(0xFFFFFFFFu - (uint32_t)best);
    const float s = nms_chain_eval_wave(a, L, base, idx, k + 1);
    if (threadIdx.x == 0) {
      a.tent[base + idx] = s;
      a.ub[base + idx] = s;
      a.ev[base + idx] = k + 1;
      if (s != -INFINITY) atomicMax(&a.bound_key[(size_t)n * a.M + k + 1], nms_key(s, idx));
    }
  }
}

void launch_nms_init(const NmsArgs& a, const float* scores, hipStream_t s) {
  int64_t total = (int64_t)a.n_img * a.K;
  const int64_t t2 = (int64_t)a.n_img * a.M;
  if (t2 > total) total = t2;
  hipLaunchKernelGGL(nms_init_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a, scores);
}

void launch_nms_epoch(const NmsArgs& a, int epoch, hipStream_t s) {
  const dim3 grid((a.K + NMS_CHUNK - 1) / NMS_CHUNK, a.n_img), block(256);
  if (epoch == 0) hipLaunchKernelGGL(nms_bound_kernel, grid, block, 0, s, a, epoch);   // later bounds ride on the commit pass
  hipLaunchKernelGGL(nms_eval_kernel, grid, block, 0, s, a, epoch);
  hipLaunchKernelGGL(nms_commit_kernel, grid, block, 0, s, a, epoch);
}

void launch_nms_finish(const NmsArgs&, int, hipStream_t) {}

// ------------------------------------------------------------------------------------ NMS, one launch
// The same epoch rule executed by ONE block per problem (image, or image x class) for all epochs: problems are
// independent, so nothing but __syncthreads is needed and the 2 x max_output_size dependent launches of the grid
// version become one.  The candidates are cut into chunks of 1024 (one per thread); the block keeps, per chunk,
//   cub[ch] = max key over alive candidates of the cached upper bound ub (<= stale: drives the search for the winner)
//   cst[ch] = max key over alive candidates of the stale score          (drives the pops)
// in LDS, so an epoch only visits chunks that can matter:
//   search: repeatedly take the unvisited chunk with the largest cub > L, evaluate the exact score of every
//           candidate in it with key(ub) > L (one thread each), L = max(L, best exact key) - until no chunk beats L;
//   pops  : every chunk with cst > key(winner) (and the winner's chunk): candidates whose STALE priority outranks
//           the winner take their exact score (begin = k), the winner is recorded and removed; cub / cst of the
//           chunk are recomputed.
// Identical selections and scores to the grid version (and to the heap of the reference) bit for bit.
constexpr int SOLO_T = 1024;
constexpr int SOLO_MAXCH = 512;

struct SoloLds {
  float sel[4 * 128];
  unsigned long long cub[SOLO_MAXCH], cst[SOLO_MAXCH];
  unsigned long long r0[SOLO_T / 64], r1[SOLO_T / 64];
  unsigned char visited[SOLO_MAXCH];
  float wgt[128];            // link weights of the epoch's first bound candidate (computed by wave 0, 64 links at a time)
  unsigned long long L;
  int pick;
};

__device__ __forceinline__ float solo_chain(const NmsArgs& a, const SoloLds& S, size_t base, size_t bbase, int i, int k) {
  float score = a.stale[base + i];
  const int begin = a.begin[base + i];
  const float4 b4 = *(const float4*)(a.boxes + (bbase + i) * 4);
  const float bx[4] = {b4.x, b4.y, b4.z, b4.w};
  for (int j = k - 1; j >= begin; --j) {
    const float sim = nms_iou(bx, S.sel + 4 * j);
    float w;
    if (a.soft || sim <= a.iou_thr) {
      const float e = a.scale * sim * sim;
      w = (e == 0.0f) ? 1.0f : (float)exp((double)e);
    } else {
      w = 0.0f;
    }
    score *= w;
    if (!a.soft && sim > a.iou_thr) return -INFINITY;
    if (score <= a.score_thr) return -INFINITY;
  }
  return score;
}

// max of two keys over the block; results valid in every thread
__device__ __forceinline__ void solo_max2(SoloLds& S, unsigned long long& v0, unsigned long long& v1) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o0 = __shfl_xor(v0, off, 64), o1 = __shfl_xor(v1, off, 64);
    v0 = o0 > v0 ? o0 : v0;
    v1 = o1 > v1 ? o1 : v1;
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { S.r0[threadIdx.x >> 6] = v0; S.r1[threadIdx.x >> 6] = v1; }
  __syncthreads();
  v0 = S.r0[0]; v1 = S.r1[0];
#pragma unroll
  for (int w = 1; w < SOLO_T / 64; ++w) {
    v0 = S.r0[w] > v0 ? S.r0[w] : v0;
    v1 = S.r1[w] > v1 ? S.r1[w] : v1;
  }
}

__global__ __launch_bounds__(SOLO_T) void nms_solo_kernel(NmsArgs a, const float* scores) {
  __shared__ SoloLds S;
  const int n = blockIdx.x, tid = threadIdx.x;
  const size_t base = (size_t)n * a.K;
  const int img = n / a.segs;
  const size_t bbase = (size_t)img * a.K;
  const int NCH = (a.K + SOLO_T - 1) / SOLO_T;
  // ---- initial state and chunk maxima
  for (int ch = 0; ch < NCH; ++ch) {
    const int i = ch * SOLO_T + tid;
    unsigned long long key = 0ull, dummy = 0ull;
    if (i < a.K) {
      const float s = scores[bbase + i];
      const bool member = (a.segs == 1) || (a.classes[bbase + i] == n % a.segs);
      const float v = (member && s > a.score_thr) ? s : -INFINITY;
      a.stale[base + i] = v;
      a.ub[base + i] = v;
      a.ev[base + i] = -1;
      a.begin[base + i] = 0;
      if (v != -INFINITY) key = nms_key(v, i);
    }
    solo_max2(S, key, dummy);
    if (tid == 0) { S.cub[ch] = key; S.cst[ch] = key; }
  }
  if (tid < a.M) {
    a.sel_idx[(size_t)n * a.M + tid] = 0;
    a.sel_score[(size_t)n * a.M + tid] = 0.f;
  }
  if (tid == 0) a.nsel[n] = 0;
  __syncthreads();

  for (int k = 0; k < a.M; ++k) {
    // ---- search for the winner of epoch k
    for (int ch = tid; ch < NCH; ch += SOLO_T) S.visited[ch] = 0;
    if (tid == 0) S.L = 0ull;
    __syncthreads();
    // a first lower bound: the exact score of the candidate with the largest upper bound of all (its index is in the
    // key), evaluated by wave 0 with the links spread over the lanes - otherwise the first visited chunk would have to
    // evaluate every one of its candidates
    if (tid < 64) {
      unsigned long long bk = 0ull;
      for (int ch = tid; ch < NCH; ch += 64) bk = S.cub[ch] > bk ? S.cub[ch] : bk;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long ok = __shfl_xor(bk, off, 64);
        bk = ok > bk ? ok : bk;
      }
      if (bk != 0ull) {
        const int i = (int)(0xFFFFFFFFu - (uint32_t)bk);
        const int begin = a.begin[base + i];
        const float4 b4 = *(const float4*)(a.boxes + (bbase + i) * 4);
        const float bx[4] = {b4.x, b4.y, b4.z, b4.w};
        const int nl = k - begin;
        for (int s0 = 0; s0 < nl; s0 += 64) {
          const int sl = s0 + tid;
          if (sl < nl) {
            const float sim = nms_iou(bx, S.sel + 4 * (k - 1 - sl));
            float w;
            if (a.soft || sim <= a.iou_thr) {
              const float e = a.scale * sim * sim;
              w = (e == 0.0f) ? 1.0f : (float)exp((double)e);
            } else {
              w = 0.0f;
            }
            if (!a.soft && sim > a.iou_thr) w = -2.0f;
            S.wgt[sl] = w;
          }
        }
        __builtin_amdgcn_wave_barrier();
        if (tid == 0) {
          float score = a.stale[base + i];
          for (int sl = 0; sl < nl; ++sl) {
            const float w = S.wgt[sl];
            if (w == -2.0f) { score = -INFINITY; break; }
            score *= w;
            if (score <= a.score_thr) { score = -INFINITY; break; }
          }
          a.tent[base + i] = score;
          a.ub[base + i] = score;
          a.ev[base + i] = k;
          if (score != -INFINITY) S.L = nms_key(score, i) - 1ull;   // "- 1": the candidate itself must still pass the > L tests below
        }
      }
    }
    __syncthreads();
    while (true) {
      if (tid < 64) {            // wave 0 picks the unvisited chunk with the largest upper bound above L
        const unsigned long long L = S.L;
        unsigned long long bk = 0ull;
        int bc = -1;
        for (int ch = tid; ch < NCH; ch += 64) {
          const unsigned long long v = S.cub[ch];
          if (!S.visited[ch] && v > L && v > bk) { bk = v; bc = ch; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          const unsigned long long ok = __shfl_xor(bk, off, 64);
          const int oc = __shfl_xor(bc, off, 64);
          if (ok > bk || (ok == bk && oc >= 0 && (bc < 0 || oc < bc))) { bk = ok; bc = oc; }
        }
        if (tid == 0) S.pick = bc;
      }
      __syncthreads();
      const int ch = S.pick;
      if (ch < 0) break;
      const unsigned long long L = S.L;
      const int i = ch * SOLO_T + tid;
      unsigned long long ke = 0ull, ku = 0ull;
      if (i < a.K) {
        float u = a.ub[base + i];
        if (u != -INFINITY && a.stale[base + i] != -INFINITY) {
          if (nms_key(u, i) > L) {
            float sc;
            if (a.ev[base + i] == k) {
              sc = a.tent[base + i];             // already exact in this epoch (the first bound candidate)
            } else {
              sc = solo_chain(a, S, base, bbase, i, k);
              a.tent[base + i] = sc;
              a.ub[base + i] = sc;
              a.ev[base + i] = k;
            }
            u = sc;
            if (sc != -INFINITY) ke = nms_key(sc, i);
          }
          if (u != -INFINITY) ku = nms_key(u, i);
        }
      }
      solo_max2(S, ke, ku);
      if (tid == 0) {
        S.cub[ch] = ku;
        S.visited[ch] = 1;
        if (ke > S.L) S.L = ke;
      }
      __syncthreads();
    }
    const unsigned long long wk = S.L;
    if (wk == 0ull) break;                       // no live candidate left: the remaining slots stay padded
    const int widx = (int)(0xFFFFFFFFu - (uint32_t)wk);
    const int wch = widx / SOLO_T;
    // ---- pops and the winner
    for (int ch = 0; ch < NCH; ++ch) {
      if (!(S.cst[ch] > wk || ch == wch)) continue;     // uniform: LDS values written before the last barrier
      const int i = ch * SOLO_T + tid;
      unsigned long long ks = 0ull, ku = 0ull;
      if (i < a.K) {
        float st = a.stale[base + i];
        if (st != -INFINITY) {
          if (i == widx) {
            const size_t o = (size_t)n * a.M + k;
            const float ws = a.tent[base + i];
            a.sel_idx[o] = i;
            a.sel_score[o] = ws;
            const float4 b4 = *(const float4*)(a.boxes + (bbase + i) * 4);
            *(float4*)(a.sel_box + o * 4) = b4;
            S.sel[4 * k + 0] = b4.x; S.sel[4 * k + 1] = b4.y; S.sel[4 * k + 2] = b4.z; S.sel[4 * k + 3] = b4.w;
            a.stale[base + i] = -INFINITY;
            a.nsel[n] = k + 1;
            st = -INFINITY;
          } else if (nms_key(st, i) > wk) {
            float e;
            if (a.ev[base + i] == k) {
              e = a.tent[base + i];
            } else {
              e = solo_chain(a, S, base, bbase, i, k);
              a.tent[base + i] = e;
              a.ub[base + i] = e;
              a.ev[base + i] = k;
            }
            a.stale[base + i] = e;
            a.begin[base + i] = k;
            st = e;
          }
          if (st != -INFINITY) {
            ks = nms_key(st, i);
            const float u = a.ub[base + i];
            if (u != -INFINITY) ku = nms_key(u, i);
          }
        }
      }
      solo_max2(S, ks, ku);
      if (tid == 0) { S.cst[ch] = ks; S.cub[ch] = ku; }
      __syncthreads();
    }
    __syncthreads();
  }
}

bool nms_solo_supported(const NmsArgs& a) { return a.K <= SOLO_T * SOLO_MAXCH && a.M <= 128; }

void launch_nms_solo(const NmsArgs& a, const float* scores, hipStream_t s) {
  if (a.n_img <= 0) return;
  hipLaunchKernelGGL(nms_solo_kernel, dim3(a.n_img), dim3(SOLO_T), 0, s, a, scores);
}

// ------------------------------------------------------------------------------------ NMS, one launch, state in registers
// Problems of up to 8192 candidates (top-k / per-class paths, score prefixes): the same epoch rule with the whole
// per-candidate state (stale score, cached exact score = upper bound, begin, epoch of the cached score, box) in the
// registers of the thread that owns the candidate (candidate i = j * 1024 + tid), the selected boxes in LDS.  Nothing
// is read from memory inside the epoch loop; an epoch costs three block reductions:
//   1. the candidate with the largest upper bound takes its exact score (links spread over the lanes of wave 0)
//      -> lower bound L on the winner;
//   2. every candidate whose upper bound beats L takes its exact score (one thread each, in parallel); max -> winner;
//   3. pops: every candidate whose STALE priority outranks the winner becomes exact (begin = k); the winner is recorded.
// Bit-identical selections and scores to the grid version and to the reference's heap.
struct RegLds {
  alignas(16) float sel[4 * 128];
  unsigned long long red2[2][SOLO_T / 64];     // reg_max2: alternating halves
  float wgt[128];
  unsigned long long L;
  float pbox[4];
  float pscore;
  int pbegin;
};

// Maximum of a 64-bit key over the 64 lanes of a wave, in every lane: two 32-bit DPP reductions (row_shr 1 / 2 / 4 / 8, row_bcast
// 15 / 31 - seven instructions each, no LDS traffic; a __shfl_xor ladder is six dependent ds_bpermute round trips per word):
// the high word (ordered score) first, then the low word (~index) among the lanes that hold the maximal high word.
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  unsigned t;
  t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); v = t > v ? t : v;   // row_shr:1
  t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); v = t > v ? t : v;   // row_shr:2
  t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xe, false); v = t > v ? t : v;   // row_shr:4, banks 1-3
  t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xc, false); v = t > v ? t : v;   // row_shr:8, banks 2-3
  t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); v = t > v ? t : v;   // row_bcast:15, rows 1 and 3
  t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); v = t > v ? t : v;   // row_bcast:31, rows 2 and 3
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned long long wave_max_key(unsigned long long k) {
  const unsigned hi = (unsigned)(k >> 32), lo = (unsigned)k;
  const unsigned mh = wave_max_u32(hi);
  const unsigned ml = wave_max_u32(hi == mh ? lo : 0u);
  return ((unsigned long long)mh << 32) | ml;
}

// Block-wide maximum of a 64-bit key with ONE barrier: per wave by DPP (wave_max_key: 14 instructions; a __shfl_xor ladder was 12
// dependent ds_bpermute round trips), the waves' maxima to alternating halves of red2 (par = 0, 1, 0, ... - uniform over the block),
// so a call never overwrites what a slower wave may still be reading from the call before (it read that before it arrived at THIS
// call's barrier).  The caller must not rely on a barrier in front of the reduction.
__device__ __forceinline__ unsigned long long reg_max2(RegLds& S, unsigned long long v, int& par) {
  v = wave_max_key(v);
  unsigned long long* red = S.red2[par];
  par ^= 1;
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  v = red[0];
#pragma unroll
  for (int w = 1; w < SOLO_T / 64; ++w) v = red[w] > v ? red[w] : v;
  return v;
}


// Pending chain of one candidate: links k-1 .. begin, newest first.  Most links are no-ops - a selected box that does not
// strictly overlap the candidate has IoU 0, weight exactly 1.0f - so the chain runs in two passes: a cheap interval test
// over all links that leaves a bit mask of the overlapping ones, then IoU + exp + multiply (the reference's order and
// early exits) over the set bits only.  Lanes of a wave then diverge over a handful of expensive links instead of
// paying the double-precision exp at nearly every one of up to 100 links.
// pass 1: bit mask of the links begin .. k-1 whose selected box strictly overlaps the candidate
__device__ __forceinline__ void chain_mask(const NmsArgs& a, const RegLds& S, int begin, const float* bx, int k, unsigned long long* m) {
  m[0] = 0ull; m[1] = 0ull;                      // links 0..63 / 64..127
  if (a.soft || a.iou_thr >= 0.f) {
    const float y0 = fminf(bx[0], bx[2]), x0 = fminf(bx[1], bx[3]), y1 = fmaxf(bx[0], bx[2]), x1 = fmaxf(bx[1], bx[3]);
    auto test = [&](const float4& sb, int j) {
      const float sy0 = fminf(sb.x, sb.z), sx0 = fminf(sb.y, sb.w), sy1 = fmaxf(sb.x, sb.z), sx1 = fmaxf(sb.y, sb.w);
      const bool ov = (fminf(y1, sy1) > fmaxf(y0, sy0)) && (fminf(x1, sx1) > fmaxf(x0, sx0));
      if (ov) m[j >> 6] |= 1ull << (j & 63);
    };
    int j = begin;
    // four selected boxes on their way at a time: one LDS round trip per link otherwise (up to 100 in a row for a candidate that was
    // never evaluated - the longest single-thread stretch of an epoch of the one-block kernel)
    for (; j + 4 <= k; j += 4) {
      const float4 s0 = *(const float4*)(S.sel + 4 * j), s1 = *(const float4*)(S.sel + 4 * j + 4);
      const float4 s2 = *(const float4*)(S.sel + 4 * j + 8), s3 = *(const float4*)(S.sel + 4 * j + 12);
      test(s0, j); test(s1, j + 1); test(s2, j + 2); test(s3, j + 3);
    }
    for (; j < k; ++j) test(*(const float4*)(S.sel + 4 * j), j);
  } else {                                        // (negative hard threshold: IoU 0 suppresses too - no link can be skipped)
    for (int j = begin; j < k; ++j) m[j >> 6] |= 1ull << (j & 63);
  }
}

// pass 2: IoU + exp + multiply over the set bits, newest first, with the reference's early exits
__device__ __forceinline__ float chain_product(const NmsArgs& a, const RegLds& S, float score, const float* bx, const unsigned long long* m) {
#pragma unroll
  for (int h = 1; h >= 0; --h) {
    unsigned long long mm = m[h];
    while (mm) {
      const int bit = 63 - __clzll((long long)mm);
      mm &= ~(1ull << bit);
      const int j = h * 64 + bit;
      const float sim = nms_iou(bx, S.sel + 4 * j);
      float w;
      if (a.soft || sim <= a.iou_thr) {
        const float e = a.scale * sim * sim;
        w = (e == 0.0f) ? 1.0f : (float)exp((double)e);
      } else {
        w = 0.0f;
      }
      score *= w;
      if (!a.soft && sim > a.iou_thr) return -INFINITY;
      if (score <= a.score_thr) return -INFINITY;
    }
  }
  return score;
}

// Pending chain of one candidate: links k-1 .. begin, newest first.  Most links are no-ops - a selected box that does not
// strictly overlap the candidate has IoU 0, weight exactly 1.0f - so the chain runs in two passes: a cheap interval test
// over all links that leaves a bit mask of the overlapping ones, then IoU + exp + multiply (the reference's order and
// early exits) over the set bits only.  Lanes of a wave then diverge over a handful of expensive links instead of
// paying the double-precision exp at nearly every one of up to 100 links.
__device__ __forceinline__ float reg_chain(const NmsArgs& a, const RegLds& S, float score, int begin, const float* bx, int k) {
  unsigned long long m[2];
  chain_mask(a, S, begin, bx, k, m);
  return chain_product(a, S, score, bx, m);
}

// The same chain evaluated by a whole wave (all 64 lanes call it with the same arguments): links spread over the lanes,
// 64 at a time, newest first; only the lanes whose selected box strictly overlaps the candidate compute a weight, and
// the product runs over those lanes in link order with the weights read out of the lanes' registers.  Every lane
// returns the same score.
__device__ __forceinline__ float chain_wave(const NmsArgs& a, const RegLds& S, float score, int begin, const float* pb, int k) {
  const int lane = threadIdx.x & 63, nl = k - begin;
  const bool sparse = a.soft || a.iou_thr >= 0.f;
  const float y0 = fminf(pb[0], pb[2]), x0 = fminf(pb[1], pb[3]), y1 = fmaxf(pb[0], pb[2]), x1 = fmaxf(pb[1], pb[3]);
  for (int s0 = 0; s0 < nl && score != -INFINITY; s0 += 64) {
    const int sl = s0 + lane;
    bool ov = false;
    float w = 1.0f;
    if (sl < nl) {
      const float* sb = S.sel + 4 * (k - 1 - sl);
      const float sy0 = fminf(sb[0], sb[2]), sx0 = fminf(sb[1], sb[3]), sy1 = fmaxf(sb[0], sb[2]), sx1 = fmaxf(sb[1], sb[3]);
      ov = !sparse || ((fminf(y1, sy1) > fmaxf(y0, sy0)) && (fminf(x1, sx1) > fmaxf(x0, sx0)));
      if (ov) {
        const float sim = nms_iou(pb, sb);
        if (a.soft || sim <= a.iou_thr) {
          const float e = a.scale * sim * sim;
          w = (e == 0.0f) ? 1.0f : (float)exp((double)e);
        } else {
          w = 0.0f;
        }
        if (!a.soft && sim > a.iou_thr) w = -2.0f;      // hard suppression marker (a weight is never negative)
      }
    }
    unsigned long long mm = __ballot(ov);
    while (mm) {
      const int ln = __ffsll((long long)mm) - 1;
      mm &= mm - 1ull;
      const float wl = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(w), ln));      // (ln is wave-uniform: v_readlane, not a ds_bpermute round trip per link)
      if (wl == -2.0f) { score = -INFINITY; break; }
      score *= wl;
      if (score <= a.score_thr) { score = -INFINITY; break; }
    }
  }
  return score;
}

template <int IPT>
__global__ __launch_bounds__(SOLO_T) void nms_reg_kernel(NmsArgs a, const float* scores) {
  // more than four candidates per thread: the boxes live in (dynamic) LDS instead of registers, 16 B per candidate
  constexpr bool BOXLDS = IPT > 4;
  constexpr int NBX = BOXLDS ? 1 : IPT;
  extern __shared__ float4 reg_boxes[];
  __shared__ RegLds S;
  const int n = blockIdx.x, tid = threadIdx.x;
  const int img = n / a.segs;
  const size_t bbase = (size_t)img * a.K;
  float st[IPT], ub[IPT], bx[NBX][4];
  int be[IPT];                  // begin | (epoch of the cached exact score + 1) << 8
#pragma unroll
  for (int j = 0; j < IPT; ++j) {
    const int i = j * SOLO_T + tid;
    float v = -INFINITY;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < a.K) {
      const float s = scores[bbase + i];
      const bool member = (a.segs == 1) || (a.classes[bbase + i] == n % a.segs);
      if (member && s > a.score_thr) v = s;
      b4 = *(const float4*)(a.boxes + (bbase + i) * 4);
    }
    st[j] = v; ub[j] = v; be[j] = 0;
    if constexpr (BOXLDS) {
      reg_boxes[j * SOLO_T + tid] = b4;
    } else {
      bx[j][0] = b4.x; bx[j][1] = b4.y; bx[j][2] = b4.z; bx[j][3] = b4.w;
    }
  }
  auto box_of = [&](int j, float* o) {
    if constexpr (BOXLDS) {
      const float4 b4 = reg_boxes[j * SOLO_T + tid];
      o[0] = b4.x; o[1] = b4.y; o[2] = b4.z; o[3] = b4.w;
    } else {
      o[0] = bx[j][0]; o[1] = bx[j][1]; o[2] = bx[j][2]; o[3] = bx[j][3];
    }
  };
  if (tid < a.M) {
    a.sel_idx[(size_t)n * a.M + tid] = 0;
    a.sel_score[(size_t)n * a.M + tid] = 0.f;
  }
  int nsel = 0, rpar = 0;

  for (int k = 0; k < a.M; ++k) {
    // ---- 1. exact score of the candidate with the largest upper bound
    unsigned long long bk = 0ull;
#pragma unroll
    for (int j = 0; j < IPT; ++j)
      if (st[j] != -INFINITY && ub[j] != -INFINITY) {
        const unsigned long long key = nms_key(ub[j], j * SOLO_T + tid);
        bk = key > bk ? key : bk;
      }
    bk = reg_max2(S, bk, rpar);
    if (bk == 0ull) break;                 // nothing alive
    const int bi = (int)(0xFFFFFFFFu - (uint32_t)bk);
    const bool own = (bi % SOLO_T) == tid;
    const int oj = bi / SOLO_T;
#pragma unroll
    for (int j = 0; j < IPT; ++j)
      if (own && j == oj) {
        float b[4];
        box_of(j, b);
        S.pbox[0] = b[0]; S.pbox[1] = b[1]; S.pbox[2] = b[2]; S.pbox[3] = b[3];
        S.pscore = st[j];
        S.pbegin = be[j] & 255;
      }
    __syncthreads();
    if (tid < 64) {
      const float pb[4] = {S.pbox[0], S.pbox[1], S.pbox[2], S.pbox[3]};
      const float score = chain_wave(a, S, S.pscore, S.pbegin, pb, k);
      __builtin_amdgcn_wave_barrier();
      if (tid == 0) {
        S.pscore = score;
        S.L = (score != -INFINITY) ? nms_key(score, bi) - 1ull : 0ull;   // "- 1": the candidate itself passes the > L test
      }
    }
    __syncthreads();
    const unsigned long long L = S.L;
    const float pscore = S.pscore;
    // ---- 2. exact scores of everything that can still beat L
    unsigned long long ke = 0ull;
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
      if (own && j == oj) { ub[j] = pscore; be[j] = (be[j] & 255) | ((k + 1) << 8); }
      if (st[j] != -INFINITY && ub[j] != -INFINITY && nms_key(ub[j], j * SOLO_T + tid) > L) {
        if ((be[j] >> 8) != k + 1) {
          float b[4];
          box_of(j, b);
          ub[j] = reg_chain(a, S, st[j], be[j] & 255, b, k);
          be[j] = (be[j] & 255) | ((k + 1) << 8);
        }
        if (ub[j] != -INFINITY) {
          const unsigned long long key = nms_key(ub[j], j * SOLO_T + tid);
          ke = key > ke ? key : ke;
        }
      }
    }
    const unsigned long long wk = reg_max2(S, ke, rpar);
    if (wk == 0ull) break;                 // no live candidate left: the remaining slots stay padded
    const int widx = (int)(0xFFFFFFFFu - (uint32_t)wk);
    // ---- 3. pops and the winner
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
      const int i = j * SOLO_T + tid;
      if (st[j] == -INFINITY) continue;
      if (i == widx) {
        const size_t o = (size_t)n * a.M + k;
        float b[4];
        box_of(j, b);
        a.sel_idx[o] = i;
        a.sel_score[o] = ub[j];
        *(float4*)(a.sel_box + o * 4) = make_float4(b[0], b[1], b[2], b[3]);
        S.sel[4 * k + 0] = b[0]; S.sel[4 * k + 1] = b[1]; S.sel[4 * k + 2] = b[2]; S.sel[4 * k + 3] = b[3];
        st[j] = -INFINITY;
      } else if (nms_key(st[j], i) > wk) {
        if ((be[j] >> 8) != k + 1) {
          float b[4];
          box_of(j, b);
          ub[j] = reg_chain(a, S, st[j], be[j] & 255, b, k);
        }
        st[j] = ub[j];
        be[j] = k | ((k + 1) << 8);
      }
    }
    nsel = k + 1;
    // (no barrier at the end of an epoch: the winner's S.sel entry and the per-epoch words of S are next touched behind the
    // barrier of the following block maximum; four block barriers per epoch instead of seven)
  }
  if (tid == 0) a.nsel[n] = nsel;
}

bool nms_reg_supported(const NmsArgs& a) { return a.K <= SOLO_T * 8 && a.M <= 128; }

template <int IPT>
static void launch_nms_reg_t(const NmsArgs& a, const float* scores, hipStream_t s) {
  const size_t lds = IPT > 4 ? (size_t)IPT * SOLO_T * sizeof(float4) : 0;
  if (lds > 48 * 1024) {
    static bool opted = false;     // above the default limit the kernel needs an explicit opt-in
    if (!opted) {
      hipFuncSetAttribute((const void*)nms_reg_kernel<IPT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      opted = true;
    }
  }
  hipLaunchKernelGGL((nms_reg_kernel<IPT>), dim3(a.n_img), dim3(SOLO_T), lds, s, a, scores);
}

void launch_nms_reg(const NmsArgs& a, const float* scores, hipStream_t s) {
  if (a.n_img <= 0) return;
  if (a.K <= SOLO_T) launch_nms_reg_t<1>(a, scores, s);
  else if (a.K <= 2 * SOLO_T) launch_nms_reg_t<2>(a, scores, s);
  else if (a.K <= 3 * SOLO_T) launch_nms_reg_t<3>(a, scores, s);
  else if (a.K <= 4 * SOLO_T) launch_nms_reg_t<4>(a, scores, s);
  else if (a.K <= 6 * SOLO_T) launch_nms_reg_t<6>(a, scores, s);
  else launch_nms_reg_t<8>(a, scores, s);
}

// ------------------------------------------------------------------------------------ NMS, one cooperative launch
// The whole anchor set (184 k candidates per image) in ONE launch for all epochs: a problem is spread over `bpi`
// blocks of 1024 threads, each thread keeps the state of IPT candidates in registers (as in nms_reg_kernel; boxes are
// fetched from the candidate table when a chain is evaluated), and the two grid-wide dependencies of an epoch - the
// lower bound and the winner, both atomicMax on per-epoch slots - are crossed with a per-image barrier (a monotonic
// counter in memory) instead of a kernel boundary.  The grid version pays a chain of dependent global loads per
// launch (state re-read from memory, 2 launches x 100 epochs); here only the two barriers remain on the critical path.
// All blocks must be co-resident: the launcher refuses grids larger than the device holds (occupancy x CUs), and the
// spin is bounded (error flag -> every block falls through to the end), so the grid always drains.
//   A  every block: exact score of its best candidate by upper bound (wave 0) -> atomicMax(bound[k])   | barrier
//   B  every candidate whose upper bound reaches bound[k]: exact score -> atomicMax(win[k])            | barrier
//   C  pops (stale priority above the winner -> exact, begin = k); the winner's owner records it.
// Grid-wide step of a problem: every block contributes a 64-bit key, all blocks get the maximum.  Data and arrival are
// ONE word per block: slot[blk] = key (or 1 = "arrived with nothing"; slots start at 0, real keys are >= 2^63), written
// by one relaxed device-scope store; wave 0 polls the problem's slots, one per lane, until none is 0.  One store and
// one load round trip per step, no counter, no atomic read-modify-write on the critical path, and no fences: the
// blocks of a problem exchange nothing else (everything else a block reads was written by itself or is read-only;
// on a multi-XCD part a release / acquire pair writes back and invalidates the whole L2 - measured: every load
// after a fenced barrier missed).  The spin is bounded: a time-out raises *err and every later step falls through.
constexpr int COOP_MAX_BPI = 64;     // blocks per problem: one exchange slot per lane of the polling wave

__device__ unsigned long long g_nms_dbg[8];
__device__ unsigned long long g_nms_dbg2[2];
__device__ unsigned long long g_nms_dbg3[8];
__device__ unsigned long long g_nms_hist[16];
__device__ unsigned long long g_nms_win[10];     // steps that settled 1, 2, ... winners (UDA_NMS_STATS)
__device__ unsigned long long g_nms_stepmax[128][4];   // per grid-wide step: the slowest block's A / B / C+D compute, ticks (UDA_NMS_STATS)
#ifdef UDA_NMS_STATS
#define NMS_STAT(slot, v) atomicAdd(&g_nms_dbg[slot], (unsigned long long)(v))
#else
#define NMS_STAT(slot, v)
#endif

__device__ __forceinline__ float coop_chain(const NmsArgs& a, const RegLds& S, float score, int begin, size_t bidx, int k) {
  NMS_STAT(0, 1); NMS_STAT(1, k - begin); if (k - begin >= 16) NMS_STAT(2, 1); if (k - begin >= 48) NMS_STAT(3, 1);
  if (k <= begin) return score;
  const float4 b4 = *(const float4*)(a.boxes + bidx * 4);
  const float bx[4] = {b4.x, b4.y, b4.z, b4.w};
  return reg_chain(a, S, score, begin, bx, k);
}

#ifndef UDA_NMS_LDS_STATE_IPT
#define UDA_NMS_LDS_STATE_IPT 4
#endif
constexpr bool coop_lds_state(int ipt) { return ipt <= UDA_NMS_LDS_STATE_IPT; }     // per-candidate state of the block in LDS (nms_coop_kernel)
constexpr int COOP_LIST = 3072;      // entries of the block-wide work list
constexpr int COOP_HEAVY = 1024;     // entries of the list of chains handed to whole waves
constexpr int COOP_HEAVY_LINKS = 6;  // overlapping links above which a chain is evaluated by a wave
constexpr int COOP_W = 8;            // winners one grid-wide step can settle, at most (round 5)
constexpr int COOP_KL = 512;         // exact keys >= the step's bound a block can rank for its contribution
constexpr unsigned long long COOP_OVF = 2ull;    // exchange word: "more keys than I could rank" (keys are >= 2^63, 1 = nothing)

// Several winners per grid-wide step (tests/test_nms_multiwinner_model.py states the rule on the CPU):
//   A  every block offers the exact score of its best candidate by upper bound; the step's bound is the W-th largest of
//      those keys - at least W candidates are then KNOWN to have an exact key >= bound;
//   B  every candidate whose upper bound reaches the bound takes its exact score; every block contributes its W best exact
//      keys; the W best of the problem, e_1 >= e_2 >= ..., are all >= bound, i.e. above everything that was not evaluated;
//   C  e_1 is the winner of epoch k; e_j is the winner of epoch k + j - 1 as long as its box does not strictly overlap
//      e_1 .. e_{j-1} (its own score is unchanged - IoU 0, weight exactly 1 - and no other score can have grown);
//   D  the pops of those epochs, per candidate and in epoch order: while its stale key outranks e_{t+1} it is popped in
//      epoch k + t (links begin .. k + t - 1, newest first; begin = k + t) - the reference's pops, hence its products.
// One pair of exchanges settles up to W selections instead of one.
struct CoopLds {
  unsigned long long mine[COOP_W];   // this block's words for the next exchange
  unsigned long long top[COOP_W];    // result of an exchange: the largest keys of the problem, descending, 0-padded
  int ovf;                           // a block could not rank its keys: only top[0] is usable
  int nwin;                          // winners this step settles
  int kcount;                        // entries of the key list
};

// Grid-wide step of a problem with `nm` words per block (X.mine[0 .. nm): keys, 0 = none, COOP_OVF): data and arrival are
// one word each, as in coop_exchange; wave 0 polls the problem's bpi x nm words (bpi <= 64: at most COOP_W per lane) and
// leaves the nt largest keys in X.top[0 .. nt) (keys are unique: the candidate index is part of them), 0 behind them.
__device__ __forceinline__ void coop_exchange_top(unsigned long long* slots, int blk, int bpi, int nm, int nt, CoopLds& X, int* err, unsigned spin_max) {
  // (no barrier in front: X.mine is written by wave 0 and sent by wave 0; everybody else waits at the barrier behind the poll)
  __builtin_amdgcn_wave_barrier();
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    if (lane < nm) {
      const unsigned long long v = X.mine[lane];
      __hip_atomic_store(&slots[blk * nm + lane], v ? v : 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int total = bpi * nm;
    unsigned long long loc[COOP_W];
    bool ovf = false, dead = false;
#pragma unroll
    for (int q = 0; q < COOP_W; ++q) {
      unsigned long long v = 1ull;
      const int w = lane + 64 * q;
      if (w < total && !dead) {
        unsigned spins = 0;
        while ((v = __hip_atomic_load(&slots[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0ull) {
          ++spins;
          if ((spins & 4095u) == 0u || spins > spin_max) {
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { dead = true; break; }
            if (spins > spin_max) { __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); dead = true; break; }
          }
        }
      }
      if (v == COOP_OVF) ovf = true;
      loc[q] = v <= COOP_OVF ? 0ull : v;
    }
    for (int r = 0; r < COOP_W; ++r) {        // the nt largest (the rest of X.top: 0)
      unsigned long long m = 0ull;
      if (r < nt) {
        m = loc[0];
#pragma unroll
        for (int q = 1; q < COOP_W; ++q) m = loc[q] > m ? loc[q] : m;
        m = wave_max_key(m);
#pragma unroll
        for (int q = 0; q < COOP_W; ++q)
          if (loc[q] == m) loc[q] = 0ull;
      }
      if (lane == 0) X.top[r] = m;
    }
    const bool any = __ballot(ovf) != 0ull;
    if (lane == 0) X.ovf = any ? 1 : 0;
  }
  __syncthreads();
}

__device__ __forceinline__ bool coop_strict_overlap(const float* p, const float* q) {      // the test of chain_mask
  const float y0 = fminf(p[0], p[2]), x0 = fminf(p[1], p[3]), y1 = fmaxf(p[0], p[2]), x1 = fmaxf(p[1], p[3]);
  const float sy0 = fminf(q[0], q[2]), sx0 = fminf(q[1], q[3]), sy1 = fmaxf(q[0], q[2]), sx1 = fmaxf(q[1], q[3]);
  return (fminf(y1, sy1) > fmaxf(y0, sy0)) && (fminf(x1, sx1) > fmaxf(x0, sx0));
}

template <int IPT>
__global__ __launch_bounds__(SOLO_T) void nms_coop_kernel(NmsArgs a, const float* scores, unsigned long long* slots_all, int* err, int bpi, int n0, unsigned spin_max,
                                                           int wcfg) {
  // Per candidate: the stale score in a register of its thread (scanned every epoch, candidate i0 + j * 1024 of its thread), the
  // cached exact score / upper bound in LDS (scanned every epoch, 128 KB), begin / epoch / a copy of the stale score in
  // the workspace arrays in memory (touched only when a chain is evaluated, together with the candidate's box).
  extern __shared__ float U[];                    // [IPT * 1024] | work list [COOP_LIST] | heavy list | "exact this epoch" bits | key list
  int* wlist = (int*)(U + IPT * SOLO_T);
  int* hlist = wlist + COOP_LIST;
  unsigned* ebits = (unsigned*)(hlist + COOP_HEAVY);  // [IPT * 1024 / 32]
  unsigned long long* klist = (unsigned long long*)(ebits + IPT * 32);     // [COOP_KL] (8-byte aligned: every part above is a multiple of 8 bytes)
  // Small batches (IPT <= 4: tens of blocks per problem, 4096 candidates per block): box, stale score and begin of the block's
  // candidates live in LDS (24 B each, 96 KB) instead of the workspace arrays in memory - a chain evaluation then starts without
  // a memory round trip, in each of the three phases of an epoch.
  constexpr bool LST = coop_lds_state(IPT);
  float4* lbox = (float4*)(klist + COOP_KL);                               // [IPT * 1024] (16-byte aligned: 16 K + 12 K + 4 K + 128 IPT + 4 K bytes above)
  float* lstale = (float*)(lbox + (LST ? IPT * SOLO_T : 0));
  int* lbegin = (int*)(lstale + (LST ? IPT * SOLO_T : 0));
  auto st_begin = [&](int rel, size_t g) -> int { if constexpr (LST) return lbegin[rel]; else return a.begin[g]; };
  auto st_stale = [&](int rel, size_t g) -> float { if constexpr (LST) return lstale[rel]; else return a.stale[g]; };
  auto st_box = [&](int rel, size_t g) -> float4 { if constexpr (LST) return lbox[rel]; else return *(const float4*)(a.boxes + g * 4); };
  __shared__ RegLds S;
  __shared__ CoopLds X;
  __shared__ int wcount, hcount;
  const int n = n0 + blockIdx.x / bpi, blk = blockIdx.x % bpi, tid = threadIdx.x;   // n0: first problem of this launch
  const size_t bbase = (size_t)n * a.K;           // one problem per image (segs == 1)
  // Candidate <-> (block, slot rel = j * 1024 + tid).  Large batches (IPT > 8, a few blocks per problem): one contiguous range of
  // IPT * 1024 candidates per block.  Small batches (IPT <= 8: tens of blocks per problem): stripes of 64 candidates (one wave)
  // dealt round-robin to the blocks of the problem, so that the spatial neighbours of a winner - the candidates that take
  // exact scores in phase B and are popped in phase D - are spread over all blocks and every block finishes its entries in
  // one eight-lane pass (run_balanced).  Measured, round 5: batch 1 / 45 blocks 1.47 -> 1.36 ms with stripes; batch 32 / 8 blocks
  // 3.5 -> 4.3 ms (every block then has entries in every phase), hence the switch on IPT.
#ifndef UDA_NMS_STRIPE_IPT
#define UDA_NMS_STRIPE_IPT 8
#endif
  constexpr bool STRIPE = IPT <= UDA_NMS_STRIPE_IPT;
  const int istride = STRIPE ? SOLO_T * bpi : SOLO_T;
  const int cbase = STRIPE ? 0 : blk * IPT * SOLO_T;
  auto gidx = [&](int rel) {
    if constexpr (STRIPE) { const int t = rel & (SOLO_T - 1); return (rel >> 10) * istride + ((t >> 6) * bpi + blk) * 64 + (t & 63); }
    else return cbase + rel;
  };
  auto rel_of = [&](int i) {                    // (STRIPE: -1 = not this block's candidate)
    if constexpr (STRIPE) {
      const int st = i >> 6, q = st / bpi;
      if (st - q * bpi != blk) return -1;
      return (q >> 4) * SOLO_T + (q & 15) * 64 + (i & 63);
    } else {
      return i - cbase;
    }
  };
  static_assert(SOLO_T == 1024, "the stripe mapping is written for 16 waves per block");
  const int i0 = gidx(tid);                     // this thread's first candidate; slot j: i0 + j * istride
  unsigned long long* slots = slots_all + (size_t)n * a.M * (1 + COOP_W) * bpi;      // [step][bound: bpi | winners: bpi x COOP_W]
  float st[IPT];
#pragma unroll
  for (int j = 0; j < IPT; ++j) {
    const int i = i0 + j * istride;
    float v = -INFINITY;
    if (i < a.K) {
      const float s = scores[bbase + i];
      if (s > a.score_thr) v = s;
      if constexpr (LST) lbox[j * SOLO_T + tid] = *(const float4*)(a.boxes + (bbase + i) * 4);
      else { a.stale[bbase + i] = v; a.begin[bbase + i] = 0; }
    }
    if constexpr (LST) { lstale[j * SOLO_T + tid] = v; lbegin[j * SOLO_T + tid] = 0; }
    st[j] = v;
    U[j * SOLO_T + tid] = v;
  }
  // (the padded form of the outputs - index 0 / score 0 in the slots that stay empty - is written by the launcher's
  // memsets: a slot must have ONE writer inside the kernel, the L2s of different XCDs are not coherent with each other)
  int nsel = 0;
  if (tid == 0) { wcount = 0; hcount = 0; X.kcount = 0; X.nwin = 1; }
  for (int f = tid; f < IPT * 32; f += SOLO_T) ebits[f] = 0u;
  __syncthreads();

  // Exact scores of the candidates flagged in the threads' bit masks.  These are spatial neighbours, so a few threads
  // would carry all the chains: they go through a block-wide list and every thread takes entries round-robin (entries
  // beyond the capacity stay with their owners).  A thread runs the cheap mask pass of its entry; chains with few
  // overlapping links it finishes itself, the others - big boxes overlap most of the selected ones, up to 100 IoU + exp
  // in a row - go to a second list that whole waves work off (chain_wave).  stale / begin / box come from memory in one
  // round trip; "already exact in this epoch" is a bit per candidate in LDS.
  //   pops == false (phase B): exact score in epoch k (links begin .. k-1); `fn(rel, v)` receives every candidate once.
  //   pops == true  (phase D): the candidate's pops of the epochs k .. k + nw - 1 (X.top[0 .. nw) are their winners'
  //                 keys): first pop in the first epoch k + t whose winner its stale key outranks (the long chain: this
  //                 is what goes through the lists), later ones - one new link each - by the same thread; stale / begin
  //                 in memory and U[] receive the final state.
  auto run_balanced = [&](unsigned mask, int k, int nw, auto pops_tag, auto&& fn) {
    constexpr bool POPS = decltype(pops_tag)::value;
    // (wcount / hcount are 0 here: reset at the end of the previous call, with block barriers in between)
    unsigned left = 0u;
    {
      unsigned m = mask;
      while (m) {
        const int j = __ffs((int)m) - 1;
        m &= m - 1u;
        const int pos = atomicAdd(&wcount, 1);
        if (pos < COOP_LIST) wlist[pos] = j * SOLO_T + tid;
        else left |= 1u << j;
      }
    }
    __syncthreads();
    const int cnt = wcount < COOP_LIST ? wcount : COOP_LIST;
#ifdef UDA_NMS_STATS
    const unsigned long long tb0 = wall_clock64();
    if (tid == 0) { atomicAdd(&g_nms_dbg3[0], (unsigned long long)wcount); atomicMax(&g_nms_dbg3[1], (unsigned long long)wcount); }
#endif
    // epoch of the candidate's first pop: the first winner its stale key outranks (the mask guarantees there is one)
    auto first_pop = [&](float stale, int gi) {
      const unsigned long long key = nms_key(stale, gi);
      int t = 0;
      while (t < nw - 1 && !(key > X.top[t])) ++t;
      return t;
    };
    // the candidate's later pops after a first pop in epoch k + t: one epoch at a time, each over the links that are new
    auto later_pops = [&](int rel, size_t g, float v, int t, const float* bx) {
      int b = k + t;
      if (POPS) {
        for (int u = t + 1; u < nw && v != -INFINITY; ++u) {
          if (nms_key(v, gidx(rel)) > X.top[u]) {
            v = reg_chain(a, S, v, b, bx, k + u);
            b = k + u;
          }
        }
        if constexpr (LST) { lstale[rel] = v; lbegin[rel] = b; }
        else { a.stale[g] = v; a.begin[g] = b; }
      }
      U[rel] = v;
      return v;
    };
    auto one = [&](int rel) {
      const bool cached = (ebits[rel >> 5] >> (rel & 31)) & 1u;
      if (!POPS && cached) { fn(rel, U[rel]); return; }
      const int gi = gidx(rel);
      const size_t g = bbase + gi;
      const int begin = st_begin(rel, g);
      const float stale = st_stale(rel, g);
      const float4 b4 = st_box(rel, g);
      const float bx[4] = {b4.x, b4.y, b4.z, b4.w};
      const int t = POPS ? first_pop(stale, gi) : 0;
      const int kk = k + t;
      float v = stale;
      if (POPS && t == 0 && cached) {
        v = U[rel];                          // exact in epoch k already (phase B): the same links in the same order
      } else if (kk > begin) {
        unsigned long long m[2];
        chain_mask(a, S, begin, bx, kk, m);
        if (__popcll(m[0]) + __popcll(m[1]) > COOP_HEAVY_LINKS) {
          const int pos = atomicAdd(&hcount, 1);
          if (pos < COOP_HEAVY) { hlist[pos] = rel; return; }
        }
        v = chain_product(a, S, stale, bx, m);
      }
      v = later_pops(rel, g, v, t, bx);
      if (!POPS) atomicOr(&ebits[rel >> 5], 1u << (rel & 31));
      fn(rel, v);
    };
    // Few entries (the usual case: a handful per block and phase): GL = SIXTEEN lanes per entry.  The interval test over the links
    // begin .. kk-1 - up to 100 LDS reads and compares in a row for a candidate that was never evaluated, the longest
    // single-thread stretch of an epoch (slowest block of a step 7-8 us in B and in D at batch 1, mean 3) - is spread over
    // the lanes of the group, GL links per pass; a wave ballot hands every group its GL overlap bits.  The rest of
    // the entry (product over the few overlapping links, stores) is done by the group's first lane, exactly as below.
    // (The product spread over the group's lanes as well - a weight per lane, then the product in link order, no heavy list -
    // was measured in the same job: batch 1 / 4 / 8 1.42 / 1.42 / 1.58 ms against 1.40 / 1.40 / 1.59 without; not kept.)
    // Many entries (pops with short chains): one thread per entry as before - the same arithmetic either way.
#ifndef UDA_NMS_GL
#define UDA_NMS_GL 16     // (A/B in one job, batch 1 / 4 / 8 NMS ms: 4 lanes 1.49 / 1.50 / 1.59, 8: 1.40 / 1.39 / 1.51, 16: 1.36 / 1.35 / 1.45, 32: 1.34 / 1.34 / 1.47)
#endif
    constexpr int GL = UDA_NMS_GL;          // lanes per entry (a power of two <= 32)
    if (cnt <= 2 * (SOLO_T / GL)) {
      const int lane = tid & 63, sub = tid & (GL - 1);
      const bool sparse = a.soft || a.iou_thr >= 0.f;
      for (int e0 = 0; e0 < cnt; e0 += SOLO_T / GL) {
        const int e = e0 + tid / GL;
        const bool valid = e < cnt;
        const int rel = valid ? wlist[e] : 0;
        bool cached = false, domask = false, plain = false;
        int gi = 0, begin = 0, t = 0, kk = 0;
        size_t g = 0;
        float stale = 0.f, bx[4] = {0.f, 0.f, 0.f, 0.f};
        if (valid) {
          cached = (ebits[rel >> 5] >> (rel & 31)) & 1u;
          plain = !POPS && cached;
          if (!plain) {
            gi = gidx(rel);
            g = bbase + gi;
            begin = st_begin(rel, g);
            stale = st_stale(rel, g);
            const float4 b4 = st_box(rel, g);
            bx[0] = b4.x; bx[1] = b4.y; bx[2] = b4.z; bx[3] = b4.w;
            t = POPS ? first_pop(stale, gi) : 0;
            kk = k + t;
            domask = !(POPS && t == 0 && cached) && kk > begin;
          }
        }
        unsigned long long m[2] = {0ull, 0ull};
        {
          const float y0 = fminf(bx[0], bx[2]), x0 = fminf(bx[1], bx[3]), y1 = fmaxf(bx[0], bx[2]), x1 = fmaxf(bx[1], bx[3]);
          const int npass = domask ? (kk - begin + GL - 1) / GL : 0;
          for (int it = 0; __ballot(it < npass) != 0ull; ++it) {       // (wave-uniform trip count: the longest chain of the wave)
            const int j = begin + it * GL + sub;
            bool ov = false;
            if (it < npass && j < kk) {
              if (sparse) {
                const float4 sb = *(const float4*)(S.sel + 4 * j);
                const float sy0 = fminf(sb.x, sb.z), sx0 = fminf(sb.y, sb.w), sy1 = fmaxf(sb.x, sb.z), sx1 = fmaxf(sb.y, sb.w);
                ov = (fminf(y1, sy1) > fmaxf(y0, sy0)) && (fminf(x1, sx1) > fmaxf(x0, sx0));
              } else {
                ov = true;
              }
            }
            const unsigned long long bal = __ballot(ov);
            const unsigned long long byte = (bal >> (lane & ~(GL - 1))) & ((1ull << GL) - 1ull);     // this group's links begin + GL it .. + GL - 1
            if (byte) {
              const int pos = begin + it * GL;                       // < 128 (M <= 128)
              if (pos < 64) {
                m[0] |= byte << pos;
                if (pos > 64 - GL) m[1] |= byte >> (64 - pos);
              } else {
                m[1] |= byte << (pos - 64);
              }
            }
          }
        }
        if (valid && sub == 0) {
          if (plain) {
            fn(rel, U[rel]);
          } else {
            float v = stale;
            bool heavy = false;
            if (POPS && t == 0 && cached) {
              v = U[rel];
            } else if (kk > begin) {
              if (__popcll(m[0]) + __popcll(m[1]) > COOP_HEAVY_LINKS) {
                const int pos = atomicAdd(&hcount, 1);
                if (pos < COOP_HEAVY) { hlist[pos] = rel; heavy = true; }
              }
              if (!heavy) v = chain_product(a, S, stale, bx, m);
            }
            if (!heavy) {
              v = later_pops(rel, g, v, t, bx);
              if (!POPS) atomicOr(&ebits[rel >> 5], 1u << (rel & 31));
              fn(rel, v);
            }
          }
        }
      }
    }
    // one thread per entry: long lists, and the entries that did not fit the list (they stay with their owners).  (ONE loop, so
    // that the entry code exists once per phase: the epoch loop is 50 KB of instructions as it is.)
    {
      int e = (cnt <= 2 * (SOLO_T / GL)) ? cnt : tid;
      for (;;) {
        int rel;
        if (e < cnt) { rel = wlist[e]; e += SOLO_T; }
        else if (left) { const int j = __ffs((int)left) - 1; left &= left - 1u; rel = j * SOLO_T + tid; }
        else break;
        one(rel);
      }
    }
    __syncthreads();
#ifdef UDA_NMS_STATS
    const unsigned long long tb1 = wall_clock64();
    if (tid == 0) { atomicAdd(&g_nms_dbg3[2], (unsigned long long)hcount); atomicMax(&g_nms_dbg3[3], (unsigned long long)hcount);
                    atomicAdd(&g_nms_dbg3[4], tb1 - tb0); atomicMax(&g_nms_dbg3[5], tb1 - tb0); }
#endif
    const int hc = hcount < COOP_HEAVY ? hcount : COOP_HEAVY;
    if (hc == 0) {                           // the common case: no heavy chain, one barrier less
      if (tid == 0) wcount = 0;
      return;
    }
    for (int e = tid >> 6; e < hc; e += SOLO_T / 64) {
      const int rel = hlist[e];
      const int gi = gidx(rel);
      const size_t g = bbase + gi;
      const float4 b4 = st_box(rel, g);
      const float bx[4] = {b4.x, b4.y, b4.z, b4.w};
      const float stale = st_stale(rel, g);
      const int t = POPS ? first_pop(stale, gi) : 0;
      float v = chain_wave(a, S, stale, st_begin(rel, g), bx, k + t);
      if ((tid & 63) == 0) {
        v = later_pops(rel, g, v, t, bx);
        if (!POPS) atomicOr(&ebits[rel >> 5], 1u << (rel & 31));
        fn(rel, v);
      }
    }
    __syncthreads();
#ifdef UDA_NMS_STATS
    if (tid == 0) { const unsigned long long tb2 = wall_clock64(); atomicAdd(&g_nms_dbg3[6], tb2 - tb1); atomicMax(&g_nms_dbg3[7], tb2 - tb1); }
#endif
    if (tid == 0) { wcount = 0; hcount = 0; }
  };

  int k = 0, rpar = 0;
  for (int step = 0; k < a.M; ++step) {
    // (the candidate index is rebuilt from an opaque base in every epoch: otherwise the per-candidate index words and
    // addresses of all IPT candidates are hoisted out of the epoch loop and spill)
    int ib = i0;
    asm volatile("" : "+v"(ib));
    unsigned long long* sslots = slots + (size_t)step * (1 + COOP_W) * bpi;
#ifdef UDA_NMS_STATS
    const unsigned long long t0 = wall_clock64();
#endif
    // ---- A. this block's best candidate by upper bound takes its exact score -> lower bound on the winner.  Scores
    // first, the index only for the best one: max over j of (ordered score, smaller j) is max over the keys
    unsigned long long bk = 0ull;
    {
      uint32_t bo = 0u;
      int bj = 0;
#pragma unroll
      for (int j = IPT - 1; j >= 0; --j) {
        const float u = U[j * SOLO_T + tid];
        if (st[j] != -INFINITY && u != -INFINITY) {
          const uint32_t o = ord32(u);
          if (o >= bo) { bo = o; bj = j; }      // descending j with >=: the smallest index wins a tie
        }
      }
      if (bo != 0u) bk = ((unsigned long long)bo << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)(ib + bj * istride));
    }
    bk = reg_max2(S, bk, rpar);

    if (bk != 0ull && tid < 64) {
      const int bi = (int)(0xFFFFFFFFu - (uint32_t)bk);
      const size_t g = bbase + bi;
      const int br = rel_of(bi);
      const int begin = st_begin(br, g);
      const float stale0 = st_stale(br, g);
      const float4 b4 = st_box(br, g);
      const float pb[4] = {b4.x, b4.y, b4.z, b4.w};
      const float score = chain_wave(a, S, stale0, begin, pb, k);
      if (tid == 0) {
        U[br] = score;
        atomicOr(&ebits[br >> 5], 1u << (br & 31));
        X.mine[0] = (score != -INFINITY) ? nms_key(score, bi) : 0ull;
      }
    }
    if (tid == 0) {
      if (bk == 0ull) X.mine[0] = 0ull;
      X.kcount = 0;
    }
#ifdef UDA_NMS_STATS
    __syncthreads();
    const unsigned long long t1 = wall_clock64();
#endif
    coop_exchange_top(sslots, blk, bpi, 1, wcfg, X, err, spin_max);
    // the step's bound: the weff-th largest block key (fewer blocks alive: the smallest one; none: 0 = everything is evaluated)
    int nz = 0;
#pragma unroll
    for (int r = 0; r < COOP_W; ++r) nz += X.top[r] != 0ull ? 1 : 0;
    int weff = wcfg < nz ? wcfg : nz;
    if (weff > a.M - k) weff = a.M - k;
    if (weff < 1) weff = 1;
    const unsigned long long bd = nz ? X.top[weff - 1] : 0ull;
#ifdef UDA_NMS_STATS
    const unsigned long long t2 = wall_clock64();
#endif
    // ---- B. exact scores of everything that can still reach the bound (few: a loop over a bit mask)
    unsigned long long ke = 0ull;
    unsigned need = 0u;
    {
      // key(u, i) >= bd  <=>  ord(u) > ord(bd)  or  (ord(u) == ord(bd) and ~i >= low word of bd)
      const uint32_t bdo = (uint32_t)(bd >> 32), bdl = (uint32_t)bd;
#pragma unroll
      for (int j = 0; j < IPT; ++j) {
        const float u = U[j * SOLO_T + tid];
        if (st[j] != -INFINITY && u != -INFINITY) {
          const uint32_t o = ord32(u);
          if (o > bdo || (o == bdo && 0xFFFFFFFFu - (uint32_t)(ib + j * istride) >= bdl)) need |= 1u << j;
        }
      }
    }
    run_balanced(need, k, 1, std::false_type{}, [&](int rel, float v) {
      if (v != -INFINITY) {
        const unsigned long long key = nms_key(v, gidx(rel));
        ke = key > ke ? key : ke;
        if (weff > 1 && key >= bd) {         // ranked below for this block's contribution
          const int pos = atomicAdd(&X.kcount, 1);
          if (pos < COOP_KL) klist[pos] = key;
        }
      }
    });
    ke = reg_max2(S, ke, rpar);
    // this block's weff best exact keys (wave 0; the list is complete: run_balanced ends with a barrier)
    if (tid < 64) {
      const int cnt = X.kcount;
      if (weff == 1 || cnt > COOP_KL) {
        if (tid < COOP_W) X.mine[tid] = tid == 0 ? ke : ((tid == 1 && weff > 1) ? COOP_OVF : 0ull);      // (words behind weff are not sent)
      } else {
        unsigned long long loc[COOP_KL / 64];
#pragma unroll
        for (int q = 0; q < COOP_KL / 64; ++q) loc[q] = (tid + 64 * q < cnt) ? klist[tid + 64 * q] : 0ull;
        for (int r = 0; r < weff; ++r) {
          unsigned long long m = loc[0];
#pragma unroll
          for (int q = 1; q < COOP_KL / 64; ++q) m = loc[q] > m ? loc[q] : m;
          m = wave_max_key(m);
          if (tid == 0) X.mine[r] = m;
#pragma unroll
          for (int q = 0; q < COOP_KL / 64; ++q)
            if (loc[q] == m) loc[q] = 0ull;
        }
      }
    }
#ifdef UDA_NMS_STATS
    const unsigned long long t3 = wall_clock64();
#endif
    coop_exchange_top(sslots + bpi, blk, bpi, weff, weff, X, err, spin_max);
#ifdef UDA_NMS_STATS
    const unsigned long long t4 = wall_clock64();
#endif
    const unsigned long long wk = X.top[0];
    if (wk == 0ull) break;                  // no live candidate in the whole problem (uniform over its blocks)
    // ---- C. the winners this step settles: boxes of the weff best keys, then the prefix that does not overlap
    if (tid < 4 * COOP_W) {
      const int t = tid >> 2;
      const unsigned long long key = X.top[t];
      if (t < weff && key != 0ull) S.sel[4 * (k + t) + (tid & 3)] = a.boxes[(bbase + (size_t)(0xFFFFFFFFu - (uint32_t)key)) * 4 + (tid & 3)];
    }
    // (one winner per step, the default: nothing reads this epoch's S.sel before the barrier inside run_balanced below)
    int nw = 1;
    if (weff > 1) {                         // (uniform)
      __syncthreads();
      if (tid == 0) {
        int nw_ = 1;
        if (!X.ovf && (a.soft || a.iou_thr >= 0.f)) {
          for (; nw_ < weff; ++nw_) {
            const unsigned long long key = X.top[nw_];
            if (key == 0ull || key < bd) break;
            bool ov = false;
            for (int p = 0; p < nw_; ++p) ov = ov || coop_strict_overlap(S.sel + 4 * (k + nw_), S.sel + 4 * (k + p));
            if (ov) break;
          }
        }
        X.nwin = nw_;
      }
      __syncthreads();
      nw = X.nwin;
    }
#ifdef UDA_NMS_STATS
    if (tid == 0 && blk == 0) atomicAdd(&g_nms_win[nw < 9 ? nw : 9], 1ull);
#endif
    // ---- D. the winners' records and the pops of the epochs k .. k + nw - 1
    unsigned pops = 0u;
    {
      unsigned wmask = 0u;                  // this thread's candidates among the winners
      for (int t = 0; t < nw; ++t) {
        const int widx = (int)(0xFFFFFFFFu - (uint32_t)X.top[t]);
        const int rel = rel_of(widx);
        if (rel >= 0 && rel < IPT * SOLO_T && (rel % SOLO_T) == tid) {
          const size_t o = (size_t)n * a.M + k + t;
          a.sel_idx[o] = widx;
          a.sel_score[o] = U[rel];          // exact in epoch k = exact in epoch k + t (no link of the step touches it)
          *(float4*)(a.sel_box + o * 4) = st_box(rel, bbase + widx);
          if constexpr (LST) lstale[rel] = -INFINITY; else a.stale[bbase + widx] = -INFINITY;
          wmask |= 1u << (rel / SOLO_T);
        }
      }
      // key(st, i) > low  <=>  ord(st) > ord(low)  or  (equal and ~i > low word): low = the last winner's key
      const unsigned long long low = X.top[nw - 1];
      const uint32_t wo = (uint32_t)(low >> 32), wl = (uint32_t)low;
#pragma unroll
      for (int j = 0; j < IPT; ++j) {
        if ((wmask >> j) & 1u) st[j] = -INFINITY;
        if (st[j] != -INFINITY) {
          const uint32_t o = ord32(st[j]);
          if (o > wo || (o == wo && 0xFFFFFFFFu - (uint32_t)(ib + j * istride) > wl)) pops |= 1u << j;
        }
      }
    }
    run_balanced(pops, k, nw, std::true_type{}, [&](int, float) {});      // (ends with a barrier: U[] of the popped candidates is complete)
#pragma unroll
    for (int j = 0; j < IPT; ++j)
      if ((pops >> j) & 1u) st[j] = U[j * SOLO_T + tid];
    k += nw;
    nsel = k;
    for (int f = tid; f < IPT * 32; f += SOLO_T) ebits[f] = 0u;      // "exact in this epoch" bits of the next step
    // (no barrier here: the next step touches ebits, U[] and the list counters only behind the barrier of its first block maximum)
#ifdef UDA_NMS_STATS
    if (tid == 0) {
      const unsigned long long t5 = wall_clock64();
      NMS_STAT(4, t1 - t0); NMS_STAT(5, t2 - t1); NMS_STAT(6, t3 - t2); NMS_STAT(7, t4 - t3);
      atomicAdd(&g_nms_dbg2[0], t5 - t4); atomicAdd(&g_nms_dbg2[1], 1ull);
      if (step < 128) { atomicMax(&g_nms_stepmax[step][0], t1 - t0); atomicMax(&g_nms_stepmax[step][1], t3 - t2); atomicMax(&g_nms_stepmax[step][2], t5 - t4); }
      {
        const unsigned long long te = t5 - t0;        // whole step of this block, 10 ns ticks
        int bkt = 0;
        while (bkt < 7 && te >= (2000ull << bkt)) ++bkt;     // < 20, 40, 80, 160, 320, 640, 1280 us, more
        atomicAdd(&g_nms_hist[bkt], 1ull);
        atomicAdd(&g_nms_hist[8 + bkt], te);
      }
    }
#endif
  }
  if (blk == 0 && tid == 0) a.nsel[n] = nsel;
}

// 64-bit words of exchange slots per problem (the caller's scratch: n_img x this; a launch uses M x 2 x its blocks per problem)
size_t nms_coop_slot_words(int M) { return (size_t)M * (1 + COOP_W) * COOP_MAX_BPI; }

template <int IPT>
constexpr size_t coop_lds_bytes() {
  return (size_t)IPT * SOLO_T * sizeof(float) + (size_t)(COOP_LIST + COOP_HEAVY) * sizeof(int) + (size_t)IPT * 32 * sizeof(unsigned) +
         (size_t)COOP_KL * sizeof(unsigned long long) + (coop_lds_state(IPT) ? (size_t)IPT * SOLO_T * 24 : 0);
}

// co-resident blocks of nms_coop_kernel<IPT> on the current device (occupancy x CUs; 0 = cannot run), cached per HIP device
template <int IPT>
static int coop_capacity(int dev, int* n_cu) {
  static int capacity_of[64], cus_of[64];
  static bool known[64];
  if (!known[dev]) {
    (void)hipGetLastError();          // an error left behind by an unrelated earlier call must not be read as ours
    // Blocks of this kernel a CU holds, from the kernel's own resources: 16 waves of <= 128 registers (launch bounds) are one
    // block per CU by registers, two below 65; LDS and the 2048-thread limit likewise.  (Not
    // hipOccupancyMaxActiveBlocksPerMultiprocessor: on ROCm 7.2 it answered 0 for this kernel - same registers, same LDS -
    // depending on which other kernels the process had run before, and 1 otherwise.)
    int per_cu = 0, cap = 0, n_cus = 0, coop = 0, lds_cu = 0;
    const hipError_t e0 = hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev);
    const hipError_t e1 = hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev);
    if (hipDeviceGetAttribute(&lds_cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) != hipSuccess) lds_cu = 0;
    hipError_t e2 = hipSuccess, e3 = hipSuccess;
    hipFuncAttributes fa{};
    cus_of[dev] = e0 == hipSuccess ? n_cus : 0;
    if (e0 == hipSuccess && e1 == hipSuccess && coop) {
      e2 = hipFuncSetAttribute((const void*)nms_coop_kernel<IPT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)coop_lds_bytes<IPT>());
      if (e2 == hipSuccess) e3 = hipFuncGetAttributes(&fa, (const void*)nms_coop_kernel<IPT>);
      if (e2 == hipSuccess && e3 == hipSuccess && fa.numRegs > 0 && fa.numRegs <= 128 && fa.maxThreadsPerBlock >= SOLO_T) {
        const size_t lds = fa.sharedSizeBytes + coop_lds_bytes<IPT>();       // the launch was accepted above: one block fits
        const int by_regs = (512 / ((fa.numRegs + 7) / 8 * 8)) * 4 / (SOLO_T / 64);
        const int by_lds = (size_t)lds_cu >= lds ? (int)((size_t)lds_cu / lds) : 1;
        per_cu = by_regs < by_lds ? by_regs : by_lds;
        if (per_cu > 2048 / SOLO_T) per_cu = 2048 / SOLO_T;
        if (per_cu < 1) per_cu = 1;
        cap = per_cu * n_cus;
      }
    }
    (void)hipGetLastError();
    if (getenv("UDA_NMS_DEBUG"))
      fprintf(stderr, "[uda] cooperative NMS capacity<%d> on device %d: %d CUs (%d B LDS each), cooperative %d, %d registers, %zu + %zu B LDS -> %d blocks per CU "
              "(%s / %s / %s / %s)\n", IPT, dev, n_cus, lds_cu, coop, fa.numRegs, fa.sharedSizeBytes, coop_lds_bytes<IPT>(), per_cu,
              hipGetErrorName(e0), hipGetErrorName(e1), hipGetErrorName(e2), hipGetErrorName(e3));
    const char* hook = getenv("UDA_NMS_COOP_CAP");     // test hook: a wrong capacity must end in the time-out path, not in a hang
    if (hook) cap = atoi(hook);
    capacity_of[dev] = cap;
    known[dev] = cap > 0 || hook != nullptr;          // a failed query is asked again next time
  }
  *n_cu = cus_of[dev];
  return capacity_of[dev];
}

template <int IPT>
static void coop_launch(const NmsArgs& a, const float* scores, unsigned long long* slots, int* err, int bpi, int per, unsigned spin_max, int wcfg, hipStream_t s) {
  for (int n0 = 0; n0 < a.n_img; n0 += per) {
    const int cnt = a.n_img - n0 < per ? a.n_img - n0 : per;
    hipLaunchKernelGGL((nms_coop_kernel<IPT>), dim3((unsigned)(bpi * cnt)), dim3(SOLO_T), coop_lds_bytes<IPT>(), s, a, scores, slots, err, bpi, n0, spin_max, wcfg);
  }
}

// 1 = launched; 0 = a problem outside this kernel's domain (segmented candidates, more than 128 outputs, nothing to do);
// -1 = wanted but NOT launched (the capacity query failed, the grid is not co-resident on this device, more candidates than
// COOP_MAX_BPI blocks hold, or the runtime refused): the caller falls back to the slower versions and counts it.
int launch_nms_coop(const NmsArgs& a, const float* scores, unsigned long long* slots, int* err, hipStream_t s) {
  if (a.segs != 1 || a.M > 128 || a.K < 1 || a.n_img <= 0) return 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
  // Candidates per thread (IPT).  An epoch is latency-bound - three scans of the block's candidates, two grid-wide steps -
  // so a small batch is spread over more, shorter blocks: the smallest IPT with which every block of the launch still has
  // a CU to itself; a batch that fills the device anyway takes 32 (fewest blocks per problem, most problems per launch).
  // UDA_NMS_COOP_IPT = 4 | 8 | 16 | 24 | 32 forces one.
  static int force = -1;
  if (force < 0) { const char* e = getenv("UDA_NMS_COOP_IPT"); force = e ? atoi(e) : 0; }
  const int ipts[5] = {4, 8, 16, 24, 32};
  int ipt = 0, bpi = 0, capacity = 0;
  for (int i = 0; i < 5 && !ipt; ++i) {
    const int t = ipts[i];
    if (force && t != force) continue;
    const int b = (a.K + t * SOLO_T - 1) / (t * SOLO_T);
    if (b > COOP_MAX_BPI) continue;
    int n_cu = 0;
    const int cap = t == 4 ? coop_capacity<4>(dev, &n_cu) : t == 8 ? coop_capacity<8>(dev, &n_cu) : t == 16 ? coop_capacity<16>(dev, &n_cu)
                  : t == 24 ? coop_capacity<24>(dev, &n_cu) : coop_capacity<32>(dev, &n_cu);
    if (force || t == 32 || ((long long)a.n_img * b <= n_cu && b <= cap)) { ipt = t; bpi = b; capacity = cap; }
  }
  // polls of an exchange slot before a block gives up (UDA_NMS_COOP_SPIN: debug knob, a tiny bound forces the time-out -> redo path)
  static long long spin_env = -1;
  if (spin_env < 0) { const char* e = getenv("UDA_NMS_COOP_SPIN"); spin_env = e ? atoll(e) : (1ll << 22); if (spin_env < 0) spin_env = 0; }
  const unsigned spin_max = (unsigned)(spin_env > 0xffffffffll ? 0xffffffffll : spin_env);
  static const bool dbg = getenv("UDA_NMS_DEBUG") != nullptr;
  if (!ipt || bpi > capacity) {
    if (dbg) fprintf(stderr, "[uda] cooperative NMS: %d blocks per problem > capacity %d\n", bpi, capacity);
    return -1;
  }
  const int per = capacity / bpi;          // problems per launch: more problems than the device holds run in consecutive grids
  // winners one grid-wide step may settle (UDA_NMS_WINNERS = 1 .. 8).  Default 1, measured: the second-best
  // candidate of an epoch overlaps the winner in 86 % of the steps under random-init weights (scores are spatially correlated:
  // the runners-up sit next to the maximum) and on every clustered score map, so extra winners are rarely certified while
  // the looser bound (the W-th largest of the blocks' offers) has every block evaluate more: 3.47 ms (1) / 3.66 (2) / 4.3 (4) /
  // 32 ms (8 of 8 blocks: the bound of the weakest block) per 32-image step
  static int wenv = -1;
  if (wenv < 0) { const char* e = getenv("UDA_NMS_WINNERS"); wenv = e ? atoi(e) : 0; }
  int wcfg = wenv > 0 ? wenv : 1;
  if (wcfg > COOP_W) wcfg = COOP_W;
  if (wcfg < 1) wcfg = 1;
  hipMemsetAsync(slots, 0, (size_t)a.n_img * a.M * (1 + COOP_W) * bpi * sizeof(unsigned long long), s);
  hipMemsetAsync(a.sel_idx, 0, (size_t)a.n_img * a.M * sizeof(int32_t), s);
  hipMemsetAsync(a.sel_score, 0, (size_t)a.n_img * a.M * sizeof(float), s);
  // An ordinary launch: the grid fits the device (checked above against the occupancy of this kernel), so every block
  // becomes resident as soon as whatever else runs on the device drains - other kernels never wait for this one - and the
  // bounded spin is the safety net.  (hipLaunchCooperativeKernel gives the same placement plus a formal check, but
  // rocprofv3's kernel tracing crashes at process exit after a cooperative launch, ROCm 7.2.)
  if (ipt == 4) coop_launch<4>(a, scores, slots, err, bpi, per, spin_max, wcfg, s);
  else if (ipt == 8) coop_launch<8>(a, scores, slots, err, bpi, per, spin_max, wcfg, s);
  else if (ipt == 16) coop_launch<16>(a, scores, slots, err, bpi, per, spin_max, wcfg, s);
  else if (ipt == 24) coop_launch<24>(a, scores, slots, err, bpi, per, spin_max, wcfg, s);
  else coop_launch<32>(a, scores, slots, err, bpi, per, spin_max, wcfg, s);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    if (dbg) fprintf(stderr, "[uda] cooperative NMS: launch refused: %s\n", hipGetErrorString(e));
    return -1;
  }
  if (dbg) fprintf(stderr, "[uda] cooperative NMS: %d problems x %d blocks of %d candidates per thread (capacity %d), up to %d winners per step\n", a.n_img, bpi, ipt, capacity, wcfg);
#ifdef UDA_NMS_STATS
  {
    hipStreamSynchronize(s);
    unsigned long long h[8];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_nms_dbg), sizeof(h));
    fprintf(stderr, "[uda] nms stats (cumulative): chains %llu links %llu chains>=16 %llu chains>=48 %llu\n", h[0], h[1], h[2], h[3]);
    unsigned long long h2[2];
    hipMemcpyFromSymbol(h2, HIP_SYMBOL(g_nms_dbg2), sizeof(h2));
    unsigned long long hh[16];
    hipMemcpyFromSymbol(hh, HIP_SYMBOL(g_nms_hist), sizeof(hh));
    fprintf(stderr, "[uda] nms epoch-time histogram (block-epochs : total ms) <20us %llu:%.1f <40 %llu:%.1f <80 %llu:%.1f <160 %llu:%.1f <320 %llu:%.1f <640 %llu:%.1f <1280 %llu:%.1f more %llu:%.1f\n",
            hh[0], hh[8] * 1e-5, hh[1], hh[9] * 1e-5, hh[2], hh[10] * 1e-5, hh[3], hh[11] * 1e-5, hh[4], hh[12] * 1e-5, hh[5], hh[13] * 1e-5, hh[6], hh[14] * 1e-5, hh[7], hh[15] * 1e-5);
    unsigned long long hw[10];
    hipMemcpyFromSymbol(hw, HIP_SYMBOL(g_nms_win), sizeof(hw));
    fprintf(stderr, "[uda] nms winners per grid-wide step (problems x steps, cumulative): 1:%llu 2:%llu 3:%llu 4:%llu 5:%llu 6:%llu 7:%llu 8:%llu\n",
            hw[1], hw[2], hw[3], hw[4], hw[5], hw[6], hw[7], hw[8]);
    unsigned long long h3[8];
    hipMemcpyFromSymbol(h3, HIP_SYMBOL(g_nms_dbg3), sizeof(h3));
    if (h2[1]) fprintf(stderr, "[uda] nms lists per block-phase: entries mean %.1f max %llu, heavy mean %.1f max %llu; stage1 mean %.1f max %llu, stage2 mean %.1f max %llu ticks\n",
                       (double)h3[0] / (2 * h2[1]), h3[1], (double)h3[2] / (2 * h2[1]), h3[3], (double)h3[4] / (2 * h2[1]), h3[5], (double)h3[6] / (2 * h2[1]), h3[7]);
    {
      static unsigned long long sm[128][4];
      hipMemcpyFromSymbol(sm, HIP_SYMBOL(g_nms_stepmax), sizeof(sm));
      double sa = 0, sb = 0, sc = 0; int ns = 0;
      for (int i = 0; i < 128; ++i) if (sm[i][0] | sm[i][1] | sm[i][2]) { sa += sm[i][0]; sb += sm[i][1]; sc += sm[i][2]; ++ns; }
      if (ns) fprintf(stderr, "[uda] nms slowest block per step (this launch and earlier ones, max), mean over %d steps in 10 ns ticks: A %.1f B %.1f C+D %.1f\n", ns, sa / ns, sb / ns, sc / ns);
      static unsigned long long zero[128][4];
      hipMemcpyToSymbol(HIP_SYMBOL(g_nms_stepmax), zero, sizeof(zero));
    }
    if (h2[1]) fprintf(stderr, "[uda] nms phases, mean per block-epoch in 10 ns ticks: A %.1f bar1 %.1f B %.1f bar2 %.1f C %.1f (%llu block-epochs)\n",
                       (double)h[4] / h2[1], (double)h[5] / h2[1], (double)h[6] / h2[1], (double)h[7] / h2[1], (double)h2[0] / h2[1], h2[1]);
  }
#endif
  return 1;
}

// ------------------------------------------------------------------------------------ NMS on a score prefix
// With the whole anchor set as candidates (184 k per image) almost none can ever be selected: a candidate is popped
// from the reference's heap only while its score is at least the score of the LAST box selected.  So the NMS is run
// (single-launch kernel above) on the candidates whose score is >= tau, tau chosen so that 2048..4096 candidates pass,
// kept in index order so that ties break exactly as in the full problem.  The result is the full problem's result iff
// no excluded candidate could have been popped before the loop ended:
//     max_out boxes selected : every excluded score <  the smallest selected (updated) score, or <= score_thresh
//     fewer selected         : every excluded score <= score_thresh (excluded candidates never enter the heap)
// prefix_check_kernel tests exactly that per image; images that fail are flagged and redone on the full set by the
// host (uda_api.hip finish_post), so the output is bit-identical either way.
constexpr int PFX_T = 1024;

struct PfxLds {
  unsigned hist[2048];
  int wsum[17];
  int bin, above, inbin;       // result of a digit selection
  int wcnt[PFX_T / 64];
  unsigned wex[PFX_T / 64];
};

// histogram of ((key >> shift) & (nb - 1)) over this block's candidates whose key matches `prefix` above hi_shift.
// Scores cluster (random-init: every score ~ 0.01), so a bin shared by >= 16 lanes of a wave is added once.
__device__ __forceinline__ void pfx_hist(const float* v, int K, uint32_t prefix, int hi_shift, int shift, int nb,
                                         unsigned* hist) {
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < nb; i += PFX_T) hist[i] = 0;
  __syncthreads();
  const int Kr = (K + 63) & ~63;
  for (int i = threadIdx.x; i < Kr; i += PFX_T) {
    uint32_t bin = 0xFFFFFFFFu;
    if (i < K) {
      const uint32_t key = ord32(v[i]);
      if (hi_shift >= 32 || (key >> hi_shift) == prefix) bin = (key >> shift) & (uint32_t)(nb - 1);
    }
    unsigned long long todo = __ballot(bin != 0xFFFFFFFFu);
    if (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const uint32_t b = (uint32_t)__shfl((int)bin, leader, 64);
      const unsigned long long same = __ballot(bin == b);
      const int ns = __popcll(same);
      if (ns >= 16) {
        if (lane == leader) atomicAdd(&hist[b], (unsigned)ns);
        if (bin == b) bin = 0xFFFFFFFFu;
      }
    }
    if (bin != 0xFFFFFFFFu) atomicAdd(&hist[bin], 1u);
  }
  __syncthreads();
}

// the bin (from the top) in which the running count reaches `need`; S.above = count in the bins above it
__device__ __forceinline__ void pfx_pick(PfxLds& S, int nb, int need) {
  // thread t owns bins nb-1-2t and nb-2-2t (descending order)
  const int hi = nb - 1 - 2 * (int)threadIdx.x, lo = hi - 1;
  const int ch = hi >= 0 ? (int)S.hist[hi] : 0, cl = lo >= 0 ? (int)S.hist[lo] : 0;
  int total;
  const int excl = block_excl_scan(ch + cl, S.wsum, &total);
  if (excl < need && need <= excl + ch + cl) {
    if (excl + ch >= need) { S.bin = hi; S.above = excl; S.inbin = ch; }
    else { S.bin = lo; S.above = excl + ch; S.inbin = cl; }
  }
  __syncthreads();
}

__global__ __launch_bounds__(PFX_T) void prefix_select_kernel(PrefixArgs a) {
  __shared__ PfxLds S;
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* v = a.scores + (size_t)n * a.K;
  const float* bx = a.boxes + (size_t)n * a.K * 4;
  int32_t* sidx = a.sub_idx + (size_t)n * a.Lcap;
  float* ssc = a.sub_scores + (size_t)n * a.Lcap;
  float4* sbx = (float4*)(a.sub_boxes + (size_t)n * a.Lcap * 4);

  // ---- tau: the coarsest key prefix with Lp <= #(key >= tau) <= Lcap
  uint32_t tau;
  bool ok = true;
  pfx_hist(v, a.K, 0u, 32, 21, 2048, S.hist);
  pfx_pick(S, 2048, a.Lp);
  const int b1 = S.bin, above1 = S.above;
  if (above1 + S.inbin <= a.Lcap) {
    tau = (uint32_t)b1 << 21;
  } else {
    __syncthreads();
    pfx_hist(v, a.K, (uint32_t)b1, 21, 10, 2048, S.hist);
    pfx_pick(S, 2048, a.Lp - above1);
    const int b2 = S.bin, above2 = above1 + S.above;
    if (above2 + S.inbin <= a.Lcap) {
      tau = ((uint32_t)b1 << 21) | ((uint32_t)b2 << 10);
    } else {
      __syncthreads();
      pfx_hist(v, a.K, ((uint32_t)b1 << 11) | (uint32_t)b2, 10, 0, 1024, S.hist);
      pfx_pick(S, 1024, a.Lp - above2);
      tau = ((uint32_t)b1 << 21) | ((uint32_t)b2 << 10) | (uint32_t)S.bin;
      ok = above2 + S.above + S.inbin <= a.Lcap;    // more exact ties than the prefix can hold: full problem
    }
  }
  __syncthreads();

  // ---- ordered compaction: wave w owns the contiguous slice [w * seg, (w + 1) * seg)
  const int seg = ((a.K + (PFX_T / 64) - 1) / (PFX_T / 64) + 63) & ~63;
  const int s0 = wave * seg, s1 = min(a.K, s0 + seg);
  int cnt = 0;
  uint32_t ex = 0u;                      // largest key among the excluded candidates (0 = none)
  if (ok) {
    for (int i = s0 + lane; i < s0 + seg; i += 64) {
      uint32_t key = 0u;
      if (i < s1) key = ord32(v[i]);
      const bool take = (i < s1) && key >= tau;
      cnt += __popcll(__ballot(take));
      if (i < s1 && !take) ex = key > ex ? key : ex;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const uint32_t o = (uint32_t)__shfl_xor((int)ex, off, 64);
    ex = o > ex ? o : ex;
  }
  if (lane == 0) { S.wcnt[wave] = cnt; S.wex[wave] = ex; }
  __syncthreads();
  int pos = 0, total = 0;
  uint32_t exall = 0u;
  for (int w = 0; w < PFX_T / 64; ++w) {
    if (w < wave) pos += S.wcnt[w];
    total += S.wcnt[w];
    exall = S.wex[w] > exall ? S.wex[w] : exall;
  }
  if (ok && total > 0) {
    for (int i = s0 + lane; i < s0 + seg; i += 64) {
      float sc = 0.f;
      bool take = false;
      if (i < s1) {
        sc = v[i];
        take = ord32(sc) >= tau;
      }
      const unsigned long long m = __ballot(take);
      if (m == 0ull) continue;
      if (take) {
        const int o = pos + __popcll(m & ((1ull << lane) - 1ull));
        sidx[o] = i;
        ssc[o] = sc;
        sbx[o] = *(const float4*)(bx + (size_t)i * 4);
      }
      pos += __popcll(m);
    }
  }
  if (!ok) total = 0;
  for (int o = total + tid; o < a.Lcap; o += PFX_T) {     // padding: dead candidates
    sidx[o] = 0;
    ssc[o] = -INFINITY;
    sbx[o] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (tid == 0) {
    a.excl_key[n] = exall;
    a.bad[n] = ok ? 0 : 1;
  }
}

void launch_prefix_select(const PrefixArgs& a, hipStream_t s) {
  if (a.n_img <= 0) return;
  hipLaunchKernelGGL(prefix_select_kernel, dim3(a.n_img), dim3(PFX_T), 0, s, a);
}

// maps the prefix problem's selections back to candidate indices and checks that the prefix was sufficient
__global__ __launch_bounds__(128) void prefix_check_kernel(PrefixCheckArgs a) {
  __shared__ uint32_t wmin[128];
  const int n = blockIdx.x, j = threadIdx.x;
  const int ns = a.sub_nsel[n];
  uint32_t key = 0xFFFFFFFFu;
  if (j < a.M) {
    const size_t o = (size_t)n * a.M + j;
    const float sc = a.sub_sel_score[o];
    a.sel_idx[o] = (j < ns) ? a.sub_idx[(size_t)n * a.Lcap + a.sub_sel_idx[o]] : 0;
    a.sel_score[o] = sc;
    if (j < ns) key = ord32(sc);
  }
  wmin[j] = key;
  __syncthreads();
  if (j == 0) {
    uint32_t m = 0xFFFFFFFFu;
    for (int t = 0; t < 128; ++t) m = wmin[t] < m ? wmin[t] : m;
    const uint32_t ek = a.excl_key[n];
    bool fine = true;
    if (ek != 0u) {                      // something was excluded
      const uint32_t eb = (ek & 0x80000000u) ? (ek ^ 0x80000000u) : ~ek;
      const float es = __uint_as_float(eb);
      fine = (es <= a.score_thr) || (ns == a.M && ek < m);
    }
    a.nsel[n] = ns;
    if (!fine) a.bad[n] = 1;
  }
}

void launch_prefix_check(const PrefixCheckArgs& a, hipStream_t s) {
  if (a.n_img <= 0) return;
  hipLaunchKernelGGL(prefix_check_kernel, dim3(a.n_img), dim3(128), 0, s, a);
}

// ------------------------------------------------------------------------------------ gather / pack
__global__ __launch_bounds__(128) void gather_kernel(GatherArgs a) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= a.n_img * a.M) return;
  const int n = gid / a.M, i = gid % a.M;
  const int valid = a.nsel[n];
  if (i == 0) a.out_valid[n] = valid;
  const int idx = (i < valid) ? a.sel_idx[gid] : 0;   // padded slots replicate candidate 0
  const size_t src = (size_t)n * a.K + idx;
  const float sc = a.scales ? a.scales[n] : 1.0f;
  float* ob = a.out_boxes + (size_t)gid * a.box_cols;
  const float hi[4] = {a.clip_h, a.clip_w, a.clip_h, a.clip_w};
  for (int k = 0; k < 4; ++k) {
    float v = a.boxes[src * 4 + k];
    if (a.clip) v = fminf(fmaxf(v, 0.0f), hi[k]);
    ob[k] = a.scales ? v * sc : v;
  }
  int col = 4;
  if (a.u_al) {
    for (int k = 0; k < 4; ++k) ob[col + k] = a.scales ? a.u_al[src * 4 + k] * sc : a.u_al[src * 4 + k];
    col += 4;
  }
  if (a.u_ep) {
    for (int k = 0; k < 4; ++k) ob[col + k] = a.scales ? a.u_ep[src * 4 + k] * sc : a.u_ep[src * 4 + k];
    col += 4;
  }
  a.out_scores[gid] = (i < valid) ? a.sel_score[gid] : 0.0f;
  float* oc = a.out_classes + (size_t)gid * a.cls_cols;
  oc[0] = (float)(a.classes[src] + 1);
  if (a.u_cls)
    for (int c = 0; c < a.ucls_cols; ++c) oc[1 + c] = a.u_cls[src * a.ucls_cols + c];
  if (a.out_logits)
    for (int c = 0; c < a.C; ++c) a.out_logits[(size_t)gid * a.C + c] = a.logits[src * a.C + c];
}

// per-class mode (postprocess.per_class_nms, :676-698): concatenate the per-class selections in
// class order, append M zero rows, take the top M by score (ties -> lower position), scale; no clip.
__global__ __launch_bounds__(256) void merge_per_class_kernel(MergeArgs a) {
  __shared__ unsigned long long wbest[4];
  __shared__ int offs[130];
  const int n = blockIdx.x;
  const int E_max = a.C * a.M + a.M;
  unsigned long long* keys = a.keys + (size_t)n * E_max;
  if (threadIdx.x == 0) {
    int run = 0;
    for (int c = 0; c < a.C; ++c) {
      if (c < 128) offs[c] = run;
      run += a.nsel[(size_t)n * a.C + c];
    }
    offs[128] = run;                         // number of real detections
  }
  __syncthreads();
  const int total_valid = offs[128];
  const int E = total_valid + a.M;
  // position e of entry (class c, rank j) in the concatenation; key = (score, earlier position first)
  for (int c = 0; c < a.C; ++c) {
    const size_t seg = (size_t)n * a.C + c;
    int start = 0;
    for (int cc = 0; cc < c; ++cc) start += a.nsel[(size_t)n * a.C + cc];
    for (int j = threadIdx.x; j < a.nsel[seg]; j += blockDim.x) {
      const int e = start + j;
      keys[e] = ((unsigned long long)ord32(a.sel_score[seg * a.M + j]) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)e);
    }
  }
  for (int j = threadIdx.x; j < a.M; j += blockDim.x) {
    const int e = total_valid + j;
    keys[e] = ((unsigned long long)ord32(0.0f) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)e);
  }
  __syncthreads();
  const float sc = a.scales ? a.scales[n] : 1.0f;
  for (int r = 0; r < a.M; ++r) {
    unsigned long long best = 0ull;
    for (int e = threadIdx.x; e < E; e += blockDim.x) best = keys[e] > best ? keys[e] : best;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned long long o = __shfl_xor(best, off, 64);
      best = o > best ? o : best;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) wbest[threadIdx.x >> 6] = best;
    __syncthreads();
    best = wbest[0];
    for (int w = 1; w < 4; ++w) best = wbest[w] > best ? wbest[w] : best;
    const int e = (int)(0xFFFFFFFFu - (uint32_t)best);
    if (threadIdx.x == 0) {
      keys[e] = 0ull;                                   // taken
      const size_t o = (size_t)n * a.M + r;
      if (e < total_valid) {
        int c = 0, start = 0;
        while (c + 1 < a.C && e >= start + a.nsel[(size_t)n * a.C + c]) { start += a.nsel[(size_t)n * a.C + c]; ++c; }
        const size_t seg = (size_t)n * a.C + c;
        const int j = e - start;
        const int idx = a.sel_idx[seg * a.M + j];
        const float* bx = a.boxes + ((size_t)n * a.K + idx) * 4;
        for (int q = 0; q < 4; ++q) a.out_boxes[o * 4 + q] = a.scales ? bx[q] * sc : bx[q];
        a.out_scores[o] = a.sel_score[seg * a.M + j];
        a.out_classes[o] = (float)(c + 1);
      } else {
        for (int q = 0; q < 4; ++q) a.out_boxes[o * 4 + q] = a.scales ? 0.0f * sc : 0.0f;
        a.out_scores[o] = 0.0f;
        a.out_classes[o] = 0.0f;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) a.out_valid[n] = total_valid < a.M ? total_valid : a.M;
}

void launch_merge_per_class(const MergeArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(merge_per_class_kernel, dim3(a.n_img), dim3(256), 0, s, a);
}

void launch_gather(const GatherArgs& a, hipStream_t s) {
  const int total = a.n_img * a.M;
  hipLaunchKernelGGL(gather_kernel, dim3((total + 127) / 128), dim3(128), 0, s, a);
}

// ------------------------------------------------------------------------------------ class probabilities
// What every caller does right after serve() (validate_model.py:159-166, infer_model.py:585-600,
// utils_class.py:36-41): p = stable_softmax(logits) and the entropy -sum p * log2(max(p, 1e-7)), float32.
__global__ __launch_bounds__(128) void probs_kernel(const float* logits, float* probs, float* entropy, int rows, int C) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float* x = logits + (size_t)r * C;
  float mx = x[0];
  for (int c = 1; c < C; ++c) mx = fmaxf(mx, x[c]);
  float sum = 0.f;
  for (int c = 0; c < C; ++c) sum += expf(x[c] - mx);
  float h = 0.f;
  for (int c = 0; c < C; ++c) {
    const float pc = expf(x[c] - mx) / sum;
    probs[(size_t)r * C + c] = pc;
    h += pc * log2f(fmaxf(pc, 1e-7f));
  }
  entropy[r] = -h;
}

void launch_probs(const float* logits, float* probs, float* entropy, int rows, int C, hipStream_t s) {
  if (rows <= 0) return;
  hipLaunchKernelGGL(probs_kernel, dim3((rows + 127) / 128), dim3(128), 0, s, logits, probs, entropy, rows, C);
}

// ------------------------------------------------------------------------------------ packed detection records
// One float32 row per (image, detection): [boxes (bc) | score | classes (cc) | logits (C, optional) | valid_len] - the record
// the multi-GPU layer gathers (dist.pack_detections builds the same row on the host).  Rows of images >= n are zero: padding
// of a ragged shard up to the collective's common size.  The gather then reads this buffer in place: the detections never
// cross PCIe before they have been collected.
__global__ __launch_bounds__(256) void pack_det_kernel(PackDetArgs a) {
  const int64_t total = (int64_t)a.rows_out * a.M * a.cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int col = (int)(i % a.cols);
    const int64_t r = i / a.cols;               // image * M + detection
    const int img = (int)(r / a.M);
    float v = 0.f;
    if (img < a.n) {
      if (col < a.bc) v = a.boxes[r * a.bc + col];
      else if (col == a.bc) v = a.scores[r];
      else if (col < a.bc + 1 + a.cc) v = a.classes[r * a.cc + (col - a.bc - 1)];
      else if (col < a.bc + 1 + a.cc + a.C) v = a.logits[r * a.C + (col - a.bc - 1 - a.cc)];
      else v = (float)a.valid[img];
    }
    a.out[i] = v;
  }
}

void launch_pack_det(const PackDetArgs& a, hipStream_t s) {
  const int64_t total = (int64_t)a.rows_out * a.M * a.cols;
  if (total <= 0) return;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_det_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
}

// ------------------------------------------------------------------------------------ box-uncertainty calibration
// CalibrateBoxUncert.calibrate_boxuncert (utils_box.py:404-524) on the selected rows: temperature scaling
// (uncert / T) or isotonic regression tables (sklearn IsotonicRegression(out_of_bounds="clip").predict = linear
// interpolation between the fitted thresholds in float64, clipped to their range), one table for all values,
// one per box coordinate, or one per (class, coordinate); the relative variant divides by the box height /
// width in float16 first (the reference passes dtype=np.float16) and multiplies back afterwards.
__device__ __forceinline__ float iso_predict(const double* xs, const double* ys, int m, float v) {
  if (m <= 0) return 0.f;
  double x = (double)v;
  if (x <= xs[0]) return (float)ys[0];
  if (x >= xs[m - 1]) return (float)ys[m - 1];
  int lo = 0, hi = m - 1;                  // xs[lo] <= x < xs[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (xs[mid] <= x) lo = mid; else hi = mid;
  }
  const double slope = (ys[hi] - ys[lo]) / (xs[hi] - xs[lo]);
  return (float)(ys[lo] + slope * (x - xs[lo]));
}

__global__ __launch_bounds__(128) void calib_kernel(CalibArgs a) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= a.rows) return;
  const float* row = a.boxes + (size_t)r * a.box_cols;
  const int cls = (int)a.classes[(size_t)r * a.cls_cols];
  const float hgt = row[2] - row[0], wid = row[3] - row[1];
  for (int j = 0; j < 4; ++j) {
    float u = row[a.col0 + j];
    if (isnan(u)) u = 0.f;                 // np.nan_to_num
    else if (isinf(u)) u = u > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;
    float o;
    if (a.mode == UDA_CALIB_TS_ALL) {
      o = u / a.temps[0];
    } else if (a.mode == UDA_CALIB_TS_PERCOO) {
      o = u / a.temps[j];
    } else {
      int t = 0;
      if (a.mode == UDA_CALIB_ISO_PERCOO) t = j;
      else if (a.mode == UDA_CALIB_ISO_PERCLSCOO) t = (cls - 1) * 4 + j;
      if (a.mode == UDA_CALIB_ISO_PERCLSCOO && (cls < 1 || cls > a.n_tables / 4)) {
        o = 0.f;                           // rows of no calibrated class stay zero (np.zeros_like)
      } else {
        const double* xs = a.xs + a.tab_off[t];
        const double* ys = a.ys + a.tab_off[t];
        const int m = a.tab_off[t + 1] - a.tab_off[t];
        if (a.relative) {
          const float norm = (j & 1) ? wid : hgt;
          float rel = 0.f;
          if (norm != 0.f) rel = __half2float(__float2half(__half2float(__float2half(u)) / __half2float(__float2half(norm))));
          o = iso_predict(xs, ys, m, rel) * norm;
        } else {
          o = iso_predict(xs, ys, m, u);
        }
      }
    }
    a.out[(size_t)r * 4 + j] = o;
  }
}

void launch_calib(const CalibArgs& a, hipStream_t s) {
  if (a.rows <= 0) return;
  hipLaunchKernelGGL(calib_kernel, dim3((a.rows + 127) / 128), dim3(128), 0, s, a);
}

// ------------------------------------------------------------------------------------ class calibration (row f2)
// CalibrateClass._perform_class_calib (utils_class.py:109-187) on the selected rows: temperature scaling of the logits
// (one temperature or one per class) followed by the stable softmax, or isotonic regression of the softmax
// probabilities (one table or one per class) followed by re-normalisation to sum 1; entropy -sum p log2(max(p, 1e-7)).
// With MC class uncertainty the reference draws 10 logit vectors from Normal(mean logits, MC std), calibrates each,
// and returns mean / population std of the calibrated probabilities and the entropy of the mean; the draws come from
// the build's Philox stream (philox_normal: counter (class, row, draw), tag 0x5A) instead of TFP's.
__device__ __forceinline__ void class_calib_one(const ClsCalibArgs& a, const float* z, float* p) {
  const int C = a.C;
  if (a.mode == UDA_CLS_TS) {
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) { p[c] = z[c] / a.temps[c]; mx = fmaxf(mx, p[c]); }
    float s = 0.f;
    for (int c = 0; c < C; ++c) { p[c] = expf(p[c] - mx); s = s + p[c]; }
    for (int c = 0; c < C; ++c) p[c] = p[c] / s;
    return;
  }
  float mx = -INFINITY;
  for (int c = 0; c < C; ++c) mx = fmaxf(mx, z[c]);
  float s = 0.f;
  for (int c = 0; c < C; ++c) { p[c] = expf(z[c] - mx); s = s + p[c]; }
  float t = 0.f;
  for (int c = 0; c < C; ++c) {
    const int tb = a.mode == UDA_CLS_ISO_ALL ? 0 : c;
    const int m = a.tab_off[tb + 1] - a.tab_off[tb];
    p[c] = iso_predict(a.xs + a.tab_off[tb], a.ys + a.tab_off[tb], m, p[c] / s);
    t = t + p[c];
  }
  for (int c = 0; c < C; ++c) p[c] = p[c] / t;
}

constexpr int CLS_MAX = 128;
__global__ __launch_bounds__(64) void class_calib_kernel(ClsCalibArgs a) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= a.rows) return;
  const int C = a.C;
  const float* lg = a.logits + (size_t)r * C;
  float z[CLS_MAX], p[CLS_MAX];
  float* out = a.probs + (size_t)r * C;
  if (a.draws <= 0) {
    for (int c = 0; c < C; ++c) z[c] = lg[c];
    class_calib_one(a, z, p);
    for (int c = 0; c < C; ++c) out[c] = p[c];
  } else {
    const float* sg = a.classes + (size_t)r * a.cls_cols + 1;
    float mean[CLS_MAX], sq[CLS_MAX];
    for (int c = 0; c < C; ++c) { mean[c] = 0.f; sq[c] = 0.f; }
    for (int d = 0; d < a.draws; ++d) {
      for (int c = 0; c < C; ++c) z[c] = lg[c] + sg[c] * (float)philox_normal(a.seed, (uint32_t)c, (uint32_t)r, (uint32_t)d, 0x5Au);
      class_calib_one(a, z, p);
      for (int c = 0; c < C; ++c) mean[c] = mean[c] + p[c];
    }
    for (int c = 0; c < C; ++c) mean[c] = mean[c] / (float)a.draws;
    for (int d = 0; d < a.draws; ++d) {           // second pass over the same draws: population std around the mean
      for (int c = 0; c < C; ++c) z[c] = lg[c] + sg[c] * (float)philox_normal(a.seed, (uint32_t)c, (uint32_t)r, (uint32_t)d, 0x5Au);
      class_calib_one(a, z, p);
      for (int c = 0; c < C; ++c) { const float dl = p[c] - mean[c]; sq[c] = sq[c] + dl * dl; }
    }
    for (int c = 0; c < C; ++c) {
      out[c] = mean[c];
      if (a.uncert) a.uncert[(size_t)r * C + c] = sqrtf(sq[c] / (float)a.draws);
      p[c] = mean[c];
    }
  }
  float h = 0.f;
  for (int c = 0; c < C; ++c) h = h + p[c] * log2f(fmaxf(p[c], 1e-7f));
  a.entropy[r] = -h;
}

void launch_class_calib(const ClsCalibArgs& a, hipStream_t s) {
  if (a.rows <= 0) return;
  hipLaunchKernelGGL(class_calib_kernel, dim3((a.rows + 63) / 64), dim3(64), 0, s, a);
}

// ------------------------------------------------------------------------------------ numpy NMS family (row a18)
// nms_np.hard_nms / diou_nms / soft_nms (reference src/nms_np.py:30-192): boxes x1,y1,x2,y2 with the +1 pixel
// convention.  One block per problem (a class of an image); T = double for the float64 arrays the family functions
// are called with directly, float for per_class_nms (float32 boxes / scores make every intermediate float32).
// hard / diou: the dets arrive sorted by score (descending, host side = argsort()[::-1]); box i is kept iff no kept
// box before it suppresses it - n greedy steps, the suppression test of a step spread over the block.
// soft (gaussian / linear): every step takes the arg-max of the current scores, emits it, rescales the others
// (exp(-iou^2 / sigma) or 1 - iou above the threshold) and drops those that fall below score_thresh.
template <typename T>
__device__ __forceinline__ T np_iou(const T* a, T area_a, const T* b, T area_b) {
  const T w = max((T)0, min(a[2], b[2]) - max(a[0], b[0]) + (T)1);
  const T h = max((T)0, min(a[3], b[3]) - max(a[1], b[1]) + (T)1);
  const T inter = w * h;
  return inter / (area_a + area_b - inter);
}

template <typename T>
__global__ __launch_bounds__(256) void nmsnp_kernel(NmsNpArgs<T> a) {
  __shared__ T red_v[256];
  __shared__ int red_i[256];
  __shared__ int s_cur, s_nout;
  const int pb = blockIdx.x;
  const int off = a.off[pb], n = a.off[pb + 1] - off;
  const T* d = a.dets + (size_t)off * 5;
  T* sc = a.score + off;          // working scores (soft) 
  int* st = a.state + off;        // 0 alive, 1 removed / emitted
  T* out = a.out + (size_t)off * 5;
  const int tid = threadIdx.x;
  for (int i = tid; i < n; i += 256) { sc[i] = d[i * 5 + 4]; st[i] = 0; }
  if (tid == 0) s_nout = 0;
  __syncthreads();
  if (a.method <= 1) {            // hard (0) / diou (1): input sorted by score, descending
    for (int i = 0; i < n; ++i) {
      if (st[i]) continue;        // uniform: st[] is only written before a barrier
      const T* bi = d + i * 5;
      const T ai = (bi[2] - bi[0] + (T)1) * (bi[3] - bi[1] + (T)1);
      for (int j = i + 1 + tid; j < n; j += 256) {
        if (st[j]) continue;
        const T* bj = d + j * 5;
        const T aj = (bj[2] - bj[0] + (T)1) * (bj[3] - bj[1] + (T)1);
        T v = np_iou(bi, ai, bj, aj);
        if (a.method == 1) {
          const T ex1 = min(bi[0], bj[0]), ex2 = max(bi[2], bj[2]), ey1 = min(bi[1], bj[1]), ey2 = max(bi[3], bj[3]);
          const T diag = (ex2 - ex1) * (ex2 - ex1) + (ey2 - ey1) * (ey2 - ey1);
          const T cxi = (bi[0] + bi[2]) / (T)2, cyi = (bi[1] + bi[3]) / (T)2;
          const T cxj = (bj[0] + bj[2]) / (T)2, cyj = (bj[1] + bj[3]) / (T)2;
          const T dist = (cxi - cxj) * (cxi - cxj) + (cyi - cyj) * (cyi - cyj);
          v = v - dist / (diag + (T)1e-10);
        }
        if (!(v <= a.iou_thr)) st[j] = 1;
      }
      if (tid == 0) {
        T* o = out + (size_t)s_nout * 5;
        for (int k = 0; k < 5; ++k) o[k] = bi[k];
        s_nout = s_nout + 1;
      }
      __syncthreads();
    }
  } else {                         // soft: gaussian (2) / linear (3)
    for (int step = 0; step < n; ++step) {
      T bv = -INFINITY;
      int bidx = -1;
      for (int i = tid; i < n; i += 256)
        if (!st[i] && (sc[i] > bv || bidx < 0)) { bv = sc[i]; bidx = i; }     // first maximum of this thread's stride
      red_v[tid] = bv; red_i[tid] = bidx;
      __syncthreads();
      for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if (tid < s2) {
          const int oi = red_i[tid + s2];
          const T ov = red_v[tid + s2];
          const int mi = red_i[tid];
          if (oi >= 0 && (mi < 0 || ov > red_v[tid] || (ov == red_v[tid] && oi < mi))) { red_v[tid] = ov; red_i[tid] = oi; }
        }
        __syncthreads();
      }
      if (tid == 0) s_cur = red_i[0];
      __syncthreads();
      const int m = s_cur;
      if (m < 0) break;
      const T* bm = d + m * 5;
      const T am = (bm[2] - bm[0] + (T)1) * (bm[3] - bm[1] + (T)1);
      if (tid == 0) {
        T* o = out + (size_t)s_nout * 5;
        for (int k = 0; k < 4; ++k) o[k] = bm[k];
        o[4] = sc[m];
        s_nout = s_nout + 1;
        st[m] = 1;
      }
      __syncthreads();
      for (int j = tid; j < n; j += 256) {
        if (st[j]) continue;
        const T* bj = d + j * 5;
        const T aj = (bj[2] - bj[0] + (T)1) * (bj[3] - bj[1] + (T)1);
        const T v = np_iou(bm, am, bj, aj);
        T wgt;
        if (a.method == 2) wgt = (T)exp(-(v * v) / a.sigma);
        else wgt = (v > a.iou_thr) ? (T)1 - v : (T)1;
        const T ns = sc[j] * wgt;
        sc[j] = ns;
        if (!(ns >= a.score_thr)) st[j] = 1;
      }
      __syncthreads();
    }
  }
  if (tid == 0) a.n_out[pb] = s_nout;
}

template <typename T>
void launch_nmsnp(const NmsNpArgs<T>& a, int n_problems, hipStream_t s) {
  if (n_problems > 0) hipLaunchKernelGGL(nmsnp_kernel<T>, dim3(n_problems), dim3(256), 0, s, a);
}
template void launch_nmsnp<float>(const NmsNpArgs<float>&, int, hipStream_t);
template void launch_nmsnp<double>(const NmsNpArgs<double>&, int, hipStream_t);

}  // namespace uda
