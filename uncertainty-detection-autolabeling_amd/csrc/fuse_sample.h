// BiFPN fast-normalised fusion of up to three resampled inputs at one output position (efficientdet_keras.py:90-136,
// utils_keras / ResampleFeatureMap): shared by fuse_kernel (kernels_conv.hip), which writes the fused tensor, and by
// sepf_kernel (kernels_sep.hip), which computes it on the fly for the tile of the separable conv that follows - one
// definition, so that both produce the same bits.
#pragma once
#include <math.h>

#include "uda_internal.h"

namespace uda {

__device__ __forceinline__ float4 fuse_sample(const FuseArgs& a, int i, int b, int y, int x, int c4) {
  const int bi = b / a.in_div[i];
  const float* base = a.in[i] + (size_t)bi * a.Hi[i] * a.Wi[i] * a.C + c4 * 4;
  if (a.mode[i] == UDA_RS_NONE) {
    return *(const float4*)(base + ((size_t)y * a.Wi[i] + x) * a.C);
  }
  if (a.mode[i] == UDA_RS_NEAREST_UP) {
    int sy = (int)floorf((float)y * a.sy[i]);
    int sx = (int)floorf((float)x * a.sx[i]);
    sy = min(sy, a.Hi[i] - 1);
    sx = min(sx, a.Wi[i] - 1);
    return *(const float4*)(base + ((size_t)sy * a.Wi[i] + sx) * a.C);
  }
  // max pool, TF SAME: padded taps never win
  float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  const int y0 = y * a.ps[i] - a.ppt[i], x0 = x * a.ps[i] - a.ppl[i];
  for (int ky = 0; ky < a.pk[i]; ++ky) {
    const int iy = y0 + ky;
    if (iy < 0 || iy >= a.Hi[i]) continue;
    for (int kx = 0; kx < a.pk[i]; ++kx) {
      const int ix = x0 + kx;
      if (ix < 0 || ix >= a.Wi[i]) continue;
      const float4 v = *(const float4*)(base + ((size_t)iy * a.Wi[i] + ix) * a.C);
      m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
    }
  }
  return m;
}

// x * sigmoid(x) with the hardware exp2 / rcp (1-ulp) instead of an IEEE divide
__device__ __forceinline__ float fuse_swish(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// act(sum_i wgt[i] * resample_i(in[i])) for channels 4 c4 .. 4 c4 + 3 of output position (b, y, x)
__device__ __forceinline__ float4 fuse_value(const FuseArgs& a, int b, int y, int x, int c4) {
  float4 s = fuse_sample(a, 0, b, y, x, c4);
  s.x *= a.wgt[0]; s.y *= a.wgt[0]; s.z *= a.wgt[0]; s.w *= a.wgt[0];
  for (int i = 1; i < a.n_in; ++i) {
    const float4 v = fuse_sample(a, i, b, y, x, c4);
    s.x = fmaf(v.x, a.wgt[i], s.x);
    s.y = fmaf(v.y, a.wgt[i], s.y);
    s.z = fmaf(v.z, a.wgt[i], s.z);
    s.w = fmaf(v.w, a.wgt[i], s.w);
  }
  if (a.act == UDA_ACT_SWISH) {
    s.x = fuse_swish(s.x); s.y = fuse_swish(s.y); s.z = fuse_swish(s.z); s.w = fuse_swish(s.w);
  } else if (a.act >= UDA_ACT_RELU) {
    s.x = act_relu_family(s.x, a.act); s.y = act_relu_family(s.y, a.act); s.z = act_relu_family(s.z, a.act); s.w = act_relu_family(s.w, a.act);
  }
  return s;
}

}  // namespace uda
