// Conv-stack kernels of the EfficientDet forward pass for gfx950 (MI355X), NHWC float32.
//
//   stem_kernel    3x3 stride-2 conv (Cin=3) + BN + swish           (reference backbone/efficientnet_model.py:588-612)
//   pw_kernel      1x1 conv as an f32 MFMA GEMM with fused SE gate (input), bias, BN, swish,
//                  MC-dropout keep-scale and residual add           (:358-373,403-418,446-486; efficientdet_keras.py:207-227,313-319)
//   dw_kernel      depthwise kxk / stride s / TF-SAME + BN + swish + MC-dropout + SE partial sums
//                  (:376-391,459-464; efficientdet_keras.py:207-227)
//   se_kernel      squeeze-excite gate: mean -> fc -> swish -> fc -> sigmoid   (:219-232)
//   fuse_kernel    BiFPN fast-normalised fusion over resampled inputs (identity / nearest-up /
//                  max-pool) + swish; also the stand-alone max-pool for P6/P7
//                  (efficientdet_keras.py:86-127,229-231,280-311,321-350)
//   philox_kernel  MC-dropout keep-scales for every (site, sample row, channel)
//
// Wavefront = 64.  The pointwise GEMM uses v_mfma_f32_32x32x2_f32 (exact f32, runs at the
// f32 vector rate); everything else is HBM-bound streaming with 16-byte accesses.
#include "uda_internal.h"

namespace uda {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float swishf(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// ------------------------------------------------------------------------------------ stem
// One thread = one output pixel x 4 output channels; the 27 x Co weights sit in LDS.
__global__ __launch_bounds__(256) void stem_kernel(StemArgs a) {
  extern __shared__ float wl[];  // [27][Co]
  for (int i = threadIdx.x; i < 27 * a.Co; i += blockDim.x) wl[i] = a.w[i];
  __syncthreads();
  const int co4n = a.Co >> 2;
  const int64_t total = (int64_t)a.rows * a.Ho * a.Wo * co4n;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  const int c4 = (int)(gid % co4n);
  int64_t p = gid / co4n;
  const int x = (int)(p % a.Wo);
  p /= a.Wo;
  const int y = (int)(p % a.Ho);
  const int b = (int)(p / a.Ho);
  const float* inb = a.in + (size_t)b * a.H * a.W * 3;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = y * 2 - a.pad_t + ky;
    if (iy < 0 || iy >= a.H) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = x * 2 - a.pad_l + kx;
      if (ix < 0 || ix >= a.W) continue;
      const float* px = inb + ((size_t)iy * a.W + ix) * 3;
#pragma unroll
      for (int ci = 0; ci < 3; ++ci) {
        const float v = px[ci];
        const float4 w = *(const float4*)(wl + ((ky * 3 + kx) * 3 + ci) * a.Co + c4 * 4);
        acc.x = fmaf(v, w.x, acc.x);
        acc.y = fmaf(v, w.y, acc.y);
        acc.z = fmaf(v, w.z, acc.z);
        acc.w = fmaf(v, w.w, acc.w);
      }
    }
  }
  const float4 s = *(const float4*)(a.bn_scale + c4 * 4);
  const float4 t = *(const float4*)(a.bn_shift + c4 * 4);
  float4 o;
  o.x = swishf(fmaf(acc.x, s.x, t.x));
  o.y = swishf(fmaf(acc.y, s.y, t.y));
  o.z = swishf(fmaf(acc.z, s.z, t.z));
  o.w = swishf(fmaf(acc.w, s.w, t.w));
  *(float4*)(a.out + (size_t)gid * 4) = o;
}

void launch_stem(const StemArgs& a, hipStream_t s) {
  const int64_t total = (int64_t)a.rows * a.Ho * a.Wo * (a.Co >> 2);
  const int grid = (int)((total + 255) / 256);
  hipLaunchKernelGGL(stem_kernel, dim3(grid), dim3(256), 27 * a.Co * sizeof(float), s, a);
}

// ------------------------------------------------------------------------------------ pointwise
// Block tile 128 pixels x (32*NT) output channels, K staged 32 at a time through LDS.
// 4 waves; wave w owns pixel rows [32w, 32w+32) and all NT 32-wide column tiles.
// A is staged transposed ([k][m], row stride 129 -> conflict-free writes and reads) so that
// lane (i = lane&31, h = lane>>5) reads A[m = i][k = kk + h] with consecutive lanes on
// consecutive banks; B is [k][n] straight from the TF kernel layout [Cin][Cout].
template <int NT>
__global__ __launch_bounds__(256) void pw_kernel(PwArgs a) {
  constexpr int BM = 128, BK = 32, BN = 32 * NT;
  __shared__ float As[BK][BM + 1];
  __shared__ float Bs[BK][BN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, b_in = b / a.in_div;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const float* A = a.in + (size_t)b_in * a.HW * a.Cin;
  const float* se = a.se ? a.se + (size_t)b_in * a.Cin : nullptr;

  f32x16 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

  for (int k0 = 0; k0 < a.Cin; k0 += BK) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = tid + 256 * i;
      const int m = f >> 3, kq = f & 7;
      const int k = k0 + 4 * kq;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m0 + m < a.HW && k < a.Cin) {
        v = *(const float4*)(A + (size_t)(m0 + m) * a.Cin + k);
        if (se) {
          const float4 g = *(const float4*)(se + k);
          v.x *= g.x; v.y *= g.y; v.z *= g.z; v.w *= g.w;
        }
      }
      As[4 * kq + 0][m] = v.x;
      As[4 * kq + 1][m] = v.y;
      As[4 * kq + 2][m] = v.z;
      As[4 * kq + 3][m] = v.w;
    }
#pragma unroll
    for (int i = 0; i < (BK * BN) / 256; ++i) {
      const int f = tid + 256 * i;
      const int kk = f / BN, n = f % BN;
      const int k = k0 + kk, col = n0 + n;
      Bs[kk][n] = (k < a.Cin && col < a.Cout) ? a.w[(size_t)k * a.Cout + col] : 0.f;
    }
    __syncthreads();
    const int kend = min(BK, a.Cin - k0);
    for (int kk = 0; kk < kend; kk += 2) {
      const float av = As[kk + lh][wave * 32 + li];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const float bv = Bs[kk + lh][n * 32 + li];
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[n], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // epilogue: lane holds column (li) of 16 rows: row = (r&3) + 8*(r>>2) + 4*lh
  const size_t out_base = (size_t)b * a.HW;
  const size_t res_base = a.res ? (size_t)(b / a.res_div) * a.HW : 0;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int col = n0 + n * 32 + li;
    if (col >= a.Cout) continue;
    const float bias = a.bias ? a.bias[col] : 0.f;
    const float sc = a.bn_scale ? a.bn_scale[col] : 1.f;
    const float sh = a.bn_scale ? a.bn_shift[col] : 0.f;
    const float mk = a.mask ? a.mask[(size_t)b * a.Cout + col] : 1.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m >= a.HW) continue;
      float v = acc[n][r] + bias;
      v = fmaf(v, sc, sh);
      if (a.act == UDA_ACT_SWISH) v = swishf(v);
      v *= mk;
      if (a.res) v += a.res[(res_base + m) * a.Cout + col];
      a.out[(out_base + m) * a.Cout + col] = v;
    }
  }
}

void launch_pw(const PwArgs& a, int rows, hipStream_t s) {
  const int gx = (a.HW + 127) / 128;
  if (a.Cout <= 32) {
    hipLaunchKernelGGL(pw_kernel<1>, dim3(gx, 1, rows), dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL(pw_kernel<2>, dim3(gx, (a.Cout + 63) / 64, rows), dim3(256), 0, s, a);
  }
}

// ------------------------------------------------------------------------------------ depthwise
// Thread = 4 channels x XB consecutive output columns x DW_ROWS output rows.  Consecutive
// threads walk the channel quads of a pixel first (16-byte loads, fully coalesced NHWC),
// then the x-groups.  The per-tile channel sums for squeeze-excite are reduced in a fixed
// order (deterministic; no float atomics).
constexpr int DW_ROWS = 8;  // output rows per block: 8x fewer SE tile sums, weights stay hot in L1

template <int K, int S, int XB>
__global__ __launch_bounds__(256) void dw_kernel(DwArgs a) {
  __shared__ float4 red[256];
  const int tid = threadIdx.x;
  const int c4l = tid % a.tc, pg = tid / a.tc;
  const int b = blockIdx.z / a.n_cchunk, cc = blockIdx.z % a.n_cchunk;
  const int c4 = cc * a.tc + c4l;
  const int C4 = a.C >> 2;
  const int x0 = (blockIdx.x * a.pxb + pg) * XB;
  const bool active = (pg < a.pxb) && (c4 < C4) && (x0 < a.Wo);
  constexpr int NCOL = (XB - 1) * S + K;
  float4 ssum = make_float4(0.f, 0.f, 0.f, 0.f);

  if (active) {
    const int b_in = b / a.in_div;
    const float* inb = a.in + (size_t)b_in * a.H * a.W * a.C + c4 * 4;
    const float* wb = a.w + c4 * 4;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bn_scale) {
      sc = *(const float4*)(a.bn_scale + c4 * 4);
      sh = *(const float4*)(a.bn_shift + c4 * 4);
    }
    float4 mk = make_float4(1.f, 1.f, 1.f, 1.f);
    if (a.mask) mk = *(const float4*)(a.mask + (size_t)b * a.C + c4 * 4);
    const int ix0 = x0 * S - a.pad_l;
    for (int r = 0; r < DW_ROWS; ++r) {
      const int y = blockIdx.y * DW_ROWS + r;
      if (y >= a.Ho) break;
      float4 acc[XB];
#pragma unroll
      for (int o = 0; o < XB; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
      const int iy0 = y * S - a.pad_t;
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        const int iy = iy0 + ky;
        if (iy < 0 || iy >= a.H) continue;
        const float* rowp = inb + (size_t)iy * a.W * a.C;
        float4 col[NCOL];
#pragma unroll
        for (int j = 0; j < NCOL; ++j) {
          const int ix = ix0 + j;
          col[j] = (ix >= 0 && ix < a.W) ? *(const float4*)(rowp + (size_t)ix * a.C)
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const float4 w = *(const float4*)(wb + (size_t)(ky * K + kx) * a.C);
#pragma unroll
          for (int o = 0; o < XB; ++o) {
            const float4 v = col[o * S + kx];
            acc[o].x = fmaf(v.x, w.x, acc[o].x);
            acc[o].y = fmaf(v.y, w.y, acc[o].y);
            acc[o].z = fmaf(v.z, w.z, acc[o].z);
            acc[o].w = fmaf(v.w, w.w, acc[o].w);
          }
        }
      }
      float* outp = a.out + (((size_t)b * a.Ho + y) * a.Wo) * a.C + c4 * 4;
#pragma unroll
      for (int o = 0; o < XB; ++o) {
        const int x = x0 + o;
        if (x >= a.Wo) break;
        float4 v;
        v.x = fmaf(acc[o].x, sc.x, sh.x);
        v.y = fmaf(acc[o].y, sc.y, sh.y);
        v.z = fmaf(acc[o].z, sc.z, sh.z);
        v.w = fmaf(acc[o].w, sc.w, sh.w);
        if (a.act == UDA_ACT_SWISH) {
          v.x = swishf(v.x); v.y = swishf(v.y); v.z = swishf(v.z); v.w = swishf(v.w);
        }
        v.x *= mk.x; v.y *= mk.y; v.z *= mk.z; v.w *= mk.w;
        *(float4*)(outp + (size_t)x * a.C) = v;
        ssum.x += v.x; ssum.y += v.y; ssum.z += v.z; ssum.w += v.w;
      }
    }
  }
  if (a.se_partial) {
    red[tid] = ssum;
    __syncthreads();
    if (pg == 0 && c4 < C4) {
      float4 t = red[c4l];
      for (int g = 1; g < a.pxb; ++g) {
        const float4 u = red[g * a.tc + c4l];
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      }
      const size_t tile = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
      *(float4*)(a.se_partial + ((size_t)b * a.n_tiles + tile) * a.C + c4 * 4) = t;
    }
  }
}

void dw_geometry(int C, int Wo, int stride, int* tc, int* pxb, int* n_cchunk, int* grid_x, int* xb) {
  const int C4 = C / 4;
  const int ncc = (C4 + 255) / 256;
  const int t = (C4 + ncc - 1) / ncc;
  int p = 256 / t;
  if (p < 1) p = 1;
  const int x = (stride == 1) ? 4 : 2;
  // do not spread one block over more columns than the row has
  const int need = (Wo + x - 1) / x;
  if (p > need) p = need;
  *tc = t;
  *pxb = p;
  *n_cchunk = ncc;
  *xb = x;
  *grid_x = (Wo + p * x - 1) / (p * x);
}

int dw_tiles(int C, int Ho, int Wo, int stride) {
  int tc, pxb, ncc, gx, xb;
  dw_geometry(C, Wo, stride, &tc, &pxb, &ncc, &gx, &xb);
  return ((Ho + DW_ROWS - 1) / DW_ROWS) * gx;
}

void launch_dw(DwArgs a, int rows, int k, int stride, hipStream_t s) {
  int gx, xb;
  dw_geometry(a.C, a.Wo, stride, &a.tc, &a.pxb, &a.n_cchunk, &gx, &xb);
  const int gy = (a.Ho + DW_ROWS - 1) / DW_ROWS;
  a.n_tiles = gy * gx;
  int threads = a.tc * a.pxb;
  threads = (threads + 63) / 64 * 64;
  const dim3 grid(gx, gy, rows * a.n_cchunk), block(threads);
  if (k == 3 && stride == 1) hipLaunchKernelGGL((dw_kernel<3, 1, 4>), grid, block, 0, s, a);
  else if (k == 3 && stride == 2) hipLaunchKernelGGL((dw_kernel<3, 2, 2>), grid, block, 0, s, a);
  else if (k == 5 && stride == 1) hipLaunchKernelGGL((dw_kernel<5, 1, 4>), grid, block, 0, s, a);
  else if (k == 5 && stride == 2) hipLaunchKernelGGL((dw_kernel<5, 2, 2>), grid, block, 0, s, a);
}

// ------------------------------------------------------------------------------------ squeeze-excite
// One block per sample row: channel means from the depthwise kernel's tile sums (fixed
// order), then the two tiny dense layers.
__global__ __launch_bounds__(256) void se_kernel(SeArgs a) {
  extern __shared__ float sm[];  // red[256 float4] | mean[C] | mid[mid]
  float4* red = (float4*)sm;
  float* mean = sm + 1024;
  float* mid = mean + a.C;
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const float* part = a.partial + (size_t)b * a.n_tiles * a.C;
  const int C4 = a.C >> 2;
  // channel sums: thread = (channel quad, tile group); groups are combined in a fixed order
  for (int cbase = 0; cbase < C4; cbase += 256) {
    const int cw = min(256, C4 - cbase);
    const int G = 256 / cw;
    const int c4 = cbase + tid % cw, g = tid / cw;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g < G) {
      for (int t = g; t < a.n_tiles; t += G) {
        const float4 v = *(const float4*)(part + (size_t)t * a.C + c4 * 4);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
    }
    red[tid] = s;
    __syncthreads();
    if (g == 0) {
      for (int g2 = 1; g2 < G; ++g2) {
        const float4 v = red[g2 * cw + tid];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      mean[c4 * 4 + 0] = s.x * a.inv_hw;
      mean[c4 * 4 + 1] = s.y * a.inv_hw;
      mean[c4 * 4 + 2] = s.z * a.inv_hw;
      mean[c4 * 4 + 3] = s.w * a.inv_hw;
    }
    __syncthreads();
  }
  for (int j = tid; j < a.mid; j += blockDim.x) {
    float s = 0.f;
    for (int c = 0; c < a.C; ++c) s = fmaf(mean[c], a.w1[(size_t)c * a.mid + j], s);
    mid[j] = swishf(s + a.b1[j]);
  }
  __syncthreads();
  for (int c = tid; c < a.C; c += blockDim.x) {
    float s = 0.f;
    for (int j = 0; j < a.mid; ++j) s = fmaf(mid[j], a.w2[(size_t)j * a.C + c], s);
    a.scale[(size_t)b * a.C + c] = sigmoidf_(s + a.b2[c]);
  }
}

void launch_se(const SeArgs& a, int rows, hipStream_t s) {
  hipLaunchKernelGGL(se_kernel, dim3(rows), dim3(256), (1024 + a.C + a.mid) * sizeof(float), s, a);
}

// ------------------------------------------------------------------------------------ fusion / pooling
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

__global__ __launch_bounds__(256) void fuse_kernel(FuseArgs a) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= a.total) return;
  const int C4 = a.C >> 2;
  const int c4 = (int)(gid % C4);
  int64_t p = gid / C4;
  const int x = (int)(p % a.W);
  p /= a.W;
  const int y = (int)(p % a.H);
  const int b = (int)(p / a.H);
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
    s.x = swishf(s.x); s.y = swishf(s.y); s.z = swishf(s.z); s.w = swishf(s.w);
  }
  *(float4*)(a.out + (size_t)gid * 4) = s;
}

void launch_fuse(const FuseArgs& a, hipStream_t s) {
  const int grid = (int)((a.total + 255) / 256);
  hipLaunchKernelGGL(fuse_kernel, dim3(grid), dim3(256), 0, s, a);
}

// ------------------------------------------------------------------------------------ Philox masks
// counter = (c >> 2, row, site, 0), key = (seed lo, seed hi); word c & 3 of the 4 outputs;
// u = (word >> 8) * 2^-24; keep iff u >= rate; scale = 1/(1-rate).   (DESIGN.md "dropout stream")
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

__global__ __launch_bounds__(256) void philox_kernel(float* masks, const int64_t* site_off,
                                                     const int32_t* site_ch, const float* site_rate,
                                                     int rows, int max_c4, uint64_t seed) {
  const int site = blockIdx.y;
  const int C = site_ch[site];
  const int C4 = (C + 3) >> 2;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)rows * max_c4) return;
  const int c4 = (int)(gid % max_c4);
  const int row = (int)(gid / max_c4);
  if (c4 >= C4) return;
  uint32_t w[4];
  philox4x32_10((uint32_t)c4, (uint32_t)row, (uint32_t)site, 0u, (uint32_t)seed,
                (uint32_t)(seed >> 32), w);
  const float rate = site_rate[site];
  const float scale = 1.0f / (1.0f - rate);
  float* dst = masks + site_off[site] + (size_t)row * C + c4 * 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (c4 * 4 + j < C) {
      const float u = (float)(w[j] >> 8) * 5.9604644775390625e-08f;  // 2^-24
      dst[j] = (u >= rate) ? scale : 0.0f;
    }
  }
}

void launch_philox_masks(float* masks, const int64_t* site_off_dev, const int32_t* site_ch_dev,
                         const float* site_rate_dev, int n_sites, int rows, int max_c4,
                         uint64_t seed, hipStream_t s) {
  if (n_sites == 0 || rows == 0) return;
  const int64_t per_site = (int64_t)rows * max_c4;
  hipLaunchKernelGGL(philox_kernel, dim3((unsigned)((per_site + 255) / 256), n_sites), dim3(256), 0, s,
                     masks, site_off_dev, site_ch_dev, site_rate_dev, rows, max_c4, seed);
}

}  // namespace uda
