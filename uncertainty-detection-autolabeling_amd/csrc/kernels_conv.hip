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
#include <stdio.h>
#include <stdlib.h>

#include "uda_internal.h"
#include "fuse_sample.h"

namespace uda {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// x * sigmoid(x) with the hardware exp2 / rcp (1-ulp) instead of an IEEE divide: the activations
// are compared with the oracle at 2e-4, and the divide sequence was a third of the pw epilogue.
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float swishf(float x) { return x * sigmoidf_(x); }

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
  float4 o = make_float4(fmaf(acc.x, s.x, t.x), fmaf(acc.y, s.y, t.y), fmaf(acc.z, s.z, t.z), fmaf(acc.w, s.w, t.w));
  if (a.act == UDA_ACT_SWISH) {
    o.x = swishf(o.x); o.y = swishf(o.y); o.z = swishf(o.z); o.w = swishf(o.w);
  } else if (a.act >= UDA_ACT_RELU) {
    o.x = act_relu_family(o.x, a.act); o.y = act_relu_family(o.y, a.act); o.z = act_relu_family(o.z, a.act); o.w = act_relu_family(o.w, a.act);
  }
  *(float4*)(a.out + (size_t)gid * 4) = o;
}

// One thread = one output pixel x 16 output channels: the 27 input values of the 3x3x3 window are loaded once and
// reused for 16 channels (the 4-channel version issued 27 loads per float4 of output and was bound by the load
// instruction rate: 1.5 TB/s), the weights come out of LDS as broadcast float4 reads.  Same summation order.
__global__ __launch_bounds__(256) void stem16_kernel(StemArgs a) {
  extern __shared__ float wl[];  // [27][Co]
  for (int i = threadIdx.x; i < 27 * a.Co; i += blockDim.x) wl[i] = a.w[i];
  __syncthreads();
  const int cgn = a.Co >> 4;
  const int64_t total = (int64_t)a.rows * a.Ho * a.Wo * cgn;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  const int cg = (int)(gid % cgn);
  int64_t p = gid / cgn;
  const int x = (int)(p % a.Wo);
  p /= a.Wo;
  const int y = (int)(p % a.Ho);
  const int b = (int)(p / a.Ho);
  const float* inb = a.in + (size_t)b * a.H * a.W * 3;
  float v[27];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = y * 2 - a.pad_t + ky;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = x * 2 - a.pad_l + kx;
      const bool in = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      const float* px = inb + ((size_t)(in ? iy : 0) * a.W + (in ? ix : 0)) * 3;
#pragma unroll
      for (int ci = 0; ci < 3; ++ci) v[(ky * 3 + kx) * 3 + ci] = in ? px[ci] : 0.f;
    }
  }
  float4 acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int t = 0; t < 27; ++t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 w = *(const float4*)(wl + t * a.Co + cg * 16 + q * 4);
      acc[q].x = fmaf(v[t], w.x, acc[q].x);
      acc[q].y = fmaf(v[t], w.y, acc[q].y);
      acc[q].z = fmaf(v[t], w.z, acc[q].z);
      acc[q].w = fmaf(v[t], w.w, acc[q].w);
    }
  }
  float* op = a.out + (size_t)gid * 16;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 s = *(const float4*)(a.bn_scale + cg * 16 + q * 4);
    const float4 t = *(const float4*)(a.bn_shift + cg * 16 + q * 4);
    float4 o;
    o.x = swishf(fmaf(acc[q].x, s.x, t.x));
    o.y = swishf(fmaf(acc[q].y, s.y, t.y));
    o.z = swishf(fmaf(acc[q].z, s.z, t.z));
    o.w = swishf(fmaf(acc[q].w, s.w, t.w));
    *(float4*)(op + q * 4) = o;
  }
}

// The same convolution straight from the raw uint8 images (StemArgs::u8): a thread loads the three 9-byte window rows of its
// output pixel (one unaligned 12-byte load each; the byte behind the last image is slack the host allocates), turns every byte
// into its normalised value through the table in LDS (one SDWA shift + one LDS read per value) and runs the float kernel's
// accumulation - same values, same order, bit-identical outputs.  Pixels beyond the raw image (zero padding of the network
// input, dataloader.py:123-152) and beyond the map (SAME padding) contribute 0.  CG output channels per thread (32: the window
// is decoded once per pixel).
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
// SW: Co == CG, every thread computes ALL output channels: the weight addresses are wave-uniform, so the 27 x Co weights come
// through the scalar cache into scalar registers (s_load) and enter the FMAs as scalar operands - no LDS traffic for them (the
// LDS-broadcast version was bound by those 216 ds_read_b128 per thread: 0.83 ms; the float kernel above has the same bound)
template <int CG, bool SW>
__global__ __launch_bounds__(256) void stem_u8_kernel(StemArgs a) {
  extern __shared__ float wl[];  // [27][Co] | table [3][256]
  float* lut = wl + (SW ? 0 : 27 * a.Co);
  if constexpr (!SW)
    for (int i = threadIdx.x; i < 27 * a.Co; i += blockDim.x) wl[i] = a.w[i];
  for (int i = threadIdx.x; i < 768; i += blockDim.x) {
    const int c = i >> 8;
    lut[i] = ((float)(i & 255) - a.mean[c]) / a.stdv[c];
  }
  __syncthreads();
  const int cgn = SW ? 1 : a.Co / CG;
  const int64_t total = (int64_t)a.rows * a.Ho * a.Wo * cgn;
  const int64_t gid_raw = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t gid = gid_raw < total ? gid_raw : total - 1;      // (threads beyond the end recompute the last unit and store nothing)
  const int cg = SW ? 0 : (int)(gid % cgn);
  int64_t p = gid / cgn;
  const int x = (int)(p % a.Wo);
  p /= a.Wo;
  const int y = (int)(p % a.Ho);
  const int b = (int)(p / a.Ho);
  const PreGeo g = a.geo[a.img0 + b];
  const uint8_t* img = a.u8 + g.off;
  float v[27];
  const int ix0 = x * 2 - a.pad_l;
  const bool colsok = ix0 >= 0 && ix0 + 2 < g.w;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = y * 2 - a.pad_t + ky;
    const bool rowok = iy >= 0 && iy < g.h;
    if (rowok && colsok) {
      const uint8_t* q = img + ((size_t)iy * g.w + ix0) * 3;
      const uint32_t u0 = *(const u32_unaligned*)q, u1 = *(const u32_unaligned*)(q + 4), u2 = *(const u32_unaligned*)(q + 8);
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        const uint32_t u = j < 4 ? u0 : (j < 8 ? u1 : u2);
        v[ky * 9 + j] = lut[(j % 3) * 256 + ((u >> (8 * (j & 3))) & 255u)];
      }
    } else {
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ix0 + kx;
        const bool in = rowok && ix >= 0 && ix < g.w;
        const uint8_t* q = img + ((size_t)(in ? iy : 0) * g.w + (in ? ix : 0)) * 3;
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) v[(ky * 3 + kx) * 3 + ci] = in ? lut[ci * 256 + q[ci]] : 0.f;
      }
    }
  }
  float4 acc[CG / 4];
#pragma unroll
  for (int q = 0; q < CG / 4; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int t = 0; t < 27; ++t) {
#pragma unroll
    for (int q = 0; q < CG / 4; ++q) {
      const float4 w = SW ? *(const float4*)(a.w + t * CG + q * 4) : *(const float4*)(wl + t * a.Co + cg * CG + q * 4);
      acc[q].x = fmaf(v[t], w.x, acc[q].x);
      acc[q].y = fmaf(v[t], w.y, acc[q].y);
      acc[q].z = fmaf(v[t], w.z, acc[q].z);
      acc[q].w = fmaf(v[t], w.w, acc[q].w);
    }
  }
  // A thread holds CG consecutive output floats (128 / 64 bytes): stored as they are, one instruction puts 16 bytes into each
  // of 64 different cache lines (the write counter showed 2.5x the tensor's size).  The wave's 64 x CG values go through a
  // wave-private LDS tile instead and leave as contiguous kilobytes: eight (four) lanes per unit, 16 bytes each.
  float* tile = lut + 768 + (threadIdx.x >> 6) * 64 * (CG + 4);
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int q = 0; q < CG / 4; ++q) {
    const float4 s = *(const float4*)(a.bn_scale + cg * CG + q * 4);
    const float4 t = *(const float4*)(a.bn_shift + cg * CG + q * 4);
    float4 o;
    o.x = swishf(fmaf(acc[q].x, s.x, t.x));
    o.y = swishf(fmaf(acc[q].y, s.y, t.y));
    o.z = swishf(fmaf(acc[q].z, s.z, t.z));
    o.w = swishf(fmaf(acc[q].w, s.w, t.w));
    *(float4*)(tile + lane * (CG + 4) + q * 4) = o;
  }
  __builtin_amdgcn_wave_barrier();
  constexpr int LPU = CG / 4;                       // lanes per unit in the store pass
  const int64_t wave0 = gid_raw - lane;             // first unit of this wave
  float* ob = a.out + (size_t)wave0 * CG;
#pragma unroll
  for (int it = 0; it < LPU; ++it) {
    const int u = it * (64 / LPU) + lane / LPU, q = lane % LPU;
    if (wave0 + u < total) *(float4*)(ob + (size_t)u * CG + q * 4) = *(const float4*)(tile + u * (CG + 4) + q * 4);
  }
}

bool stem_u8_supported(int Co) { return (Co & 15) == 0 && Co <= 64; }

void launch_stem(const StemArgs& a, hipStream_t s) {
  if (a.u8) {
    // weights [27][Co] (not for the scalar-weight variant) | table [3][256] | four wave-private store tiles [64][CG + 4]
    auto lds_of = [&](int cg_, bool sw) { return ((sw ? 0 : 27 * (size_t)a.Co) + 768 + 4 * 64 * (size_t)(cg_ + 4)) * sizeof(float); };
    if (a.Co == 32) {              // (EfficientNet-B0 ... B2)
      const int64_t total = (int64_t)a.rows * a.Ho * a.Wo;
      hipLaunchKernelGGL((stem_u8_kernel<32, true>), dim3((unsigned)((total + 255) / 256)), dim3(256), lds_of(32, true), s, a);
    } else if ((a.Co & 31) == 0) {
      const int64_t total = (int64_t)a.rows * a.Ho * a.Wo * (a.Co >> 5);
      hipLaunchKernelGGL((stem_u8_kernel<32, false>), dim3((unsigned)((total + 255) / 256)), dim3(256), lds_of(32, false), s, a);
    } else {
      const int64_t total = (int64_t)a.rows * a.Ho * a.Wo * (a.Co >> 4);
      hipLaunchKernelGGL((stem_u8_kernel<16, false>), dim3((unsigned)((total + 255) / 256)), dim3(256), lds_of(16, false), s, a);
    }
    return;
  }
  static int wide = -1;
  if (wide < 0) { const char* e = getenv("UDA_STEM16"); wide = e ? atoi(e) : 1; }
  if (wide && (a.Co & 15) == 0 && a.act == UDA_ACT_SWISH) {     // (stem16 / stem_u8 are swish kernels; the executor keeps other activations off the uint8 route)
    const int64_t total = (int64_t)a.rows * a.Ho * a.Wo * (a.Co >> 4);
    hipLaunchKernelGGL(stem16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 27 * a.Co * sizeof(float), s, a);
    return;
  }
  const int64_t total = (int64_t)a.rows * a.Ho * a.Wo * (a.Co >> 2);
  const int grid = (int)((total + 255) / 256);
  hipLaunchKernelGGL(stem_kernel, dim3(grid), dim3(256), 27 * a.Co * sizeof(float), s, a);
}

// ------------------------------------------------------------------------------------ pointwise
// 1x1 conv as an f32 MFMA GEMM.  Block = 4 waves, tile = 128 pixels x (32*NT) output channels;
// wave w owns pixel rows [32w, 32w+32) and all NT column tiles (NT accumulators of 16 VGPRs).
//  * K is staged 32 deep through LDS; the next chunk's global loads are issued before the MFMAs
//    of the current one (register prefetch), so HBM latency hides behind the matrix work.
//  * A is staged transposed ([k][m], row stride 129: conflict-free writes and reads): lane
//    (i = lane&31, h = lane>>5) feeds A[m = i][k = kk + h] to v_mfma_f32_32x32x2_f32;
//    B is [k][n] straight from the TF kernel layout [Cin][Cout].
//  * epilogue: accumulators go through a wave-private LDS tile so that every lane stores 16
//    contiguous bytes (a 64-channel row segment = 256 contiguous bytes per 16 lanes), with
//    bias / BN / swish / dropout keep-scale / residual applied on the float4.
constexpr int PW_BK = 32;
constexpr int PW_STG = 68;  // staging row stride (floats): 64 columns + 4 pad

template <int NT, int NW>   // NT 32-column tiles per wave, NW waves (32 pixel rows each) per block
__global__ __launch_bounds__(NW * 64) void pw_kernel(PwArgs a) {
  constexpr int NTH = NW * 64;
  constexpr int BM = 32 * NW, BK = PW_BK, BN = 32 * NT;
  constexpr int A_FLOATS = BK * (BM + 1);
  constexpr int B_FLOATS = BK * BN;
  constexpr int STG_FLOATS = NW * 32 * PW_STG;
  constexpr int LDS_FLOATS = (A_FLOATS + B_FLOATS) > STG_FLOATS ? (A_FLOATS + B_FLOATS) : STG_FLOATS;
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  float(*As)[BM + 1] = (float(*)[BM + 1])lds;
  float(*Bs)[BN] = (float(*)[BN])(lds + A_FLOATS);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, b_in = b / a.in_div;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const float* A = a.in + (size_t)b_in * a.HW * a.Cin;
  const float* se = a.se ? a.se + (size_t)(b / a.se_div) * a.Cin : nullptr;
  const bool vecB = (a.Cout & 3) == 0;

  f32x16 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

  constexpr int B_VEC_ITERS = (BK * BN / 4 + NTH - 1) / NTH;
  constexpr int B_SCL_ITERS = (BK * BN + NTH - 1) / NTH;
  float4 ra[4];
  float4 rb[B_VEC_ITERS];
  float4 rg = make_float4(1.f, 1.f, 1.f, 1.f);   // SE gate of this thread's 4 k columns (same for its 4 rows)

  auto load_chunk = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = tid + NTH * i;
      const int m = f >> 3, kq = f & 7;
      const int k = k0 + 4 * kq;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m0 + m < a.HW && k < a.Cin) v = *(const float4*)(A + (size_t)(m0 + m) * a.Cin + k);
      ra[i] = v;
    }
    if (se) {   // applied when the chunk is written to LDS, so the loads stay in flight over the MFMAs
      const int k = k0 + 4 * (tid & 7);
      rg = (k < a.Cin) ? *(const float4*)(se + k) : make_float4(1.f, 1.f, 1.f, 1.f);
    }
    if (vecB) {
#pragma unroll
      for (int i = 0; i < B_VEC_ITERS; ++i) {
        const int f = tid + NTH * i;
        const int kk = f / (BN / 4), nq = f % (BN / 4);
        const int k = k0 + kk, col = n0 + 4 * nq;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f < BK * BN / 4 && k < a.Cin && col < a.Cout) v = *(const float4*)(a.w + (size_t)k * a.Cout + col);
        rb[i] = v;
      }
    }
  };
  auto store_chunk = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = tid + NTH * i;
      const int m = f >> 3, kq = f & 7;
      As[4 * kq + 0][m] = ra[i].x * rg.x;
      As[4 * kq + 1][m] = ra[i].y * rg.y;
      As[4 * kq + 2][m] = ra[i].z * rg.z;
      As[4 * kq + 3][m] = ra[i].w * rg.w;
    }
    if (vecB) {
#pragma unroll
      for (int i = 0; i < B_VEC_ITERS; ++i) {
        const int f = tid + NTH * i;
        if (f < BK * BN / 4) {
          const int kk = f / (BN / 4), nq = f % (BN / 4);
          *(float4*)&Bs[kk][4 * nq] = rb[i];
        }
      }
    } else {  // Cout not a multiple of 4 (class head: 9*7 = 63): scalar weight loads
#pragma unroll
      for (int i = 0; i < B_SCL_ITERS; ++i) {
        const int f = tid + NTH * i;
        const int kk = f / BN, n = f % BN;
        const int k = k0 + kk, col = n0 + n;
        if (f < BK * BN) Bs[kk][n] = (k < a.Cin && col < a.Cout) ? a.w[(size_t)k * a.Cout + col] : 0.f;
      }
    }
  };

  load_chunk(0);
  for (int k0 = 0; k0 < a.Cin; k0 += BK) {
    store_chunk(k0);
    __syncthreads();
    if (k0 + BK < a.Cin) load_chunk(k0 + BK);   // in flight during the MFMAs below
    const int kend = min(BK, a.Cin - k0);
    if (kend == BK) {
#pragma unroll
      for (int kk = 0; kk < BK; kk += 2) {
        const float av = As[kk + lh][wave * 32 + li];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const float bv = Bs[kk + lh][n * 32 + li];
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[n], 0, 0, 0);
        }
      }
    } else {
      for (int kk = 0; kk < kend; kk += 2) {
        const float av = As[kk + lh][wave * 32 + li];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const float bv = Bs[kk + lh][n * 32 + li];
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[n], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  const size_t out_base = (size_t)b * a.HW;
  const size_t res_base = a.res ? (size_t)(b / a.res_div) * a.HW : 0;

  if (!vecB) {
    // scalar epilogue: lane holds column li of 16 rows: row = (r&3) + 8*(r>>2) + 4*lh
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
        else if (a.act >= UDA_ACT_RELU) v = act_relu_family(v, a.act);
        v *= mk;
        if (a.res) v += a.res[(res_base + m) * a.Cout + col];
        a.out[(out_base + m) * a.Cout + col] = v;
      }
    }
    return;
  }

  // vector epilogue through a wave-private staging tile [32 rows][64 cols (+4 pad)]
  float* stg = lds + wave * 32 * PW_STG;
  const int rrow = lane >> 4, c4 = lane & 15;   // read-back: 16 lanes per row, 4 rows per pass
#pragma unroll
  for (int p = 0; p < (NT + 1) / 2; ++p) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int n = 2 * p + q;
      if (n < NT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * lh) * PW_STG + q * 32 + li] = acc[n][r];
      }
    }
    __syncthreads();
    const int col = n0 + p * 64 + 4 * c4;
    const bool colok = (col < a.Cout) && (p * 64 + 4 * c4 < BN);
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f), sc = make_float4(1.f, 1.f, 1.f, 1.f);
    float4 sh = make_float4(0.f, 0.f, 0.f, 0.f), mk = make_float4(1.f, 1.f, 1.f, 1.f);
    if (colok) {
      if (a.bias) bias = *(const float4*)(a.bias + col);
      if (a.bn_scale) {
        sc = *(const float4*)(a.bn_scale + col);
        sh = *(const float4*)(a.bn_shift + col);
      }
      if (a.mask) mk = *(const float4*)(a.mask + (size_t)b * a.Cout + col);
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = it * 4 + rrow;
      const int m = m0 + wave * 32 + row;
      if (colok && m < a.HW) {
        float4 v = *(const float4*)(stg + row * PW_STG + 4 * c4);
        v.x = fmaf(v.x + bias.x, sc.x, sh.x);
        v.y = fmaf(v.y + bias.y, sc.y, sh.y);
        v.z = fmaf(v.z + bias.z, sc.z, sh.z);
        v.w = fmaf(v.w + bias.w, sc.w, sh.w);
        if (a.act == UDA_ACT_SWISH) {
          v.x = swishf(v.x); v.y = swishf(v.y); v.z = swishf(v.z); v.w = swishf(v.w);
        } else if (a.act >= UDA_ACT_RELU) {
          v.x = act_relu_family(v.x, a.act); v.y = act_relu_family(v.y, a.act); v.z = act_relu_family(v.z, a.act); v.w = act_relu_family(v.w, a.act);
        }
        v.x *= mk.x; v.y *= mk.y; v.z *= mk.z; v.w *= mk.w;
        if (a.res) {
          const float4 rr = *(const float4*)(a.res + (res_base + m) * a.Cout + col);
          v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
        }
        *(float4*)(a.out + (out_base + m) * a.Cout + col) = v;
      }
    }
    __syncthreads();
  }
}

template <int NW>
static void launch_pw_nw(const PwArgs& a, int rows, int nt, hipStream_t s) {
  const int gx = (a.HW + 32 * NW - 1) / (32 * NW);
  const dim3 grid(gx, (a.Cout + 32 * nt - 1) / (32 * nt), rows), block(NW * 64);
  switch (nt) {
    case 1: hipLaunchKernelGGL((pw_kernel<1, NW>), grid, block, 0, s, a); break;
    case 2: hipLaunchKernelGGL((pw_kernel<2, NW>), grid, block, 0, s, a); break;
    case 3: hipLaunchKernelGGL((pw_kernel<3, NW>), grid, block, 0, s, a); break;
    default: hipLaunchKernelGGL((pw_kernel<4, NW>), grid, block, 0, s, a); break;
  }
}

void launch_pw(const PwArgs& a, int rows, hipStream_t s) {
  // one pass over the columns when Cout <= 192, else the fewest passes of <= 6 column tiles
  static int maxnt = -1, maxnt_small = -1;
  if (maxnt < 0) {
    const char* e = getenv("UDA_PW_MAXNT");
    maxnt = e ? atoi(e) : 4;
    const char* e2 = getenv("UDA_PW_MAXNT_SMALLK");
    maxnt_small = e2 ? atoi(e2) : maxnt;
  }
  const int cap = 32 * (a.Cin <= 48 ? maxnt_small : maxnt);
  const int passes = (a.Cout + cap - 1) / cap;
  const int per = (a.Cout + passes - 1) / passes;
  int nt = (per + 31) / 32;
  if (nt > 4) nt = 4;
  static int big = -1;
  if (big < 0) { const char* e = getenv("UDA_PW_BIG"); big = e ? atoi(e) : 0; }
  // 256-pixel blocks (8 waves) halve the weight-tile traffic per pixel; measured 6 % slower on the deep layers (off)
  if (big && a.Cin >= 80 && a.HW >= 256) launch_pw_nw<8>(a, rows, nt, s);
  else launch_pw_nw<4>(a, rows, nt, s);
}

// ------------------------------------------------------------------------------------ depthwise
// Thread = 4 channels x XB consecutive output columns x DW_ROWS output rows.  Consecutive
// threads walk the channel quads of a pixel first (16-byte loads, fully coalesced NHWC),
// then the x-groups.  The per-tile channel sums for squeeze-excite are reduced in a fixed
// order (deterministic; no float atomics).
// output rows per block: 8, 16 for 5x5 stride 1 (vertical halo re-read (R + 4) / R: 1.5 -> 1.25)
static inline int dw_rows(int k, int stride) { return (k == 5 && stride == 1) ? 16 : 8; }

// Each thread keeps a K-row x NCOL-column window of its 4 channels in registers and slides it
// down the DW_ROWS output rows: every input element is loaded once per block column strip
// (instead of K times), the K*K weight quads of the block's channels sit in LDS.
// PF = input rows requested ahead of the row being consumed: the window ring holds K + PF rows, so the
// loads of row r + PF are in flight while the outputs that end at row r are computed (the compiler
// turns the distance into a counted s_waitcnt vmcnt).
template <int K, int S, int XB, int PF>
__global__ __launch_bounds__(256, (K == 5) ? 2 : 3) void dw_kernel(DwArgs a) {
  constexpr int DW_ROWS = (K == 5 && S == 1) ? 16 : 8;
  extern __shared__ float4 dsm[];          // wts[K*K][tc] | red[blockDim]
  float4* wts = dsm;
  float4* red = dsm + K * K * a.tc;
  const int tid = threadIdx.x;
  const int c4l = tid % a.tc, pg = tid / a.tc;
  const int b = blockIdx.z / a.n_cchunk, cc = blockIdx.z % a.n_cchunk;
  const int c4 = cc * a.tc + c4l;
  const int C4 = a.C >> 2;
  for (int i = tid; i < K * K * a.tc; i += blockDim.x) {
    const int tap = i / a.tc, q = cc * a.tc + i % a.tc;
    wts[i] = (q < C4) ? *(const float4*)(a.w + (size_t)tap * a.C + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  const int x0 = (blockIdx.x * a.pxb + pg) * XB;
  const bool active = (pg < a.pxb) && (c4 < C4) && (x0 < a.Wo);
  constexpr int NCOL = (XB - 1) * S + K;
  constexpr int IN_ROWS = (DW_ROWS - 1) * S + K;
  constexpr int RING = K + PF;
  float4 ssum = make_float4(0.f, 0.f, 0.f, 0.f);

  if (active) {
    const int b_in = b / a.in_div;
    const float* inb = a.in + (size_t)b_in * a.H * a.W * a.C + c4 * 4;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bn_scale) {
      sc = *(const float4*)(a.bn_scale + c4 * 4);
      sh = *(const float4*)(a.bn_shift + c4 * 4);
    }
    float4 mk = make_float4(1.f, 1.f, 1.f, 1.f);
    if (a.mask) mk = *(const float4*)(a.mask + (size_t)b * a.C + c4 * 4);
    const int y0 = blockIdx.y * DW_ROWS;
    const int iy_base = y0 * S - a.pad_t, ix0 = x0 * S - a.pad_l;
    float4 win[RING][NCOL];
#pragma unroll
    for (int it = 0; it < IN_ROWS + PF; ++it) {
      if (it < IN_ROWS) {                      // request input row `it`
        const int iy = iy_base + it;
        const bool rowok = (iy >= 0) && (iy < a.H);
        const float* rowp = inb + (size_t)(rowok ? iy : 0) * a.W * a.C;
#pragma unroll
        for (int j = 0; j < NCOL; ++j) {
          const int ix = ix0 + j;
          win[it % RING][j] = (rowok && ix >= 0 && ix < a.W) ? *(const float4*)(rowp + (size_t)ix * a.C)
                                                           : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      const int r = it - PF;                   // newest row that must have arrived
      if (r >= K - 1 && (r - (K - 1)) % S == 0) {
        const int o_row = (r - (K - 1)) / S;
        const int y = y0 + o_row;
        if (y < a.Ho) {
          float4 acc[XB];
#pragma unroll
          for (int o = 0; o < XB; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int ky = 0; ky < K; ++ky) {
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
              const float4 w = wts[(ky * K + kx) * a.tc + c4l];
#pragma unroll
              for (int o = 0; o < XB; ++o) {
                const float4 v = win[(o_row * S + ky) % RING][o * S + kx];
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
            if (x < a.Wo) {
              float4 v;
              v.x = fmaf(acc[o].x, sc.x, sh.x);
              v.y = fmaf(acc[o].y, sc.y, sh.y);
              v.z = fmaf(acc[o].z, sc.z, sh.z);
              v.w = fmaf(acc[o].w, sc.w, sh.w);
              if (a.act == UDA_ACT_SWISH) {
                v.x = swishf(v.x); v.y = swishf(v.y); v.z = swishf(v.z); v.w = swishf(v.w);
              } else if (a.act >= UDA_ACT_RELU) {
                v.x = act_relu_family(v.x, a.act); v.y = act_relu_family(v.y, a.act); v.z = act_relu_family(v.z, a.act); v.w = act_relu_family(v.w, a.act);
              }
              v.x *= mk.x; v.y *= mk.y; v.z *= mk.z; v.w *= mk.w;
              *(float4*)(outp + (size_t)x * a.C) = v;
              ssum.x += v.x; ssum.y += v.y; ssum.z += v.z; ssum.w += v.w;
            }
          }
        }
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

static inline int dw_xb(int k, int stride) { return k == 5 ? 1 : (stride == 1 ? 4 : 2); }

void dw_geometry(int C, int Wo, int k, int stride, int* tc, int* pxb, int* n_cchunk, int* grid_x, int* xb) {
  const int C4 = C / 4;
  // channel quads per block: <= 64 for 3x3 (K*K weight quads in LDS <= 9.2 KB); <= 16 for 5x5 so that a block
  // spans >= 16 output columns (horizontal halo re-read (16 + 4) / 16 instead of (4 + 4) / 4)
  const int cap = (k == 5) ? 16 : 64;
  const int ncc = (C4 + cap - 1) / cap;
  const int t = (C4 + ncc - 1) / ncc;
  int p = 256 / t;
  if (p < 1) p = 1;
  const int x = dw_xb(k, stride);
  // do not spread one block over more columns than the row has, and balance the blocks of a row
  const int need = (Wo + x - 1) / x;
  if (p > need) p = need;
  if (k == 5) p = (need + (need + p - 1) / p - 1) / ((need + p - 1) / p);
  *tc = t;
  *pxb = p;
  *n_cchunk = ncc;
  *xb = x;
  *grid_x = (Wo + p * x - 1) / (p * x);
}

int dw_tiles(int C, int Ho, int Wo, int k, int stride) {
  int tc, pxb, ncc, gx, xb;
  dw_geometry(C, Wo, k, stride, &tc, &pxb, &ncc, &gx, &xb);
  return ((Ho + dw_rows(k, stride) - 1) / dw_rows(k, stride)) * gx;
}

void launch_dw(DwArgs a, int rows, int k, int stride, hipStream_t s) {
  int gx, xb;
  dw_geometry(a.C, a.Wo, k, stride, &a.tc, &a.pxb, &a.n_cchunk, &gx, &xb);
  const int gy = (a.Ho + dw_rows(k, stride) - 1) / dw_rows(k, stride);
  a.n_tiles = gy * gx;
  int threads = a.tc * a.pxb;
  threads = (threads + 63) / 64 * 64;
  const dim3 grid(gx, gy, rows * a.n_cchunk), block(threads);
  const size_t lds = ((size_t)k * k * a.tc + threads) * sizeof(float4);
  if (k == 3 && stride == 1) hipLaunchKernelGGL((dw_kernel<3, 1, 4, 0>), grid, block, lds, s, a);
  else if (k == 3 && stride == 2) hipLaunchKernelGGL((dw_kernel<3, 2, 2, 0>), grid, block, lds, s, a);
  else if (k == 5 && stride == 1) hipLaunchKernelGGL((dw_kernel<5, 1, 1, 2>), grid, block, lds, s, a);
  else hipLaunchKernelGGL((dw_kernel<5, 2, 1, 3>), grid, block, lds, s, a);
}

// ------------------------------------------------------------------------------------ squeeze-excite
// One block per sample row: channel means from the depthwise kernel's tile sums (fixed
// order), then the two tiny dense layers.
constexpr int SE_THREADS = 1024;   // the tile-sum reduction is a chain of dependent loads per thread: more threads = shorter chains

__global__ __launch_bounds__(SE_THREADS) void se_kernel(SeArgs a) {
  extern __shared__ float sm[];  // red[SE_THREADS float4] | mean[C] | mid[mid]
  float4* red = (float4*)sm;
  float* mean = sm + 4 * SE_THREADS;
  float* mid = mean + a.C;
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  // deferred dropout: the depthwise output is shared by the samples of an image and its keep-scale m[b][c]
  // (per sample row, per channel) commutes with the spatial mean: mean_b = m[b] * mean_image; the gate that
  // the projection applies to its (shared) input is then sigmoid(...) * m[b]   (SpatialDropout2D, noise [N,1,1,C])
  const float* part = a.partial + (size_t)(b / a.in_div) * a.n_tiles * a.C;
  const float* dm = a.mask ? a.mask + (size_t)b * a.C : nullptr;
  const int C4 = a.C >> 2;
  // channel sums: thread = (channel quad, tile group); groups are combined in a fixed order (deterministic)
  for (int cbase = 0; cbase < C4; cbase += SE_THREADS) {
    const int cw = min(SE_THREADS, C4 - cbase);
    const int G = SE_THREADS / cw;
    const int c4 = cbase + tid % cw, g = tid / cw;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g < G) {
      // tiles g, g + G, g + 2G, ...: eight loads in flight per step (the reduction is a chain of load latencies: 1120 tiles
      // over 42 groups = 27 per thread were 14 round trips with two loads per step, now 4), two accumulation chains
      // (even / odd position), summed in a fixed order; positions past the end add zeros
      float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int t = g; t < a.n_tiles; t += 8 * G) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int tt = t + k * G;
          v[k] = tt < a.n_tiles ? *(const float4*)(part + (size_t)tt * a.C + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
          s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w;
          s1.x += v[k + 1].x; s1.y += v[k + 1].y; s1.z += v[k + 1].z; s1.w += v[k + 1].w;
        }
      }
      s.x += s1.x; s.y += s1.y; s.z += s1.z; s.w += s1.w;
    }
    red[tid] = s;
    __syncthreads();
    if (g == 0) {
      for (int g2 = 1; g2 < G; ++g2) {
        const float4 v = red[g2 * cw + tid];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      float4 m1 = make_float4(1.f, 1.f, 1.f, 1.f);
      if (dm) m1 = *(const float4*)(dm + c4 * 4);
      mean[c4 * 4 + 0] = s.x * a.inv_hw * m1.x;
      mean[c4 * 4 + 1] = s.y * a.inv_hw * m1.y;
      mean[c4 * 4 + 2] = s.z * a.inv_hw * m1.z;
      mean[c4 * 4 + 3] = s.w * a.inv_hw * m1.w;
    }
    __syncthreads();
  }
  // first dense layer: thread = (hidden unit j, slice of the channels); slices are combined in a fixed order
  {
    const int P = min(SE_THREADS / a.mid, a.C);          // channel slices per hidden unit (mid <= SE_THREADS)
    const int j = tid % a.mid, part = tid / a.mid;
    float s = 0.f;
    if (part < P) {
      // (sixteen weights in flight per step, summed in channel order: left as a plain loop this was one load - wait - fma round
      // trip per channel, 55 in a row for C = 1152, and the whole kernel 38 us at batch 1)
      const float* wp = a.w1 + j;
      int c = part;
      for (; c + 15 * P < a.C; c += 16 * P) {
        float w[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) w[q] = wp[(size_t)(c + q * P) * a.mid];
#pragma unroll
        for (int q = 0; q < 16; ++q) s = fmaf(mean[c + q * P], w[q], s);
      }
      for (; c + 3 * P < a.C; c += 4 * P) {
        float w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = wp[(size_t)(c + q * P) * a.mid];
#pragma unroll
        for (int q = 0; q < 4; ++q) s = fmaf(mean[c + q * P], w[q], s);
      }
      for (; c < a.C; c += P) s = fmaf(mean[c], wp[(size_t)c * a.mid], s);
    }
    float* redf = (float*)red;
    redf[tid] = s;
    __syncthreads();
    if (part == 0) {
      for (int q = 1; q < P; ++q) s += redf[q * a.mid + j];
      const float h = s + a.b1[j];
      mid[j] = a.act == UDA_ACT_SWISH ? swishf(h) : (a.act >= UDA_ACT_RELU ? act_relu_family(h, a.act) : h);
    }
  }
  __syncthreads();
  for (int c = tid; c < a.C; c += blockDim.x) {
    float s = 0.f;
    const float* wp = a.w2 + c;
    int j = 0;
    for (; j + 16 <= a.mid; j += 16) {
      float w[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) w[q] = wp[(size_t)(j + q) * a.C];
#pragma unroll
      for (int q = 0; q < 16; ++q) s = fmaf(mid[j + q], w[q], s);
    }
    for (; j + 4 <= a.mid; j += 4) {
      float w[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) w[q] = wp[(size_t)(j + q) * a.C];
#pragma unroll
      for (int q = 0; q < 4; ++q) s = fmaf(mid[j + q], w[q], s);
    }
    for (; j < a.mid; ++j) s = fmaf(mid[j], wp[(size_t)j * a.C], s);
    a.scale[(size_t)b * a.C + c] = sigmoidf_(s + a.b2[c]) * (dm ? dm[c] : 1.f);
  }
}

void launch_se(const SeArgs& a, int rows, hipStream_t s) {
  hipLaunchKernelGGL(se_kernel, dim3(rows), dim3(SE_THREADS), (4 * SE_THREADS + a.C + a.mid) * sizeof(float), s, a);
}

// ------------------------------------------------------------------------------------ fusion / pooling
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
  *(float4*)(a.out + (size_t)gid * 4) = fuse_value(a, b, y, x, c4);     // fuse_sample.h
}

void launch_fuse(const FuseArgs& a, hipStream_t s) {
  const int grid = (int)((a.total + 255) / 256);
  hipLaunchKernelGGL(fuse_kernel, dim3(grid), dim3(256), 0, s, a);
}

// ------------------------------------------------------------------------------------ Philox masks
// counter = (c >> 2, row, site, 0), key = (seed lo, seed hi); word c & 3 of the 4 outputs;
// u = (word >> 8) * 2^-24; keep iff u >= rate; scale = 1/(1-rate).   (DESIGN.md "dropout stream")
__global__ __launch_bounds__(256) void philox_kernel(float* masks, const int64_t* site_off,
                                                     const int32_t* site_ch, const float* site_rate,
                                                     int rows, uint32_t row_base, int max_c4, uint64_t seed,
                                                     int t_local, int t_total, int t_first, int t_stride) {
  const int site = blockIdx.y;
  const int C = site_ch[site];
  const int C4 = (C + 3) >> 2;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)rows * max_c4) return;
  const int c4 = (int)(gid % max_c4);
  const int row = (int)(gid / max_c4);
  if (c4 >= C4) return;
  uint32_t w[4];
  // row of the GLOBAL sample axis: local sample j of image n is sample t_first + j * t_stride of t_total (a handle that runs
  // every sample: t_local = t_total, first 0, stride 1 - the local row itself)
  const uint32_t grow = row_base + (uint32_t)(row / t_local) * (uint32_t)t_total + (uint32_t)t_first + (uint32_t)(row % t_local) * (uint32_t)t_stride;
  philox4x32_10((uint32_t)c4, grow, (uint32_t)site, 0u, (uint32_t)seed,
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
                         const float* site_rate_dev, int n_sites, int rows, uint32_t row_base, int max_c4,
                         uint64_t seed, int t_local, int t_total, int t_first, int t_stride, hipStream_t s) {
  if (n_sites == 0 || rows == 0) return;
  const int64_t per_site = (int64_t)rows * max_c4;
  hipLaunchKernelGGL(philox_kernel, dim3((unsigned)((per_site + 255) / 256), n_sites), dim3(256), 0, s,
                     masks, site_off_dev, site_ch_dev, site_rate_dev, rows, row_base, max_c4, seed, t_local, t_total, t_first, t_stride);
}

// ------------------------------------------------------------------------------------ fused MBConv front half
// expand 1x1 (f32 MFMA) + BN + swish + dropout -> depthwise kxk / stride s (TF SAME) + BN + swish +
// dropout + SE tile sums, one output tile per block, 32 expanded channels at a time:
//   X  : the input tile with halo, all Cin channels, transposed [k][pixel] in LDS (MFMA A operand)
//   E  : the expanded + activated tile of the current 32 channels [pixel][33] in LDS; positions
//        outside the image are ZERO (TF pads the depthwise INPUT, i.e. the expanded activation)
//   the depthwise stage reads E with a sliding window along x, writes the output tile (128-byte
//   channel segments) and the per-tile channel sums for squeeze-excite.
// The 6x-expanded tensor (94 MB per image-sample at block 1) never goes to HBM; the price is the
// halo recompute of the expand GEMM (1.4x for 3x3, 1.9x for 5x5 at an 8x16 tile).
namespace {
struct MbxCfg { int th, tw; };
__host__ __device__ constexpr MbxCfg mbx_cfg(int k, int s) {
  return s == 1 ? MbxCfg{8, 16} : (k == 3 ? MbxCfg{4, 16} : MbxCfg{4, 8});
}
}  // namespace

template <int K, int S, int KS, int NW>   // KS = Cin / 2 MFMA k-steps; NW = waves per block (4 or 8)
__global__ __launch_bounds__(NW * 64) void mbx_kernel(MbxArgs a) {
  constexpr int TH = mbx_cfg(K, S).th, TW = mbx_cfg(K, S).tw;
  constexpr int IH = (TH - 1) * S + K, IW = (TW - 1) * S + K;
  constexpr int NP = IH * IW;
  constexpr int NPP = (NP + 31) / 32 * 32;
  constexpr int NMT = NPP / 32;               // MFMA row tiles of the input tile
  constexpr int NTH = NW * 64;                // threads per block
  constexpr int NG = NTH / 32;                // depthwise thread groups (32 channels each)
  constexpr int MT_PER_WAVE = (NMT + NW - 1) / NW;
  constexpr int XS = NPP + 1;                 // X row stride (floats)
  constexpr int ES = 33;                      // E row stride
  constexpr int GPR = NG / TH;                // thread groups per output row
  constexpr int XW = TW / GPR;                // outputs per thread along x
  constexpr int NCOL = (XW - 1) * S + K;
  extern __shared__ float mlds[];
  float* X = mlds;                            // [Cin + 2][XS]: row Cin = 1 inside the image else 0, row Cin+1 = 0
  float* E = mlds + (size_t)(a.Cin + 2) * XS; // [NPP][ES]
  float* red = E + (size_t)NPP * ES;          // [NG][32]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, b_in = b / a.in_div;
  // diagnostic build only: wave-0 / wave-3 phase stamps of a few blocks (never set in production)
  unsigned long long* stp = nullptr;
  int stn = 0;
  if (a.stamps && (lane == 0) && (wave == 0 || wave == NW - 1) && blockIdx.x == 3 && blockIdx.y == 5 && blockIdx.z < 8)
    stp = a.stamps + ((size_t)blockIdx.z * 2 + (wave != 0)) * 64;
#define MBX_STAMP() do { if (stp && stn < 64) stp[stn++] = clock64(); } while (0)
  MBX_STAMP();
  const int oy0 = blockIdx.y * TH, ox0 = blockIdx.x * TW;
  const int iy0 = oy0 * S - a.pad_t, ix0 = ox0 * S - a.pad_l;
  const float* xin = a.in + (size_t)b_in * a.H * a.W * a.Cin;

  // ---- stage the input tile (zero outside the image / beyond NP); loads are issued 4 deep
  const int cq = a.Cin >> 2;
  const int nf = NPP * cq;
  for (int f0 = tid; f0 < nf; f0 += 4 * NTH) {
    float4 v[4];
    int pp[4], qq[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int f = f0 + u * NTH;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      pp[u] = f / cq;
      qq[u] = f - pp[u] * cq;
      if (f < nf && pp[u] < NP) {
        const int iy = iy0 + pp[u] / IW, ix = ix0 + pp[u] % IW;
        if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
          v[u] = *(const float4*)(xin + ((size_t)iy * a.W + ix) * a.Cin + 4 * qq[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (f0 + u * NTH < nf) {
        float* xp = X + (size_t)(4 * qq[u]) * XS + pp[u];
        xp[0] = v[u].x; xp[XS] = v[u].y; xp[2 * XS] = v[u].z; xp[3 * XS] = v[u].w;
      }
    }
  }
  // The BN shift rides on an extra input channel that is 1 inside the image and 0 in the halo /
  // padding: acc = sum_k x_k (w_k * scale) + inside * shift, so a position outside the image gives
  // exactly swish(0) = 0 — the zero padding TF applies to the depthwise INPUT — with no per-element test.
  for (int p = tid; p < NPP; p += NTH) {
    const int iy = iy0 + p / IW, ix = ix0 + p % IW;
    X[(size_t)a.Cin * XS + p] = (p < NP && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) ? 1.f : 0.f;
    X[(size_t)(a.Cin + 1) * XS + p] = 0.f;
  }
  MBX_STAMP();
  __syncthreads();
  MBX_STAMP();

  const int c = tid & 31, g = tid >> 5;       // depthwise stage: channel within the chunk, thread group
  const int orow = g / GPR, oxs = (g % GPR) * XW;
  const size_t tile = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  const int oy = oy0 + orow;

  // All per-chunk operands (expand column of We, BN/mask scalars, the K*K depthwise taps) are
  // requested one chunk ahead, right after the expand phase and BEFORE the depthwise phase issues
  // its output stores: the next expand phase then waits with a counted vmcnt on loads that are
  // older than those stores instead of draining them (vmcnt retires loads and stores in order).
  struct ChunkParams {
    float bf[KS + 1];       // bf[KS]: the (inside, 0) row pair -> BN shift for lh == 0, 0 for lh == 1
    float mk0;
    float wk[K * K];
    float sc1, sh1, mk1;
  };
  auto load_params = [&](int c0, ChunkParams& q) {
    const int ecol = c0 + li;
    const bool eok = ecol < a.Cmid;
    const float sc = eok ? a.sc0[ecol] : 0.f;
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) q.bf[s2] = eok ? a.we[(size_t)(2 * s2 + lh) * a.Cmid + ecol] * sc : 0.f;
    q.bf[KS] = (eok && lh == 0) ? a.sh0[ecol] : 0.f;
    q.mk0 = (eok && a.mask0) ? a.mask0[(size_t)b * a.Cmid + ecol] : 1.f;
    const int dcol_ = c0 + c;
    const bool dok = dcol_ < a.Cmid;
#pragma unroll
    for (int t = 0; t < K * K; ++t) q.wk[t] = dok ? a.wd[(size_t)t * a.Cmid + dcol_] : 0.f;
    q.sc1 = dok ? a.sc1[dcol_] : 0.f;
    q.sh1 = dok ? a.sh1[dcol_] : 0.f;
    q.mk1 = (dok && a.mask1) ? a.mask1[(size_t)b * a.Cmid + dcol_] : 1.f;
  };
  ChunkParams cur, nxt;
  load_params(0, cur);

  for (int c0 = 0; c0 < a.Cmid; c0 += 32) {
    const int col = c0 + c;
    const bool dcol = col < a.Cmid;
    float (&bfr)[KS + 1] = cur.bf;
    float (&wk)[K * K] = cur.wk;
    const float mk0 = cur.mk0;
    const float sc1 = cur.sc1, sh1 = cur.sh1, mk1 = cur.mk1;

    // ---- expand: E[p][j] = swish(bn0(sum_k X[k][p] * We[k][c0 + j])) * mask0, zero outside the image
#pragma unroll
    for (int t = 0; t < MT_PER_WAVE; ++t) {
      const int mt = wave + NW * t;
      if (mt < NMT) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* xa = X + (size_t)lh * XS + mt * 32 + li;
#pragma unroll
        for (int s2 = 0; s2 <= KS; ++s2)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[(size_t)2 * s2 * XS], bfr[s2], acc, 0, 0, 0);
        float* ep = E + (size_t)(mt * 32 + 4 * lh) * ES + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) ep[((r & 3) + 8 * (r >> 2)) * ES] = swishf(acc[r]) * mk0;
      }
    }
    // operands of the NEXT chunk: in flight during the depthwise phase, older than its stores
    if (c0 + 32 < a.Cmid) load_params(c0 + 32, nxt);
    MBX_STAMP();
    __syncthreads();
    MBX_STAMP();
    // ---- depthwise on E for channel c0 + c
    float ssum = 0.f;
    if (dcol && oy < a.Ho) {
      float acc[XW];
#pragma unroll
      for (int o = 0; o < XW; ++o) acc[o] = 0.f;
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        float rowv[NCOL];
        const float* er = E + ((size_t)(orow * S + ky) * IW + oxs * S) * ES + c;
#pragma unroll
        for (int j = 0; j < NCOL; ++j) rowv[j] = er[j * ES];
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
#pragma unroll
          for (int o = 0; o < XW; ++o) acc[o] = fmaf(rowv[o * S + kx], wk[ky * K + kx], acc[o]);
        }
      }
      float* op = a.out + (((size_t)b * a.Ho + oy) * a.Wo + ox0 + oxs) * a.Cmid + col;
#pragma unroll
      for (int o = 0; o < XW; ++o) {
        if (ox0 + oxs + o < a.Wo) {
          const float v = swishf(fmaf(acc[o], sc1, sh1)) * mk1;
          op[(size_t)o * a.Cmid] = v;
          ssum += v;
        }
      }
    }
    MBX_STAMP();
    if (a.se_partial) {
      red[g * 32 + c] = ssum;
      __syncthreads();
      if (g == 0 && dcol) {
        float t = red[c];
#pragma unroll
        for (int gg = 1; gg < NG; ++gg) t += red[gg * 32 + c];
        a.se_partial[((size_t)b * a.n_tiles + tile) * a.Cmid + col] = t;
      }
    }
    __syncthreads();   // E (and red) are rewritten by the next channel chunk
    MBX_STAMP();
    cur = nxt;
  }
#undef MBX_STAMP
}

bool mbx_supported(int Cin, int Cmid, int k, int stride) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("UDA_FUSE_MBX"); on = e ? atoi(e) : 1; }
  return on && Cin % 8 == 0 && Cin >= 16 && Cin <= 48 && Cmid % 4 == 0 && (k == 3 || k == 5) && (stride == 1 || stride == 2);
}

int mbx_tiles(int Ho, int Wo, int k, int stride) {
  const MbxCfg c = mbx_cfg(k, stride);
  return ((Ho + c.th - 1) / c.th) * ((Wo + c.tw - 1) / c.tw);
}

template <int K, int S, int KS, int NW>
static void launch_mbx_t(const MbxArgs& a, int rows, hipStream_t s) {
  constexpr int TH = mbx_cfg(K, S).th, TW = mbx_cfg(K, S).tw;
  constexpr int IH = (TH - 1) * S + K, IW = (TW - 1) * S + K;
  constexpr int NPP = (IH * IW + 31) / 32 * 32;
  const size_t lds = ((size_t)(a.Cin + 2) * (NPP + 1) + (size_t)NPP * 33 + NW * 64) * sizeof(float);
  static size_t attr_lds = 64 * 1024;      // above the default limit the kernel needs an explicit opt-in
  if (lds > attr_lds) {
    hipFuncSetAttribute((const void*)mbx_kernel<K, S, KS, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_lds = lds;
  }
  const dim3 grid((a.Wo + TW - 1) / TW, (a.Ho + TH - 1) / TH, rows);
  hipLaunchKernelGGL((mbx_kernel<K, S, KS, NW>), grid, dim3(NW * 64), lds, s, a);
}

template <int K, int S, int NW>
static void launch_mbx_nw(const MbxArgs& a, int rows, hipStream_t s) {
  switch (a.Cin) {
    case 16: launch_mbx_t<K, S, 8, NW>(a, rows, s); break;
    case 24: launch_mbx_t<K, S, 12, NW>(a, rows, s); break;
    case 32: launch_mbx_t<K, S, 16, NW>(a, rows, s); break;
    case 40: launch_mbx_t<K, S, 20, NW>(a, rows, s); break;
    default: launch_mbx_t<K, S, 24, NW>(a, rows, s); break;   // 48
  }
}

template <int K, int S>
static void launch_mbx_ks(const MbxArgs& a, int rows, hipStream_t s) {
  static int nw = -1;
  if (nw < 0) { const char* e = getenv("UDA_MBX_WAVES"); nw = e ? atoi(e) : 4; }
  if (nw == 4) launch_mbx_nw<K, S, 4>(a, rows, s);
  else launch_mbx_nw<K, S, 8>(a, rows, s);
}

void launch_mbx(const MbxArgs& a0, int rows, int k, int stride, hipStream_t s) {
  MbxArgs a = a0;
  static unsigned long long* d_stamps = nullptr;
  static int want = -1;
  if (want < 0) { const char* e = getenv("UDA_MBX_STAMPS"); want = e ? atoi(e) : 0; }
  if (want) {   // diagnostic: dump the stamps of the previous launch, then arm this one
    static unsigned long long h[8 * 2 * 64];
    if (!d_stamps) { hipMalloc((void**)&d_stamps, sizeof(h)); hipMemset(d_stamps, 0, sizeof(h)); }
    else {
      hipStreamSynchronize(s);
      hipMemcpy(h, d_stamps, sizeof(h), hipMemcpyDeviceToHost);
      for (int bz = 0; bz < 2; ++bz)
        for (int w = 0; w < 2; ++w) {
          fprintf(stderr, "[mbx stamps z=%d wave%d]", bz, w ? 3 : 0);
          for (int i = 1; i < 64 && h[(bz * 2 + w) * 64 + i]; ++i)
            fprintf(stderr, " %llu", h[(bz * 2 + w) * 64 + i] - h[(bz * 2 + w) * 64 + i - 1]);
          fprintf(stderr, "\n");
        }
      hipMemset(d_stamps, 0, sizeof(h));
    }
    a.stamps = d_stamps;
  }
  if (k == 3 && stride == 1) launch_mbx_ks<3, 1>(a, rows, s);
  else if (k == 3 && stride == 2) launch_mbx_ks<3, 2>(a, rows, s);
  else if (k == 5 && stride == 1) launch_mbx_ks<5, 1>(a, rows, s);
  else launch_mbx_ks<5, 2>(a, rows, s);
}

}  // namespace uda
