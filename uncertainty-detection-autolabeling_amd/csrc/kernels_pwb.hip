// 1x1 convolution on the bf16 matrix cores with float32-class accuracy: split-precision GEMM.
//
// gfx950 has no TF32/xf32 path and its f32-input MFMA runs at the f32 vector rate (64 FLOP/clk/SIMD,
// 1/16 of the bf16 rate), which left the deep 1x1 layers MFMA-bound far below the HBM roofline.  Here
// every f32 operand x is written as a sum of bf16 pieces, x = x0 + x1 (+ x2), each piece the
// round-to-nearest bf16 of what the previous pieces left over, and the product is accumulated in f32 on
// v_mfma_f32_32x32x16_bf16 from the significant cross terms:
//     PARTS = 2 ("x3"): a0*b0 + a0*b1 + a1*b0                       (rel. error per product ~2^-17)
//     PARTS = 3 ("x6"): a0*b0 + a0*b1 + a1*b0 + a0*b2 + a2*b0 + a1*b1   (~2^-24: float32-equivalent)
// Weights are split and laid out in MFMA B-fragment order ONCE on the host (uda_create); activations are
// split while they are staged through LDS, so HBM still holds plain float32 tensors.
// (reference: the Conv2D 1x1 / SeparableConv2D pointwise call sites, backbone/efficientnet_model.py:358-373,
//  403-418,471-486; efficientdet_keras.py:207-227,313-319,421-446,584-626.)
//
// Block = 4 waves (WM x WN), tile = (32*MT*WM) pixels x (32*NT*WN) output channels, K staged 32 deep:
//   A (pixels x k)  : 16-byte coalesced global loads (8 lanes = one 128-byte line of a pixel), optional SE
//                     gate, split, ds_write_b64 into per-piece images [row][32 k] with 80-byte rows
//                     (conflict-free ds_read_b128 of the 8-k operand fragments);
//   B (k x channels): the packed fragments are copied linearly (1 KiB per fragment) and read back linearly;
//   next chunk's global loads are in flight during the MFMAs of the current one (register prefetch);
//   epilogue        : accumulators through a wave-private LDS tile so that every lane stores 16 contiguous
//                     bytes, with bias / BN / swish / dropout keep-scale / residual applied on the float4.
#include <stdlib.h>
#include <string.h>

#include "uda_internal.h"

namespace uda {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float sigmoidf_b(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float swishf_b(float x) { return x * sigmoidf_b(x); }

// two floats -> packed bf16 pair (round to nearest even; v_cvt_pk_bf16_f32), element 0 in the low half
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf16_lo_f32(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf16_hi_f32(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

constexpr int PWB_BK = 32;         // k per staged chunk = 2 MFMA k-steps of 16
constexpr int PWB_AROW = 80;       // bytes per A image row: 32 bf16 + 16 pad
constexpr int PWB_STG = 68;        // epilogue staging row stride (floats)

template <int MT, int NT, int WM, int WN, int PARTS>
__global__ __launch_bounds__(256) void pwb_kernel(PwArgs a) {
  static_assert(WM * WN == 4, "four waves per block");
  constexpr int BM = 32 * MT * WM, NTB = NT * WN, BN = 32 * NTB;
  constexpr int A_BYTES = PARTS * BM * PWB_AROW;
  constexpr int B_BYTES = 2 * NTB * PARTS * 1024;
  constexpr int STG_BYTES = 4 * 32 * PWB_STG * 4;
  constexpr int LDS_BYTES = (A_BYTES + B_BYTES) > STG_BYTES ? (A_BYTES + B_BYTES) : STG_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  unsigned char* As = lds;                 // [PARTS][BM][80 B]
  uint4* Bs = (uint4*)(lds + A_BYTES);     // [2 k-steps][NTB][PARTS][64 lanes] x 16 B

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wr = wave / WN, wc = wave % WN;
  const int b = blockIdx.z, b_in = b / a.in_div;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int nt0 = blockIdx.y * NTB;
  const float* A = a.in + (size_t)b_in * a.HW * a.Cin;
  const float* se = a.se ? a.se + (size_t)b_in * a.Cin : nullptr;
  const uint4* Wp = (const uint4*)a.wsplit;
  const int KS = (a.Cin + 15) >> 4;        // MFMA k-steps in the packed weights
  const int NTL = (a.Cout + 31) >> 5;      // 32-column tiles in the packed weights

  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  constexpr int A_ITERS = BM * 8 / 256;
  constexpr int B_TOTAL = 2 * NTB * PARTS * 64;
  constexpr int B_ITERS = (B_TOTAL + 255) / 256;
  float4 ra[A_ITERS];
  uint4 rb[B_ITERS];
  float4 rg = make_float4(1.f, 1.f, 1.f, 1.f);

  auto load_chunk = [&](int k0) {
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
      const int f = tid + 256 * i;
      const int m = f >> 3, k = k0 + 4 * (f & 7);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m0 + m < a.HW && k < a.Cin) v = *(const float4*)(A + (size_t)(m0 + m) * a.Cin + k);
      ra[i] = v;
    }
    if (se) {
      const int k = k0 + 4 * (tid & 7);
      rg = (k < a.Cin) ? *(const float4*)(se + k) : make_float4(1.f, 1.f, 1.f, 1.f);
    }
    const int ks0 = k0 >> 4;
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) {
      const int f = tid + 256 * i;
      int q = f >> 6;
      const int part = q % PARTS; q /= PARTS;
      const int nt = q % NTB, ks = q / NTB;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (f < B_TOTAL && ks0 + ks < KS && nt0 + nt < NTL)
        v = Wp[(((size_t)(ks0 + ks) * NTL + (nt0 + nt)) * PARTS + part) * 64 + (f & 63)];
      rb[i] = v;
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
      const int f = tid + 256 * i;
      const int m = f >> 3, kq = f & 7;
      float r0 = ra[i].x * rg.x, r1 = ra[i].y * rg.y, r2 = ra[i].z * rg.z, r3 = ra[i].w * rg.w;
#pragma unroll
      for (int p = 0; p < PARTS; ++p) {
        const unsigned u0 = pack_bf16(r0, r1), u1 = pack_bf16(r2, r3);
        *(uint2*)(As + (size_t)(p * BM + m) * PWB_AROW + kq * 8) = make_uint2(u0, u1);
        if (p + 1 < PARTS) {
          r0 -= bf16_lo_f32(u0); r1 -= bf16_hi_f32(u0);
          r2 -= bf16_lo_f32(u1); r3 -= bf16_hi_f32(u1);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) {
      const int f = tid + 256 * i;
      if (f < B_TOTAL) Bs[f] = rb[i];
    }
  };

  load_chunk(0);
  for (int k0 = 0; k0 < a.Cin; k0 += PWB_BK) {
    store_chunk();
    __syncthreads();
    if (k0 + PWB_BK < a.Cin) load_chunk(k0 + PWB_BK);   // in flight during the MFMAs below
    const int nks = (a.Cin - k0 > 16) ? 2 : 1;
    for (int ks = 0; ks < nks; ++ks) {
      bf16x8 af[MT][PARTS];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int p = 0; p < PARTS; ++p)
          af[m][p] = *(const bf16x8*)(As + (size_t)(p * BM + (wr * MT + m) * 32 + li) * PWB_AROW + ks * 32 + lh * 16);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        bf16x8 bf[PARTS];
#pragma unroll
        for (int p = 0; p < PARTS; ++p)
          bf[p] = __builtin_bit_cast(bf16x8, Bs[((ks * NTB + wc * NT + n) * PARTS + p) * 64 + lane]);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          if constexpr (PARTS == 3) {
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m][2], bf[0], acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m][0], bf[2], acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m][1], bf[1], acc[m][n], 0, 0, 0);
          }
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m][1], bf[0], acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m][0], bf[1], acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m][0], bf[0], acc[m][n], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  const size_t out_base = (size_t)b * a.HW;
  const size_t res_base = a.res ? (size_t)(b / a.res_div) * a.HW : 0;

  if ((a.Cout & 3) != 0) {
    // scalar epilogue (class head: 9*7 = 63 channels): lane holds column li of 16 rows
#pragma unroll
    for (int m_ = 0; m_ < MT; ++m_)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int col = n0 + (wc * NT + n) * 32 + li;
        if (col >= a.Cout) continue;
        const float bias = a.bias ? a.bias[col] : 0.f;
        const float sc = a.bn_scale ? a.bn_scale[col] : 1.f;
        const float sh = a.bn_scale ? a.bn_shift[col] : 0.f;
        const float mk = a.mask ? a.mask[(size_t)b * a.Cout + col] : 1.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + (wr * MT + m_) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m >= a.HW) continue;
          float v = acc[m_][n][r] + bias;
          v = fmaf(v, sc, sh);
          if (a.act == UDA_ACT_SWISH) v = swishf_b(v);
          v *= mk;
          if (a.res) v += a.res[(res_base + m) * a.Cout + col];
          a.out[(out_base + m) * a.Cout + col] = v;
        }
      }
    return;
  }

  // vector epilogue through a wave-private staging tile [32 rows][64 cols (+4 pad)]
  float* stg = (float*)lds + wave * 32 * PWB_STG;
  const int rrow = lane >> 4, c4 = lane & 15;   // read-back: 16 lanes per row, 4 rows per pass
#pragma unroll
  for (int m_ = 0; m_ < MT; ++m_) {
#pragma unroll
    for (int p = 0; p < (NT + 1) / 2; ++p) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int n = 2 * p + q;
        if (n < NT) {
#pragma unroll
          for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * lh) * PWB_STG + q * 32 + li] = acc[m_][n][r];
        }
      }
      __syncthreads();
      const int lcol = (wc * NT + 2 * p) * 32 + 4 * c4;     // column inside the block tile
      const int col = n0 + lcol;
      const bool colok = (col < a.Cout) && (2 * p * 32 + 4 * c4 < NT * 32);
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
        const int m = m0 + (wr * MT + m_) * 32 + row;
        if (colok && m < a.HW) {
          float4 v = *(const float4*)(stg + row * PWB_STG + 4 * c4);
          v.x = fmaf(v.x + bias.x, sc.x, sh.x);
          v.y = fmaf(v.y + bias.y, sc.y, sh.y);
          v.z = fmaf(v.z + bias.z, sc.z, sh.z);
          v.w = fmaf(v.w + bias.w, sc.w, sh.w);
          if (a.act == UDA_ACT_SWISH) {
            v.x = swishf_b(v.x); v.y = swishf_b(v.y); v.z = swishf_b(v.z); v.w = swishf_b(v.w);
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
}

template <int MT, int NT, int WM, int WN>
static void launch_pwb_cfg(const PwArgs& a, int rows, hipStream_t s) {
  constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
  const dim3 grid((a.HW + BM - 1) / BM, (a.Cout + BN - 1) / BN, rows), block(256);
  if (a.wparts == 3) hipLaunchKernelGGL((pwb_kernel<MT, NT, WM, WN, 3>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((pwb_kernel<MT, NT, WM, WN, 2>), grid, block, 0, s, a);
}

// tile = 128 pixels x {32, 64, 96, 128} channels (four waves stacked along the pixels for narrow outputs,
// 2 x 2 waves of 64 x 64 for wide ones)
void launch_pwb(const PwArgs& a, int rows, hipStream_t s) {
  static int force = -1;
  if (force < 0) { const char* e = getenv("UDA_PWB_CFG"); force = e ? atoi(e) : 0; }
  int cfg = force;
  if (cfg == 0) cfg = a.Cout <= 32 ? 1 : (a.Cout <= 64 ? 2 : (a.Cout <= 96 ? 3 : 4));
  switch (cfg) {
    case 1: launch_pwb_cfg<1, 1, 4, 1>(a, rows, s); break;
    case 2: launch_pwb_cfg<1, 2, 4, 1>(a, rows, s); break;
    case 3: launch_pwb_cfg<1, 3, 4, 1>(a, rows, s); break;
    case 5: launch_pwb_cfg<2, 4, 2, 2>(a, rows, s); break;
    default: launch_pwb_cfg<2, 2, 2, 2>(a, rows, s); break;
  }
}

// ---------------------------------------------------------------- host-side weight split / packing
static inline uint16_t f32_to_bf16_rne(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);     // weights are finite: no NaN handling needed
  return (uint16_t)(u >> 16);
}
static inline float bf16_to_f32(uint16_t h) {
  const uint32_t u = (uint32_t)h << 16;
  float x;
  memcpy(&x, &u, 4);
  return x;
}

size_t pwb_packed_elems(int K, int N, int parts) {
  return (size_t)((K + 15) / 16) * ((N + 31) / 32) * parts * 64 * 8;
}

// w [K][N] float32 -> fragments [k-step][32-col tile][part][lane][8] bf16 of v_mfma_f32_32x32x16_bf16's B operand:
// lane l holds B[16 s + 8 (l >> 5) + e][32 j + (l & 31)], e = 0..7
void pwb_pack_weights(const float* w, int K, int N, int parts, uint16_t* out) {
  const int KS = (K + 15) / 16, NTL = (N + 31) / 32;
  for (int s = 0; s < KS; ++s)
    for (int j = 0; j < NTL; ++j)
      for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 8; ++e) {
          const int k = 16 * s + 8 * (l >> 5) + e, n = 32 * j + (l & 31);
          float r = (k < K && n < N) ? w[(size_t)k * N + n] : 0.f;
          for (int p = 0; p < parts; ++p) {
            const uint16_t h = f32_to_bf16_rne(r);
            out[((((size_t)s * NTL + j) * parts + p) * 64 + l) * 8 + e] = h;
            r -= bf16_to_f32(h);
          }
        }
}

}  // namespace uda
