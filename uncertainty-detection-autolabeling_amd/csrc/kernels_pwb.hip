// 1x1 convolution on the bf16 matrix cores with float32-class accuracy: split-precision GEMM.
//
// gfx950 has no TF32/xf32 path and its f32-input MFMA runs at the f32 vector rate (64 FLOP/clk/SIMD,
// 1/16 of the bf16 rate), which left the deep 1x1 layers MFMA-bound far below the HBM roofline.  Here
// every f32 operand x is written as a sum of bf16 pieces, x = x0 + x1 (+ x2), each piece the
// round-to-nearest bf16 of what the previous pieces left over, and the product is accumulated in f32 on
// v_mfma_f32_32x32x16_bf16 from the significant cross terms:
//     PARTS = 2 (bf16 x2): a0*b0 + a0*b1 + a1*b0                       (rel. error per product ~2^-17)
//     PARTS = 3 (bf16 x3): a0*b0 + a0*b1 + a1*b0 + a0*b2 + a2*b0 + a1*b1   (~2^-24: float32-equivalent)
//     PARTS = 4 (fp16 x2): the three terms of PARTS = 2 on v_mfma_f32_32x32x16_f16 with 11-bit pieces (~2^-22; the default:
//                          float32-class accuracy at the matrix-core cost of the three-term scheme, mfma_common.h)
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
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "mfma_common.h"

namespace uda {

// DEEP (round 5): K-heavy contractions (1152 -> 192 / 320 at 24 x 40: 36 chunks per block) ran one 16 KB chunk per block on its
// way - a chunk costs a memory latency, not its matrix time.  The chunk loop is unrolled by two with TWO register sets, so that
// the chunk after next is requested before the current one is multiplied: the order of the outstanding loads is the same at the
// loop header from the prologue and from the back edge (set 0 older, set 1 younger), which is what lets the compiler wait for
// a set with vmcnt(loads of the other set) instead of vmcnt(0).
template <int MT, int NT, int WM, int WN, int PARTS, int OCC, bool DEEP = false>
__global__ __launch_bounds__(256, OCC) void pwb_kernel(PwArgs a) {
  static_assert(WM * WN == 4, "four waves per block");
  constexpr int NPC = split_np(PARTS);     // pieces per operand (PARTS names the scheme: UDA_SPLIT_*)
  constexpr int BM = 32 * MT * WM, NTB = NT * WN, BN = 32 * NTB;
  constexpr int A_BYTES = NPC * BM * PWB_AROW;
  constexpr int B_BYTES = 2 * NTB * NPC * 1024;
  constexpr int STG_BYTES = 4 * 32 * PWB_STG * 4;
  constexpr int LDS_BYTES = (A_BYTES + B_BYTES) > STG_BYTES ? (A_BYTES + B_BYTES) : STG_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  unsigned char* As = lds;                 // [NPC][BM][80 B]
  uint4* Bs = (uint4*)(lds + A_BYTES);     // [2 k-steps][NTB][NPC][64 lanes] x 16 B

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wr = wave / WN, wc = wave % WN;
  // (an XCD-aware block order that keeps the column blocks of a pixel tile on one XCD's L2 was measured: no gain)
  const int bx = blockIdx.x, by = blockIdx.y;
  const int b = blockIdx.z, b_in = b / a.in_div;
  const int m0 = bx * BM, n0 = by * BN;
  const int nt0 = by * NTB;
  const float* A = a.in + (size_t)b_in * a.HW * a.Cin;
  const float* se = a.se ? a.se + (size_t)(b / a.se_div) * a.Cin : nullptr;
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
  constexpr int B_TOTAL = 2 * NTB * NPC * 64;
  constexpr int B_ITERS = (B_TOTAL + 255) / 256;
  struct ChunkRegs { float4 ra[A_ITERS]; uint4 rb[B_ITERS]; float4 rg; };       // one chunk's operands on their way
  ChunkRegs c0, c1;
  c0.rg = make_float4(1.f, 1.f, 1.f, 1.f);
  c1.rg = c0.rg;
  float amax = 0.f;                        // fp16 pieces: largest operand magnitude this lane has split

  auto load_chunk = [&](int k0, ChunkRegs& cr) {
    float4* ra = cr.ra; uint4* rb = cr.rb; float4& rg = cr.rg;
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
      const int f = tid + 256 * i;
      const int m = f >> 3, k = k0 + 4 * (f & 7);
      if constexpr (DEEP) {
        // branch-free (the launcher checked Cin % 32 == 0 and whole column tiles): a load under a lane condition sits in a
        // basic block of its own and the compiler then waits for EVERYTHING at the first use (vmcnt(0)).  Rows past the map
        // read the last row: row m of A only reaches row m of the result, which is never stored.
        const int mr = m0 + m < a.HW ? m0 + m : a.HW - 1;
        ra[i] = *(const float4*)(A + (size_t)mr * a.Cin + k);
      } else {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m0 + m < a.HW && k < a.Cin) v = *(const float4*)(A + (size_t)(m0 + m) * a.Cin + k);
        ra[i] = v;
      }
    }
    if (se) {
      const int k = k0 + 4 * (tid & 7);
      if constexpr (DEEP) rg = *(const float4*)(se + k);
      else rg = (k < a.Cin) ? *(const float4*)(se + k) : make_float4(1.f, 1.f, 1.f, 1.f);
    }
    const int ks0 = k0 >> 4;
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) {
      const int f = tid + 256 * i;
      int q = f >> 6;
      const int part = q % NPC; q /= NPC;
      const int nt = q % NTB, ks = q / NTB;
      if constexpr (DEEP && B_TOTAL % 256 == 0) {
        rb[i] = Wp[(((size_t)(ks0 + ks) * NTL + (nt0 + nt)) * NPC + part) * 64 + (f & 63)];
      } else {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (f < B_TOTAL && ks0 + ks < KS && nt0 + nt < NTL)
          v = Wp[(((size_t)(ks0 + ks) * NTL + (nt0 + nt)) * NPC + part) * 64 + (f & 63)];
        rb[i] = v;
      }
    }
  };
  auto store_chunk = [&](const ChunkRegs& cr) {
    const float4* ra = cr.ra; const uint4* rb = cr.rb; const float4 rg = cr.rg;
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
      const int f = tid + 256 * i;
      const int m = f >> 3, kq = f & 7;
      float r0 = ra[i].x * rg.x, r1 = ra[i].y * rg.y, r2 = ra[i].z * rg.z, r3 = ra[i].w * rg.w;
      split_track<PARTS>(amax, r0, r1);
      split_track<PARTS>(amax, r2, r3);
#pragma unroll
      for (int p = 0; p < NPC; ++p) {
        const unsigned u0 = pack_piece<PARTS>(r0, r1), u1 = pack_piece<PARTS>(r2, r3);
        *(uint2*)(As + (size_t)(p * BM + m) * PWB_AROW + kq * 8) = make_uint2(u0, u1);
        if (p + 1 < NPC) {
          r0 -= piece_lo<PARTS>(u0); r1 -= piece_hi<PARTS>(u0);
          r2 -= piece_lo<PARTS>(u1); r3 -= piece_hi<PARTS>(u1);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) {
      const int f = tid + 256 * i;
      if (f < B_TOTAL) Bs[f] = rb[i];
    }
  };

  auto mma_chunk = [&](int k0) {
    const int nks = (a.Cin - k0 > 16) ? 2 : 1;
    for (int ks = 0; ks < nks; ++ks) {
      bf16x8 af[MT][NPC];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int p = 0; p < NPC; ++p)
          af[m][p] = *(const bf16x8*)(As + (size_t)(p * BM + (wr * MT + m) * 32 + li) * PWB_AROW + ks * 32 + lh * 16);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        bf16x8 bf[NPC];
#pragma unroll
        for (int p = 0; p < NPC; ++p)
          bf[p] = __builtin_bit_cast(bf16x8, Bs[((ks * NTB + wc * NT + n) * NPC + p) * 64 + lane]);
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m][n] = mfma_terms<PARTS>(af[m], bf, acc[m][n]);
      }
    }
  };
  if constexpr (DEEP) {
    load_chunk(0, c0);
    if (PWB_BK < a.Cin) load_chunk(PWB_BK, c1);
    for (int k0 = 0; k0 < a.Cin; k0 += 2 * PWB_BK) {
      store_chunk(c0);
      __syncthreads();
      if (k0 + 2 * PWB_BK < a.Cin) load_chunk(k0 + 2 * PWB_BK, c0);     // two chunks ahead
      mma_chunk(k0);
      __syncthreads();
      if (k0 + PWB_BK < a.Cin) {
        store_chunk(c1);
        __syncthreads();
        if (k0 + 3 * PWB_BK < a.Cin) load_chunk(k0 + 3 * PWB_BK, c1);
        mma_chunk(k0 + PWB_BK);
        __syncthreads();
      }
    }
  } else {
    load_chunk(0, c0);
    for (int k0 = 0; k0 < a.Cin; k0 += PWB_BK) {
      store_chunk(c0);
      __syncthreads();
      if (k0 + PWB_BK < a.Cin) load_chunk(k0 + PWB_BK, c0);   // in flight during the MFMAs below
      mma_chunk(k0);
      __syncthreads();
    }
  }
  split_report<PARTS>(amax, a.oor);

  const size_t out_base = (size_t)b * a.HW;
  const float un = a.wunscale;             // the packed weights carry a power-of-two factor 1 / un (fp16 pieces; else 1)
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
          float v = fmaf(acc[m_][n][r], un, bias);
          v = fmaf(v, sc, sh);
          if (a.act == UDA_ACT_SWISH) v = swishf_b(v);
          else if (a.act >= UDA_ACT_RELU) v = act_relu_family(v, a.act);
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
      // Loads first, all of them and unconditionally (dead lanes read a clamped, valid address), then the arithmetic, then
      // the eight stores back to back.  Written the obvious way - per row group: staged values, residual load, store, all
      // under the lane's bounds condition - every row group began with s_waitcnt vmcnt(0): the residual load is younger
      // than the previous group's store and memory operations retire in order, and even without a residual the compiler
      // re-waits for the epilogue parameters at each conditional use.  Either way a wave had ONE store in flight.
      const int colc = colok ? col : 0;
      float4 bias = make_float4(0.f, 0.f, 0.f, 0.f), sc = make_float4(1.f, 1.f, 1.f, 1.f);
      float4 sh = make_float4(0.f, 0.f, 0.f, 0.f), mk = make_float4(1.f, 1.f, 1.f, 1.f);
      if (a.bias) bias = *(const float4*)(a.bias + colc);
      if (a.bn_scale) {
        sc = *(const float4*)(a.bn_scale + colc);
        sh = *(const float4*)(a.bn_shift + colc);
      }
      if (a.mask) mk = *(const float4*)(a.mask + (size_t)b * a.Cout + colc);
      const int mrow0 = m0 + (wr * MT + m_) * 32 + rrow;
      float4 rr[8], v[8];
      if (a.res) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int mc = mrow0 + it * 4 < a.HW ? mrow0 + it * 4 : a.HW - 1;
          rr[it] = *(const float4*)(a.res + (res_base + mc) * a.Cout + colc);
        }
      }
#pragma unroll
      for (int it = 0; it < 8; ++it) v[it] = *(const float4*)(stg + (it * 4 + rrow) * PWB_STG + 4 * c4);
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        float4 t = v[it];
        t.x = fmaf(fmaf(t.x, un, bias.x), sc.x, sh.x);
        t.y = fmaf(fmaf(t.y, un, bias.y), sc.y, sh.y);
        t.z = fmaf(fmaf(t.z, un, bias.z), sc.z, sh.z);
        t.w = fmaf(fmaf(t.w, un, bias.w), sc.w, sh.w);
        if (a.act == UDA_ACT_SWISH) {
          t.x = swishf_b(t.x); t.y = swishf_b(t.y); t.z = swishf_b(t.z); t.w = swishf_b(t.w);
        } else if (a.act >= UDA_ACT_RELU) {
          t.x = act_relu_family(t.x, a.act); t.y = act_relu_family(t.y, a.act); t.z = act_relu_family(t.z, a.act); t.w = act_relu_family(t.w, a.act);
        }
        t.x *= mk.x; t.y *= mk.y; t.z *= mk.z; t.w *= mk.w;
        if (a.res) { t.x += rr[it].x; t.y += rr[it].y; t.z += rr[it].z; t.w += rr[it].w; }
        v[it] = t;
      }
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int m = mrow0 + it * 4;
        if (colok && m < a.HW) *(float4*)(a.out + (out_base + m) * a.Cout + col) = v[it];
      }
      __syncthreads();
    }
  }
}

// 1x1 conv whose INPUT is shared by the `in_div` sample rows of an image (block 0's projection under full MC
// dropout: the depthwise output exists once per image, the per-sample SE gate x dropout scale multiplies it on the
// way in).  A block keeps its 128 x K (K <= 32) input tile in registers and loops over the samples: gate, split,
// MFMA, epilogue per sample -> the shared tensor is read once instead of in_div times.  Everything is wave-private
// (each wave owns 32 pixel rows: its slice of the A image and its staging tile), so there is no block barrier.
__global__ __launch_bounds__(256, 3) void pwb_shared_kernel(PwArgs a) {
  constexpr int WAVE_BYTES = 32 * PWB_STG * 4;          // staging tile (8.5 KB) >= 2 x 32 x 80 B of the A image
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * WAVE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  unsigned char* Aw = lds + wave * WAVE_BYTES;           // [2 parts][32 rows][80 B]
  float* stg = (float*)Aw;
  const int b_in = blockIdx.z;
  const int m0 = blockIdx.x * 128 + wave * 32;
  const float* A = a.in + (size_t)b_in * a.HW * a.Cin;
  const uint4* Wp = (const uint4*)a.wsplit;
  const int NTL = (a.Cout + 31) >> 5;                    // == 1 (Cout <= 32)
  // raw input rows of this wave: lane -> (row = lane / 8 + 8 i, k quad = lane % 8)
  float4 ra[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = (lane >> 3) + 8 * i, k = 4 * (lane & 7);
    ra[i] = (m0 + m < a.HW && k < a.Cin) ? *(const float4*)(A + (size_t)(m0 + m) * a.Cin + k) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // weight fragments of both k-steps, both pieces: straight from the packed image into registers
  bf16x8 bf[2][2];
  const int KS = (a.Cin + 15) >> 4;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int p = 0; p < 2; ++p)
      bf[ks][p] = __builtin_bit_cast(bf16x8, ks < KS ? Wp[(((size_t)ks * NTL) * 2 + p) * 64 + lane] : make_uint4(0u, 0u, 0u, 0u));
  const int rrow = lane >> 4, c4 = lane & 15;
  const int col = 4 * c4;
  const bool colok = col < a.Cout;
  float4 bias = make_float4(0.f, 0.f, 0.f, 0.f), sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (colok && (a.Cout & 3) == 0) {
    if (a.bias) bias = *(const float4*)(a.bias + col);
    if (a.bn_scale) { sc = *(const float4*)(a.bn_scale + col); sh = *(const float4*)(a.bn_shift + col); }
  }

  for (int t = 0; t < a.in_div; ++t) {
    const int b = b_in * a.in_div + t;
    float4 rg = make_float4(1.f, 1.f, 1.f, 1.f);
    if (a.se) {
      const int k = 4 * (lane & 7);
      if (k < a.Cin) rg = *(const float4*)(a.se + (size_t)(b / a.se_div) * a.Cin + k);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = (lane >> 3) + 8 * i, kq = lane & 7;
      float r0 = ra[i].x * rg.x, r1 = ra[i].y * rg.y, r2 = ra[i].z * rg.z, r3 = ra[i].w * rg.w;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const unsigned u0 = pack_bf16(r0, r1), u1 = pack_bf16(r2, r3);
        *(uint2*)(Aw + (size_t)(p * 32 + m) * PWB_AROW + kq * 8) = make_uint2(u0, u1);
        if (p == 0) {
          r0 -= bf16_lo_f32(u0); r1 -= bf16_hi_f32(u0);
          r2 -= bf16_lo_f32(u1); r3 -= bf16_hi_f32(u1);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8 a0 = *(const bf16x8*)(Aw + (size_t)(0 * 32 + li) * PWB_AROW + ks * 32 + lh * 16);
      const bf16x8 a1 = *(const bf16x8*)(Aw + (size_t)(1 * 32 + li) * PWB_AROW + ks * 32 + lh * 16);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bf[ks][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bf[ks][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bf[ks][0], acc, 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();     // the A image of this wave has been read: the staging tile may overwrite it
#pragma unroll
    for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * lh) * PWB_STG + li] = acc[r];
    __builtin_amdgcn_wave_barrier();
    float4 mk = make_float4(1.f, 1.f, 1.f, 1.f);
    if (colok && a.mask) mk = *(const float4*)(a.mask + (size_t)b * a.Cout + col);
    const size_t out_base = (size_t)b * a.HW;
    const size_t res_base = a.res ? (size_t)(b / a.res_div) * a.HW : 0;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = it * 4 + rrow;
      const int m = m0 + row;
      if (colok && m < a.HW) {
        float4 v = *(const float4*)(stg + row * PWB_STG + col);
        v.x = fmaf(v.x + bias.x, sc.x, sh.x);
        v.y = fmaf(v.y + bias.y, sc.y, sh.y);
        v.z = fmaf(v.z + bias.z, sc.z, sh.z);
        v.w = fmaf(v.w + bias.w, sc.w, sh.w);
        if (a.act == UDA_ACT_SWISH) {
          v.x = swishf_b(v.x); v.y = swishf_b(v.y); v.z = swishf_b(v.z); v.w = swishf_b(v.w);
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
    __builtin_amdgcn_wave_barrier();     // the staging tile has been read: the next sample's A image may overwrite it
  }
}

// ---------------------------------------------------------------- 1x1 convolution, few pixels x many input channels
// A projection at the bottom of the backbone in a serve of ONE image under head-only MC dropout (the reference's shipped
// inference configuration): 960 pixels x 1152 -> 192 channels.  pwb_kernel covers that with 16 blocks that each walk 36 chunks,
// one (DEEP: two) memory round trips at a time - 55 us for 0.4 GFLOP, a chain of latencies on an empty device
// (profiles/r05_batch1_headonly_per_op.txt).  A split over K would fill the device but change the float32 summation order with
// the number of rows; this kernel keeps the order - the same cross terms, k-step after k-step, into the same accumulator
// layout, hence results bit-identical to pwb_kernel's - and shortens the chain instead: one WAVE per 32-pixel x (32 NT)-channel
// tile, no LDS, no barrier; the wave reads its operand fragments straight into registers (lane = pixel, 8 consecutive
// channels; weight fragments as packed on the host) and keeps PWS_D k-steps on their way while it multiplies the PWS_D before.
// The input tile is re-read by every column tile (through the L2): only for launches of a few thousand pixels (launch_pws).
#ifndef UDA_PWS_D
#define UDA_PWS_D 4
#endif
constexpr int PWS_D = UDA_PWS_D;
template <int PARTS, int NT, bool GATE>
__global__ __launch_bounds__(64) void pws_kernel(PwArgs a) {
  constexpr int NPC = split_np(PARTS);
  const int lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * 32, nt0 = blockIdx.y * NT;
  const int b = blockIdx.z, b_in = b / a.in_div;
  const int KS = a.Cin >> 4;                 // (launcher: Cin % 16 == 0, KS >= PWS_D)
  const int NTL = (a.Cout + 31) >> 5;
  const int mr = m0 + li < a.HW ? m0 + li : a.HW - 1;       // (rows past the map: row m of A only reaches row m of the result, never stored)
  const float* arow = a.in + ((size_t)b_in * a.HW + mr) * a.Cin + 8 * lh;
  const float* se = GATE ? a.se + (size_t)(b / a.se_div) * a.Cin + 8 * lh : nullptr;
  const uint4* Wp = (const uint4*)a.wsplit + lane;
  size_t woff[NT];                           // (a column tile past the end repeats the last one; not stored)
#pragma unroll
  for (int n = 0; n < NT; ++n) woff[n] = (size_t)(nt0 + n < NTL ? nt0 + n : NTL - 1) * NPC * 64;
  const size_t wks = (size_t)NTL * NPC * 64;       // uint4 per k-step of the packed weights

  f32x16 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  float amax = 0.f;

  struct Step { float4 a0, a1, g0, g1; uint4 w[NT][NPC]; };
  struct Group { Step s[PWS_D]; };
  // plain, unconditional loads (a load under a condition gets a basic block of its own and a vmcnt(0) at its first use)
  auto load_step = [&](int ks, Step& t) {
    t.a0 = *(const float4*)(arow + 16 * ks);
    t.a1 = *(const float4*)(arow + 16 * ks + 4);
    if constexpr (GATE) {
      t.g0 = *(const float4*)(se + 16 * ks);
      t.g1 = *(const float4*)(se + 16 * ks + 4);
    }
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int p = 0; p < NPC; ++p) t.w[n][p] = Wp[(size_t)ks * wks + woff[n] + p * 64];
  };
  auto mma_step = [&](const Step& t) {
    float4 v0 = t.a0, v1 = t.a1;
    if constexpr (GATE) {
      v0.x *= t.g0.x; v0.y *= t.g0.y; v0.z *= t.g0.z; v0.w *= t.g0.w;
      v1.x *= t.g1.x; v1.y *= t.g1.y; v1.z *= t.g1.z; v1.w *= t.g1.w;
    }
    bf16x8 af[NPC];
    split_parts<PARTS>(v0, v1, af, amax);
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      bf16x8 bf[NPC];
#pragma unroll
      for (int p = 0; p < NPC; ++p) bf[p] = __builtin_bit_cast(bf16x8, t.w[n][p]);
      acc[n] = mfma_terms<PARTS>(af, bf, acc[n]);
    }
  };
  auto load_group = [&](int g, Group& G) {
#pragma unroll
    for (int d = 0; d < PWS_D; ++d) load_step(g * PWS_D + d, G.s[d]);
  };
  auto mma_group = [&](const Group& G) {
#pragma unroll
    for (int d = 0; d < PWS_D; ++d) mma_step(G.s[d]);
  };
  // whole groups of PWS_D k-steps, two sets of registers: the group after next is requested before the current one is
  // multiplied (a group index past the end reloads the last whole group - never multiplied); then the k-steps that are left
  const int ng = KS / PWS_D;
  Group c0, c1;
  load_group(0, c0);
  load_group(ng > 1 ? 1 : 0, c1);
#pragma unroll 1
  for (int g = 0; g < ng; g += 2) {
    mma_group(c0);
    load_group(g + 2 < ng ? g + 2 : ng - 1, c0);
    if (g + 1 < ng) mma_group(c1);          // (uniform)
    load_group(g + 3 < ng ? g + 3 : ng - 1, c1);
  }
#pragma unroll 1
  for (int ks = ng * PWS_D; ks < KS; ++ks) {
    Step t;
    load_step(ks, t);
    mma_step(t);
  }
  split_report<PARTS>(amax, a.oor);

  // epilogue: lane holds column li of 16 rows (the arithmetic of pwb_kernel's epilogues, element by element)
  const size_t out_base = (size_t)b * a.HW;
  const float un = a.wunscale;
  const size_t res_base = a.res ? (size_t)(b / a.res_div) * a.HW : 0;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int col = (nt0 + n) * 32 + li;
    if (nt0 + n >= NTL || col >= a.Cout) continue;
    const float bias = a.bias ? a.bias[col] : 0.f;
    const float sc = a.bn_scale ? a.bn_scale[col] : 1.f;
    const float sh = a.bn_scale ? a.bn_shift[col] : 0.f;
    const float mk = a.mask ? a.mask[(size_t)b * a.Cout + col] : 1.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m >= a.HW) continue;
      float v = fmaf(acc[n][r], un, bias);
      v = fmaf(v, sc, sh);
      if (a.act == UDA_ACT_SWISH) v = swishf_b(v);
      else if (a.act >= UDA_ACT_RELU) v = act_relu_family(v, a.act);
      v *= mk;
      if (a.res) v += a.res[(res_base + m) * a.Cout + col];
      a.out[(out_base + m) * a.Cout + col] = v;
    }
  }
}

// few pixels x many input channels -> pws_kernel (see there); 0 = not its case
static int launch_pws(const PwArgs& a, int rows, hipStream_t s) {
  // Where it pays (tools/debug/pw_skinny_sweep.py, us, skinny / tiled): 960 px 1152 -> 192: 17 / 52; 3840 px 672 -> 112: 16 / 32;
  // 9600 px 1152 -> 192: 48 / 56; 9600 px 1152 -> 320: 90 / 56; 38400 px 672 -> 112: 88 / 44 - every column tile re-reads the
  // input tile, so the measure is pixels x column tiles.  UDA_PW_SKINNY = that product up to which it is used (0 = never).
  static long long max_work = -1;
  if (max_work < 0) { const char* e = getenv("UDA_PW_SKINNY"); max_work = e ? atoll(e) : 60000; }
  if ((long long)rows * a.HW * ((a.Cout + 31) / 32) > max_work || a.Cin < 256 || (a.Cin & 15) || !a.wsplit || a.wparts == UDA_SPLIT_NONE) return 0;
  const int row_tiles = (a.HW + 31) / 32, NTL = (a.Cout + 31) / 32;
  // one column tile per wave while that still leaves fewer waves than the device has SIMDs, two otherwise
  const int nt = ((long long)row_tiles * rows * NTL <= 1024) ? 1 : 2;
  const dim3 grid(row_tiles, (NTL + nt - 1) / nt, rows), block(64);
  auto go = [&](auto k1, auto k2) { if (nt == 1) hipLaunchKernelGGL(k1, grid, block, 0, s, a); else hipLaunchKernelGGL(k2, grid, block, 0, s, a); };
  if (a.wparts == UDA_SPLIT_F16X2) { if (a.se) go(pws_kernel<4, 1, true>, pws_kernel<4, 2, true>); else go(pws_kernel<4, 1, false>, pws_kernel<4, 2, false>); }
  else if (a.wparts == UDA_SPLIT_BF16X3) { if (a.se) go(pws_kernel<3, 1, true>, pws_kernel<3, 2, true>); else go(pws_kernel<3, 1, false>, pws_kernel<3, 2, false>); }
  else if (a.wparts == UDA_SPLIT_BF16X2) { if (a.se) go(pws_kernel<2, 1, true>, pws_kernel<2, 2, true>); else go(pws_kernel<2, 1, false>, pws_kernel<2, 2, false>); }
  else return 0;
  return 1;
}

template <int MT, int NT, int WM, int WN>
static void launch_pwb_cfg(const PwArgs& a, int rows, hipStream_t s) {
  constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
  const dim3 grid((a.HW + BM - 1) / BM, (a.Cout + BN - 1) / BN, rows), block(256);
  // three blocks per CU (<= 168 registers): measured 10-20 % faster than the unconstrained allocation (2 per CU)
  if (a.wparts == UDA_SPLIT_BF16X3) hipLaunchKernelGGL((pwb_kernel<MT, NT, WM, WN, 3, 2>), grid, block, 0, s, a);
  else if (a.wparts == UDA_SPLIT_F16X2) {
    // (K-heavy projections on the big tiles - two blocks per CU by registers anyway: two chunks in flight per block)
    static const bool deep_on = !(getenv("UDA_PWB_DEEP") && atoi(getenv("UDA_PWB_DEEP")) == 0);
    if constexpr (MT * NT > 4) {
      if (deep_on && a.Cin >= 384 && a.Cin % PWB_BK == 0 && a.Cout % BN == 0 && (2 * NT * WN * 2 * 64) % 256 == 0) { hipLaunchKernelGGL((pwb_kernel<MT, NT, WM, WN, 4, 2, true>), grid, block, 0, s, a); return; }
    }
    hipLaunchKernelGGL((pwb_kernel<MT, NT, WM, WN, 4, (MT * NT > 4 ? 2 : 3)>), grid, block, 0, s, a);
  }
  else hipLaunchKernelGGL((pwb_kernel<MT, NT, WM, WN, 2, (MT * NT > 4 ? 2 : 3)>), grid, block, 0, s, a);   // (5-6 accumulator tiles per wave: 168 registers spill)
}

// tile = 128 pixels x {32, 64, 96, 128, 160, 192} channels (four waves stacked along the pixels for narrow outputs and
// for 160, 2 x 2 waves of 64 x 64 / 64 x 96 for 128 / 192)
void launch_pwb(const PwArgs& a, int rows, hipStream_t s) {
  static int shared = -1;
  if (shared < 0) { const char* e = getenv("UDA_PW_SHARED"); shared = e ? atoi(e) : 1; }
  if (shared && a.in_div > 1 && a.Cin <= 32 && a.Cout <= 32 && (a.Cout & 3) == 0 && a.wparts == 2 && rows % a.in_div == 0) {
    const dim3 grid((a.HW + 127) / 128, 1, rows / a.in_div);
    hipLaunchKernelGGL(pwb_shared_kernel, grid, dim3(256), 0, s, a);
    return;
  }
  if (launch_pws(a, rows, s)) return;
  static int force = -1;
  if (force < 0) { const char* e = getenv("UDA_PWB_CFG"); force = e ? atoi(e) : 0; }
  static int wide = -1;
  if (wide < 0) { const char* e = getenv("UDA_PWB_WIDE"); wide = e ? atoi(e) : 1; }
  int cfg = force;
  if (cfg == 0) {
    cfg = a.Cout <= 32 ? 1 : (a.Cout <= 64 ? 2 : (a.Cout <= 96 ? 3 : 4));
    if (wide && a.Cout > 128) {
      // wide outputs: the column tile (128 / 160 / 192) that leaves the fewest idle columns in the last column block, the
      // wider one on a tie - 192 channels: one block of 192 instead of 128 + 64 (a quarter of the MFMAs idle, the input
      // tile read and split twice); 320: 2 x 160; 672: 4 x 192 instead of 6 x 128
      const int bn[3] = {128, 160, 192}, id[3] = {4, 6, 5};
      int best = 1 << 30;
      for (int i = 0; i < 3; ++i) {
        const int waste = (a.Cout + bn[i] - 1) / bn[i] * bn[i] - a.Cout;
        if (waste <= best) { best = waste; cfg = id[i]; }
      }
    }
  }
  switch (cfg) {
    case 1: launch_pwb_cfg<1, 1, 4, 1>(a, rows, s); break;
    case 2: launch_pwb_cfg<1, 2, 4, 1>(a, rows, s); break;
    case 3: launch_pwb_cfg<1, 3, 4, 1>(a, rows, s); break;
    case 5: launch_pwb_cfg<2, 3, 2, 2>(a, rows, s); break;
    case 6: launch_pwb_cfg<1, 5, 4, 1>(a, rows, s); break;
    default: launch_pwb_cfg<2, 2, 2, 2>(a, rows, s); break;
  }
}

// ---------------------------------------------------------------- fused MBConv front half, split-bf16 expand
// expand 1x1 + BN + swish + dropout -> depthwise kxk / stride s (TF SAME) + BN + swish + dropout + SE tile
// sums, one output tile per block, 32 expanded channels at a time (backbone/efficientnet_model.py:446-464).
//   A  : every wave keeps the split-bf16 operand fragments of ITS 32-pixel slices of the input tile (with
//        halo) in registers for the whole block: the tile is read from HBM/L2 once, straight into fragment
//        shape (lane = pixel, 8 consecutive channels), no LDS image of the input.
//   B  : the expand kernel times the BN scale, plus one extra k row holding the BN shift, packed on the host
//        in fragment order.  The matching extra input channel is 1 inside the image and 0 in the halo /
//        padding, so a position outside the image gives exactly swish(0) = 0 - the zero padding TF applies to
//        the depthwise INPUT - with no per-element test.
//   E  : the expanded + activated 32-channel slab [pixel][33] in LDS; the depthwise stage reads it with a
//        sliding window along x, writes the output tile (128-byte channel segments) and the SE tile sums.
// Two barriers per 32-channel slab (E complete / E free); the SE reduction rides on the second one.
// E transposed (5x5 stride 1, every fused MBConv variant): the 32-channel slab is kept [channel][pixel] with a channel pitch
// of 260 floats instead of [pixel][channel].  The accumulator of the expand MFMA holds 4 x 4 CONSECUTIVE pixels of
// one channel per lane and the input tile is 20 pixels wide, so a group of 4 never straddles a tile row: the activation is
// stored with 4 ds_write_b128 instead of 16 ds_write_b32, and the depthwise stage reads the 12-pixel window of a tap row with
// 3 ds_read_b128 instead of 12 ds_read_b32 - half the LDS-array cycles (256 instead of 128 B/clk) and a quarter of the LDS
// instructions of the phase that bounds the deep 5x5 blocks.  Pitch 260 = 4 x 65 (odd): the 16 lanes of a ds_read_b128 group
// (16 different channels) land on 16 different 16-byte slots of the bank row, and the 8 lanes of a ds_write_b128 group on 8.
// 3x3 stride 1 (18-pixel-wide tile: a tap row starts at an even, not always a fourth, pixel) does the same with 8-byte accesses
// and pitch 258 = 2 x 129: 8 ds_write_b64 per slice, 3-5 ds_read_b64 per tap row, conflict-free (lane stride 2 banks).
#ifndef UDA_MBX_ET
#define UDA_MBX_ET 3      // bit 0: 5x5 stride 1 (16-byte accesses), bit 1: 3x3 stride 1 (8-byte accesses)
#endif
// floats per LDS access of the transposed slab: 4, 2, or 0 = slab not transposed
__host__ __device__ constexpr int mbx_et_w(int k, int s) {
  return s != 1 ? 0 : (k == 5 ? ((UDA_MBX_ET & 1) ? 4 : 0) : ((UDA_MBX_ET & 2) ? 2 : 0));
}
__host__ __device__ constexpr int mbx_et_pitch(int w) { return w == 2 ? 258 : 260; }
// activated accumulator of one 32-pixel slice -> transposed slab; ep = E + channel * pitch + slice * 32 + 4 * lane half
// (registers 4q .. 4q+3 of a lane = pixels 8q .. 8q+3 of its half of the slice)
template <int W>
__device__ __forceinline__ void et_store_slice(float* ep, const f32x16& acc) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float v0 = swish_core(acc[4 * q]), v1 = swish_core(acc[4 * q + 1]), v2 = swish_core(acc[4 * q + 2]), v3 = swish_core(acc[4 * q + 3]);
    if constexpr (W == 4) {
      *(float4*)(ep + 8 * q) = make_float4(v0, v1, v2, v3);
    } else {
      *(float2*)(ep + 8 * q) = make_float2(v0, v1);
      *(float2*)(ep + 8 * q + 2) = make_float2(v2, v3);
    }
  }
}
// access jj (W floats) of a tap row of the transposed slab
template <int W>
__device__ __forceinline__ void et_read(float* dst, const float* er, int jj) {
  if constexpr (W == 4) {
    const float4 v = *(const float4*)(er + 4 * jj);
    dst[4 * jj] = v.x; dst[4 * jj + 1] = v.y; dst[4 * jj + 2] = v.z; dst[4 * jj + 3] = v.w;
  } else {
    const float2 v = *(const float2*)(er + 2 * jj);
    dst[2 * jj] = v.x; dst[2 * jj + 1] = v.y;
  }
}

namespace {
struct MbxCfgB { int th, tw; };
__host__ __device__ constexpr MbxCfgB mbxb_cfg(int k, int s) {
  // input tile (with halo) = 8 slices of 32 pixels, two per wave: 3x3 s1 12x16 (14x18 = 252), 3x3 s2 7x8
  // (15x17 = 255: 224 of the 256 expanded pixels belong to the tile proper; 4x12 -> 9x25 = 225 had 192), 5x5 s2 4x10
  // (11x23 = 253), 5x5 s1 8x16 (12x20 = 240)
#ifndef UDA_MBXB_S2_TILE
#define UDA_MBXB_S2_TILE 78
#endif
  return s == 1 ? (k == 3 ? MbxCfgB{12, 16} : MbxCfgB{8, 16}) : (k == 3 ? (UDA_MBXB_S2_TILE == 78 ? MbxCfgB{7, 8} : MbxCfgB{4, 12}) : MbxCfgB{4, 10});
}
}  // namespace

// FUSE0: the block's 16-channel input does not exist in memory.  It is the previous block's 1x1 projection
// (block 0: 32 -> 16, BN, no activation) of a tensor D that is shared by the samples of an image, gated per sample
// (SE gate x deferred dropout scale).  The prologue computes it per 32-pixel slice on the matrix cores as
// X^T = (W0^T * gate * BN scale) D^T: the 32 x 32 accumulator then has the PIXEL on the lane and 8 of the 16 channels
// in registers 0..7 of each lane half - exactly an A-operand fragment of the expand MFMA with the k order
// (j, h) -> channel (j & 3) + 8 (j >> 2) + 4 h, which the host applies to the rows of the packed expand weights.
// No lane movement, no LDS, and the 16-channel tensor never goes to HBM.
// Occupancy is what this kernel lives on (PMC, round 2: VALU 53 % busy at two waves per SIMD, a quarter of the wave
// time spent in waits): the packed expand weights of a slab go through a 4-8 KB LDS image instead of 2 x 17 registers
// per wave, and E has no padding (both its writers and its readers put consecutive channels of ONE pixel on consecutive
// lanes: conflict-free at a row stride of 32 floats), so that the 3x3 variants fit 128 registers and 40 KB of LDS:
// four blocks = 16 waves per CU (5x5: three blocks; the 25 taps stay in registers).
// Wider inputs (3 or 4 k-steps: 48-64 registers of operand fragments per wave) do not reach that occupancy either way
// and keep the weight fragments of the current and the next slab in registers (measured: the LDS image costs them 6-9 %,
// its reads sit on the critical path right behind the slab barrier).
// 5x5 with 3 k-steps (block 4: 40 -> 240): the 25 depthwise taps are read from LDS at their use and the tap-row loop is not
// unrolled (as in mbxd_kernel), which brings the kernel under 168 registers = three blocks per CU instead of two.
#ifndef UDA_MBXB_WK_LDS
#define UDA_MBXB_WK_LDS 1
#endif
// Three pieces per operand (PARTS = 3, six cross terms, UDA_PW_TERMS=6): half as many operand fragments again per wave, so a
// block per CU less than the three-term variant.
#ifndef UDA_MBXB6_W
#define UDA_MBXB6_W 3      // (four would fit the registers of the 3x3 stride-1 variant, but 43 KB of LDS per block hold three anyway)
#endif
#ifndef UDA_MBXB_MINW
#define UDA_MBXB_MINW(K, S, KSF, PARTS) ((PARTS) == 3 ? (((KSF) <= 2 && (K) == 3) ? UDA_MBXB6_W : 2) : \
    ((KSF) <= 2 ? ((K) == 3 ? 4 : 3) : (((KSF) <= 3 && ((K) == 3 ? (S) == 2 : UDA_MBXB_WK_LDS)) ? 3 : 2)))
#endif
template <int K, int S, int KSF, bool FUSE0, int PARTS>   // KSF = 16-deep MFMA k-steps covering Cin + 1
__global__ __launch_bounds__(256, UDA_MBXB_MINW(K, S, KSF, PARTS)) void mbxb_kernel(MbxArgs a) {
  constexpr int NW = 4;
  constexpr int NPC = split_np(PARTS);        // pieces per operand (PARTS names the scheme: UDA_SPLIT_*)
  constexpr int TH = mbxb_cfg(K, S).th, TW = mbxb_cfg(K, S).tw;
  constexpr int IH = (TH - 1) * S + K, IW = (TW - 1) * S + K;
  constexpr int NP = IH * IW;
  constexpr int NPP = (NP + 31) / 32 * 32;
  constexpr int NMT = NPP / 32;
  constexpr int NG = NW * 2;                  // depthwise thread groups (32 channels each)
  constexpr int MTW = (NMT + NW - 1) / NW;    // pixel slices per wave
  constexpr int ES = 32;
  // stride 1: E is kept TRANSPOSED, [channel][pixel] with a channel pitch of CP floats (see mbx_et_w above)
  constexpr int ETW = mbx_et_w(K, S);
  constexpr bool ET = ETW != 0;
  constexpr int CP = mbx_et_pitch(ETW);
  // depthwise units: XW consecutive outputs of one row; NUNIT units over the NG thread groups
  constexpr int XW = (S == 1) ? 8 : (K == 3 ? (TW == 8 ? 8 : 6) : 5);
  constexpr int UPR = TW / XW;                // units per output row
  constexpr int NUNIT = TH * UPR;             // 24 (3x3 s1), 16 (5x5 s1), 8 (5x5 s2), 7 (3x3 s2: the eighth thread group idles)
  static_assert(TW % XW == 0, "units must tile the output tile");
  constexpr int NCOL = (XW - 1) * S + K;
  extern __shared__ __attribute__((aligned(16))) float mlds[];
  float* E = mlds;                            // [NPP][ES]  (ET: [32][CP])
  float* red = E + (ET ? (size_t)32 * CP : (size_t)NPP * ES);          // [NG][32]
  constexpr int NPAR = (K * K + 2) * 32;      // per slab: depthwise taps [K*K][32] | BN scale | BN shift (host-packed, a.wpar)
  float* par = red + NG * 32;                 // [2][NPAR]
  constexpr bool B_LDS = KSF <= 2;            // packed expand weights of a slab: LDS image (else registers, one slab ahead)
  constexpr int BSLAB = KSF * NPC * 64;     // uint4 per slab of packed expand weights
  uint4* Bs = (uint4*)(par + 2 * NPAR);       // [KSF][NPC][64 lanes]: rewritten between the two barriers of a slab

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  // block -> (tile, sample row).  Shared input (in_div = T samples per image): a 1-D grid in which the T blocks of a tile
  // sit at ids slot + 8 t of an 8 T wide group - workgroups are dealt round-robin over the 8 XCDs, so those T blocks
  // share one XCD's L2 and run at about the same time: the input tile comes from HBM / the Infinity Cache once, not T times
  // (PMC, round 1: 2.23x the algorithmic bytes).  Placement is a speed matter only; nothing depends on it.
  int tbx = blockIdx.x, tby = blockIdx.y, b = blockIdx.z;
  if (a.remap_T > 0) {
    const int T = a.remap_T, nt = a.tiles_x * a.tiles_y, ntp = (nt + 7) & ~7;
    const unsigned L = blockIdx.x;
    const unsigned per_img = (unsigned)ntp * (unsigned)T;
    const int img = (int)(L / per_img);
    const unsigned r = L - (unsigned)img * per_img;
    const int grp = (int)(r / (8u * (unsigned)T)), w = (int)(r % (8u * (unsigned)T));
    const int tl = grp * 8 + (w & 7);
    if (tl >= nt) return;                      // padding of the last group (whole block leaves before any barrier)
    b = img * T + (w >> 3);
    tby = tl / a.tiles_x;
    tbx = tl - tby * a.tiles_x;
  }
  const int b_in = b / a.in_div;
  const int oy0 = tby * TH, ox0 = tbx * TW;
  const int iy0 = oy0 * S - a.pad_t, ix0 = ox0 * S - a.pad_l;
  const int cin_mem = FUSE0 ? a.c0 : a.Cin;   // channels of the tensor that is actually read
  const float* xin = a.in + (size_t)b_in * a.H * a.W * cin_mem;

  float amax = 0.f;                           // fp16 pieces: largest operand magnitude this lane has split
  // FUSE0: A' = W0^T (row = projected channel li) x gate of this sample row, both k-steps of the 32 input channels
  bf16x8 w0p[2][NPC];
  float sh0v[8];
  if constexpr (FUSE0) {
    if (a.w0frag) {
      // the gated, split projection kernel of this sample row comes ready-made from w0gate_kernel: every one of the ~1100 tiles
      // of a row used to redo the same 16 multiplies + 2 splits (a tenth of this kernel's vector instructions)
      const uint4* fp = a.w0frag + (size_t)(b / a.g_div) * (2 * NPC * 64) + lane;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int p_ = 0; p_ < NPC; ++p_) w0p[ks][p_] = __builtin_bit_cast(bf16x8, fp[(ks * NPC + p_) * 64]);
    } else {
      const float* gp = a.gate + (size_t)(b / a.g_div) * a.c0;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int k = ks * 16 + 8 * lh;
        float4 w0 = *(const float4*)(a.w0t + li * 32 + k), w1 = *(const float4*)(a.w0t + li * 32 + k + 4);
        const float4 g0 = *(const float4*)(gp + k), g1 = *(const float4*)(gp + k + 4);
        w0.x *= g0.x; w0.y *= g0.y; w0.z *= g0.z; w0.w *= g0.w;
        w1.x *= g1.x; w1.y *= g1.y; w1.z *= g1.z; w1.w *= g1.w;
        split_parts<PARTS>(w0, w1, w0p[ks], amax);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) sh0v[j] = a.sh0f[(j & 3) + 8 * (j >> 2) + 4 * lh];
  }

  // ---- this wave's operand fragments: pixel = slice * 32 + li, channels 16 ks + 8 lh .. + 7
  bf16x8 ap[MTW][KSF][NPC];
#pragma unroll
  for (int t = 0; t < MTW; ++t) {
    const int mt = wave + NW * t;
    const int p = mt * 32 + li;
    const int iy = iy0 + p / IW, ix = ix0 + p % IW;
    const bool in = (mt < NMT) && (p < NP) && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    const float* px = xin + ((size_t)(in ? iy : 0) * a.W + (in ? ix : 0)) * cin_mem;
    if constexpr (FUSE0) {
      static_assert(!FUSE0 || KSF == 2, "the fused projection feeds a 16-channel expand");
      f32x16 xacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) xacc[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int k = ks * 16 + 8 * lh;
        float4 d0 = make_float4(0.f, 0.f, 0.f, 0.f), d1 = d0;
        if (in) {
          d0 = *(const float4*)(px + k);
          d1 = *(const float4*)(px + k + 4);
        }
        bf16x8 dp[NPC];
        split_parts<PARTS>(d0, d1, dp, amax);
        xacc = mfma_terms<PARTS>(w0p[ks], dp, xacc);
      }
      // registers 0..7 = channels (j & 3) + 8 (j >> 2) + 4 lh of pixel li: BN shift, zero outside the image
      float4 v0, v1;
      v0.x = in ? xacc[0] + sh0v[0] : 0.f; v0.y = in ? xacc[1] + sh0v[1] : 0.f;
      v0.z = in ? xacc[2] + sh0v[2] : 0.f; v0.w = in ? xacc[3] + sh0v[3] : 0.f;
      v1.x = in ? xacc[4] + sh0v[4] : 0.f; v1.y = in ? xacc[5] + sh0v[5] : 0.f;
      v1.z = in ? xacc[6] + sh0v[6] : 0.f; v1.w = in ? xacc[7] + sh0v[7] : 0.f;
      split_parts<PARTS>(v0, v1, ap[t][0], amax);
      float4 f0 = make_float4((lh == 0 && in) ? 1.f : 0.f, 0.f, 0.f, 0.f), f1 = make_float4(0.f, 0.f, 0.f, 0.f);
      split_parts<PARTS>(f0, f1, ap[t][1]);     // k-step 1: only the "inside the image" channel
      continue;
    }
#pragma unroll
    for (int ks = 0; ks < KSF; ++ks) {
      const int k = ks * 16 + 8 * lh;
      float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
      if (in && k < a.Cin) {       // Cin % 8 == 0: a group of 8 is either inside or past the channels
        v0 = *(const float4*)(px + k);
        v1 = *(const float4*)(px + k + 4);
      }
      if (k == a.Cin) v0.x = in ? 1.f : 0.f;          // the "inside the image" channel that carries the BN shift
      split_parts<PARTS>(v0, v1, ap[t][ks], amax);
    }
  }
  split_report<PARTS>(amax, a.oor);

  const int c = tid & 31, g = tid >> 5;       // depthwise stage: channel within the slab, thread group
  const size_t tile = (size_t)tby * ((a.Wo + TW - 1) / TW) + tbx;
  const int NCH = (a.Cmid + 31) >> 5;
  const uint4* Wp = (const uint4*)a.wsplit;

  // ---- loop-invariant addressing of the depthwise stage.  Unit u = g + NG * ui of this thread: XW outputs of output row
  // orow starting at column oxs.  eoff = float offset of the unit's E window (tap rows / columns are compile-time
  // displacements), ooff = element offset of the unit's first output relative to the block's output base, which is
  // UNIFORM (scalar registers): the stores use a scalar base + a 32-bit lane offset, no 64-bit vector arithmetic.
  constexpr int UPT = (NUNIT + NG - 1) / NG;
  constexpr bool UNIT_GUARD = (NUNIT % NG) != 0;      // the last round of units does not fill the thread groups
  int eoff[UPT];
  unsigned ooff[UPT];
#pragma unroll
  for (int ui = 0; ui < UPT; ++ui) {
    const int u = (UNIT_GUARD && g + NG * ui >= NUNIT) ? 0 : g + NG * ui;
    const int orow = u / UPR, oxs = (u % UPR) * XW;
    eoff[ui] = ET ? c * CP + (orow * S) * IW + oxs * S : ((orow * S) * IW + oxs * S) * ES + c;
    ooff[ui] = (unsigned)((orow * a.Wo + oxs) * a.Cmid + c) * 4u;      // bytes
  }
  float* const obase = a.out + (((size_t)b * a.Ho + oy0) * a.Wo + ox0) * a.Cmid;
  const unsigned cm = (unsigned)a.Cmid;
  // a tile that lies inside the output takes the unguarded path (block-uniform decision; the channel guard of a last,
  // partial slab is one exec mask around the whole stage)
  const bool full = (oy0 + TH <= a.Ho) && (ox0 + TW <= a.Wo);

  // Per-slab operands - packed expand weights (LDS image Bs), depthwise taps + BN scale / shift (LDS, one contiguous
  // host-packed block per slab), the two dropout scales (registers) - are requested while the PREVIOUS slab's expand
  // phase runs, i.e. before that slab's output stores are issued (vmcnt retires loads and stores in order), by plain
  // branch-free loads, so that no wait is placed right behind them; they are written to LDS between the two barriers.
  constexpr int P_PER = (NPAR + 255) / 256;
  constexpr int B_PER = (BSLAB + 255) / 256;
  auto b_src = [&](int ch, int f) -> const uint4* {       // element f of slab ch in the [ks][part][lane] image
    const int ks = f / (NPC * 64), rest = f - ks * (NPC * 64);
    return Wp + (((size_t)ks * NCH + ch) * NPC + (rest >> 6)) * 64 + (rest & 63);
  };
  constexpr int KR = B_LDS ? 1 : KSF;         // register copies (B_LDS: unused)
  uint4 rb[KR][NPC], nbr[KR][NPC];
  auto load_regs = [&](int ch, uint4 (*dst)[NPC]) {
#pragma unroll
    for (int ks = 0; ks < KR; ++ks)
#pragma unroll
      for (int p_ = 0; p_ < NPC; ++p_) dst[ks][p_] = Wp[(((size_t)ks * NCH + ch) * NPC + p_) * 64 + lane];
  };
  if constexpr (B_LDS) {
    for (int f = tid; f < BSLAB; f += 256) Bs[f] = *b_src(0, f);
  } else {
    load_regs(0, rb);
  }
  for (int f = tid; f < NPAR; f += 256) par[f] = a.wpar[f];
  auto mask_at = [&](const float* m, int ch) -> float {   // keep-scale x (-ln 2) of channel 32 ch + c (c == li) of this sample row
    const int col_ = ch * 32 + c;
    const float* mp = m ? m + (size_t)b * a.Cmid + (col_ < a.Cmid ? col_ : 0) : a.wpar;
    const float v = *mp;
    return (m ? v : 1.f) * UDA_NEG_LN2;
  };
  float mk0 = mask_at(a.mask0, 0), mk0n = mk0;            // expand side: applied after the depthwise (see swish_core)
  float mk1 = mask_at(a.mask1, 0), mk1n = mk1;
  __syncthreads();

  for (int ch = 0; ch < NCH; ++ch) {
    const int col = ch * 32 + c;
    const bool dcol = col < a.Cmid;
    const float* pcur = par + (ch & 1) * NPAR;
    // ---- the NEXT slab's operands are requested first: unconditional, branch-free loads with a whole phase to arrive
    const bool more = ch + 1 < NCH;
    const int chn = more ? ch + 1 : ch;
    uint4 nb[B_PER];
    if constexpr (B_LDS) {
#pragma unroll
      for (int i = 0; i < B_PER; ++i) {
        const int f = tid + 256 * i;
        nb[i] = *b_src(chn, f < BSLAB ? f : 0);
      }
    } else {
      load_regs(chn, nbr);
    }
    float np_[P_PER];
    const float* wnext = a.wpar + (size_t)chn * NPAR;
#pragma unroll
    for (int i = 0; i < P_PER; ++i) {
      const int f = tid + 256 * i;
      np_[i] = wnext[f < NPAR ? f : 0];
    }
    mk0n = mask_at(a.mask0, chn);
    mk1n = mask_at(a.mask1, chn);
    // this slab's weight fragments: one LDS image for the block, read once per wave and used for both of its slices
    bf16x8 bp[KSF][NPC];
#pragma unroll
    for (int ks = 0; ks < KSF; ++ks)
#pragma unroll
      for (int p_ = 0; p_ < NPC; ++p_) {
        if constexpr (B_LDS) bp[ks][p_] = __builtin_bit_cast(bf16x8, Bs[(ks * NPC + p_) * 64 + lane]);
        else bp[ks][p_] = __builtin_bit_cast(bf16x8, rb[ks < KR ? ks : 0][p_]);
      }
    // ---- expand: E[p][j] = swish(sum_k X[p][k] We'[k][32 ch + j]) / (-ln 2)   (dropout scale: after the depthwise)
#pragma unroll
    for (int t = 0; t < MTW; ++t) {
      const int mt = wave + NW * t;
      if (mt < NMT) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSF; ++ks) acc = mfma_terms<PARTS>(ap[t][ks], bp[ks], acc);
        if constexpr (ET) {
          et_store_slice<ETW>(E + li * CP + mt * 32 + 4 * lh, acc);
        } else {
          float* ep = E + (size_t)(mt * 32 + 4 * lh) * ES + li;
#pragma unroll
          for (int r = 0; r < 16; ++r) ep[((r & 3) + 8 * (r >> 2)) * ES] = swish_core(acc[r]);
        }
      }
    }
    __syncthreads();
    // the next slab's operands have had the whole expand phase to arrive; they are consumed HERE, ahead of this
    // slab's output stores, so that no later wait has to retire those stores (vmcnt retires in order).  (The single
    // weight image is free: every wave read its fragments before the barrier above.)
    {
      if constexpr (B_LDS) {
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
          const int f = tid + 256 * i;
          if (more && f < BSLAB) Bs[f] = nb[i];
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < KR; ++ks)
#pragma unroll
          for (int p_ = 0; p_ < NPC; ++p_)
            asm volatile("" : "+v"(nbr[ks][p_].x), "+v"(nbr[ks][p_].y), "+v"(nbr[ks][p_].z), "+v"(nbr[ks][p_].w));
      }
      float* pnext = par + ((ch + 1) & 1) * NPAR;
#pragma unroll
      for (int i = 0; i < P_PER; ++i) {
        const int f = tid + 256 * i;
        if (more && f < NPAR) pnext[f] = np_[i];
      }
      asm volatile("" : "+v"(mk0n), "+v"(mk1n));
    }
    // ---- depthwise on E for channel 32 ch + c
    constexpr bool WK_LDS = UDA_MBXB_WK_LDS && K == 5 && KSF == 3;
    float wk[WK_LDS ? 1 : K * K];
    if constexpr (!WK_LDS) {
#pragma unroll
      for (int t = 0; t < K * K; ++t) wk[t] = pcur[t * 32 + c];
    }
    // BN scale x (-ln 2) x expand-side dropout scale of this channel (see swish_core)
    const float sc1 = pcur[K * K * 32 + c] * mk0, sh1 = pcur[(K * K + 1) * 32 + c];
    float ssum = 0.f;
    float* const ob = obase + ch * 32;            // uniform
    auto dw_units = [&](auto guard) {
      constexpr bool GUARD = decltype(guard)::value;
      constexpr int KYU = (UDA_MBXB_WK_LDS && K == 5 && KSF == 3) ? 1 : K;      // (an unrolled tap-row loop hoists the tap reads back into registers)
#pragma unroll
      for (int ui = 0; ui < UPT; ++ui) {
        if constexpr (UNIT_GUARD) {
          if (g + NG * ui >= NUNIT) continue;
        }
        if constexpr (GUARD) {
          if (oy0 + (g + NG * ui) / UPR >= a.Ho) continue;
        }
        float acc[XW];
#pragma unroll
        for (int o = 0; o < XW; ++o) acc[o] = 0.f;
        const float* eu = E + eoff[ui];
#pragma unroll KYU
        for (int ky = 0; ky < K; ++ky) {
          float rowv[NCOL];
          if constexpr (ET) {
            static_assert(!ET || NCOL % ETW == 0, "whole 16- / 8-byte reads");
#pragma unroll
            for (int jj = 0; jj < NCOL / (ET ? ETW : 1); ++jj) et_read<ETW>(rowv, eu + ky * IW, jj);
          } else {
#pragma unroll
            for (int j = 0; j < NCOL; ++j) rowv[j] = eu[(ky * IW + j) * ES];
          }
#pragma unroll
          for (int kx = 0; kx < K; ++kx) {
            const float w = WK_LDS ? pcur[(ky * K + kx) * 32 + c] : wk[WK_LDS ? 0 : ky * K + kx];
#pragma unroll
            for (int o = 0; o < XW; ++o) acc[o] = fmaf(rowv[o * S + kx], w, acc[o]);
          }
        }
#pragma unroll
        for (int o = 0; o < XW; ++o) {
          bool ok = true;
          if constexpr (GUARD) ok = ox0 + (int)((g + NG * ui) % UPR) * XW + o < a.Wo;
          if (ok) {
            const float v = swish_folded(fmaf(acc[o], sc1, sh1), mk1);
            store_uniform_base(ob + (size_t)o * cm, ooff[ui], v);
            ssum += v;
          }
        }
        // units are scheduled one after the other: hoisting the E reads of all units to the front costs 20-30 registers
        // (and with them a block per CU)
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (dcol) {                  // (lane-varying only in a last, partial slab: Cmid = 144, 240)
      if (full) dw_units(std::false_type());
      else dw_units(std::true_type());
    }
    if (a.se_partial) red[g * 32 + c] = ssum;
    __syncthreads();   // E may be rewritten; red[] of this slab and the next slab's operands (written after the first barrier) are complete
    if (a.se_partial && g == 0 && dcol) {
      float t = red[c];
#pragma unroll
      for (int gg = 1; gg < NG; ++gg) t += red[gg * 32 + c];
      a.se_partial[((size_t)b * a.n_tiles + tile) * a.Cmid + col] = t;
    }
    mk0 = mk0n;
    mk1 = mk1n;
    if constexpr (!B_LDS) {
#pragma unroll
      for (int ks = 0; ks < KR; ++ks)
#pragma unroll
        for (int p_ = 0; p_ < NPC; ++p_) rb[ks][p_] = nbr[ks][p_];
    }
  }
}

// FUSE0 operand prep: A' = W0^T x gate of every gate row, split into PARTS bf16 pieces and laid out as the A fragments the
// fused kernel's prologue wants ([row][k-step][piece][lane] x 16 B) - the same arithmetic, once per sample row instead of once
// per tile.  One wave per gate row.
template <int PARTS>
__global__ __launch_bounds__(64) void w0gate_kernel(const float* gate, const float* w0t, int c0, uint4* out, unsigned* oor) {
  constexpr int NPC = split_np(PARTS);
  const int row = blockIdx.x, lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
  const float* gp = gate + (size_t)row * c0;
  float amax = 0.f;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int k = ks * 16 + 8 * lh;
    float4 w0 = *(const float4*)(w0t + li * 32 + k), w1 = *(const float4*)(w0t + li * 32 + k + 4);
    const float4 g0 = *(const float4*)(gp + k), g1 = *(const float4*)(gp + k + 4);
    w0.x *= g0.x; w0.y *= g0.y; w0.z *= g0.z; w0.w *= g0.w;
    w1.x *= g1.x; w1.y *= g1.y; w1.z *= g1.z; w1.w *= g1.w;
    bf16x8 pc[NPC];
    split_parts<PARTS>(w0, w1, pc, amax);
#pragma unroll
    for (int p_ = 0; p_ < NPC; ++p_) out[((size_t)row * 2 * NPC + ks * NPC + p_) * 64 + lane] = __builtin_bit_cast(uint4, pc[p_]);
  }
  split_report<PARTS>(amax, oor);
}

// gate rows x (2 k-steps x parts x 64 lanes) uint4
size_t mbxb_w0frag_elems(int gate_rows, int scheme) { return (size_t)gate_rows * 2 * uda_split_pieces(scheme) * 64; }
void launch_w0gate(const float* gate, const float* w0t, int c0, int gate_rows, int scheme, uint4* out, unsigned* oor, hipStream_t s) {
  if (scheme == UDA_SPLIT_BF16X3) hipLaunchKernelGGL(w0gate_kernel<3>, dim3(gate_rows), dim3(64), 0, s, gate, w0t, c0, out, oor);
  else if (scheme == UDA_SPLIT_F16X2) hipLaunchKernelGGL(w0gate_kernel<4>, dim3(gate_rows), dim3(64), 0, s, gate, w0t, c0, out, oor);
  else hipLaunchKernelGGL(w0gate_kernel<2>, dim3(gate_rows), dim3(64), 0, s, gate, w0t, c0, out, oor);
}

int mbxb_tiles(int Ho, int Wo, int k, int stride) {
  const MbxCfgB c = mbxb_cfg(k, stride);
  return ((Ho + c.th - 1) / c.th) * ((Wo + c.tw - 1) / c.tw);
}

bool mbxb_supported(int Cin, int Cmid, int k, int stride) {
  return Cin % 8 == 0 && Cin >= 16 && Cin <= 48 && Cmid % 4 == 0 && (k == 3 || k == 5) && (stride == 1 || stride == 2);
}

template <int K, int S, int KSF, int PARTS>
static void launch_mbxb_t(const MbxArgs& a, int rows, hipStream_t s) {
  constexpr int TH = mbxb_cfg(K, S).th, TW = mbxb_cfg(K, S).tw;
  constexpr int IH = (TH - 1) * S + K, IW = (TW - 1) * S + K;
  constexpr int NPP = (IH * IW + 31) / 32 * 32;
  const size_t lds = ((mbx_et_w(K, S) ? (size_t)32 * mbx_et_pitch(mbx_et_w(K, S)) : (size_t)NPP * 32) + 8 * 32 + 2 * (K * K + 2) * 32) * sizeof(float) +
                     (size_t)KSF * split_np(PARTS) * 64 * sizeof(uint4);
  dim3 grid((a.Wo + TW - 1) / TW, (a.Ho + TH - 1) / TH, rows);
  MbxArgs b = a;
  static int remap = -1;
  if (remap < 0) { const char* e = getenv("UDA_MBX_REMAP"); remap = e ? atoi(e) : 1; }
  b.remap_T = 0;
  if (remap && a.in_div > 1 && rows % a.in_div == 0) {      // input shared by the samples of an image: XCD-grouped 1-D grid
    const long long nt = (long long)grid.x * grid.y, ntp = (nt + 7) / 8 * 8;
    const long long total = ntp * rows;
    if (total < (1ll << 31)) {
      b.remap_T = a.in_div; b.tiles_x = (int)grid.x; b.tiles_y = (int)grid.y;
      grid = dim3((unsigned)total, 1, 1);
    }
  }
  if constexpr (KSF == 2) {
    if (a.gate) { hipLaunchKernelGGL((mbxb_kernel<K, S, KSF, true, PARTS>), grid, dim3(256), lds, s, b); return; }
  }
  hipLaunchKernelGGL((mbxb_kernel<K, S, KSF, false, PARTS>), grid, dim3(256), lds, s, b);
}

template <int K, int S, int PARTS>
static void launch_mbxb_ks(const MbxArgs& a, int rows, hipStream_t s) {
  switch ((a.Cin + 1 + 15) / 16) {
    case 2: launch_mbxb_t<K, S, 2, PARTS>(a, rows, s); break;
    case 3: launch_mbxb_t<K, S, 3, PARTS>(a, rows, s); break;
    default: launch_mbxb_t<K, S, 4, PARTS>(a, rows, s); break;
  }
}

template <int PARTS>
static void launch_mbxb_p(const MbxArgs& a, int rows, int k, int stride, hipStream_t s) {
  if (k == 3 && stride == 1) launch_mbxb_ks<3, 1, PARTS>(a, rows, s);
  else if (k == 3 && stride == 2) launch_mbxb_ks<3, 2, PARTS>(a, rows, s);
  else if (k == 5 && stride == 1) launch_mbxb_ks<5, 1, PARTS>(a, rows, s);
  else launch_mbxb_ks<5, 2, PARTS>(a, rows, s);
}

// a.wparts: split scheme of the packed expand weights (UDA_SPLIT_*)
void launch_mbxb(const MbxArgs& a, int rows, int k, int stride, hipStream_t s) {
  if (a.wparts == UDA_SPLIT_BF16X3) launch_mbxb_p<3>(a, rows, k, stride, s);
  else if (a.wparts == UDA_SPLIT_F16X2) launch_mbxb_p<4>(a, rows, k, stride, s);
  else launch_mbxb_p<2>(a, rows, k, stride, s);
}

// ---------------------------------------------------------------- fused MBConv front half, deep blocks
// The same fusion for the deep stride-1 blocks (Cin 56..208, expanded width up to 6 * Cin): 8 waves per block,
// wave w keeps the operand fragments of the w-th 32-pixel slice of the input tile (12 x 16 + halo for 3x3,
// 8 x 16 + halo for 5x5 = 8 slices) in registers; the packed expand weights of a 32-channel slab are shared
// through a double-buffered LDS image (requested during the depthwise phase of the previous slab).
// Per slab: expand (3 MFMAs per k-step) -> E slab in LDS -> barrier -> depthwise + output + SE sums -> barrier.
// (the second __launch_bounds__ argument is waves per SIMD: a 512-thread block puts 2 waves on every SIMD, so "4" = two
// blocks per CU = at most 128 VGPRs.  The 5x5 variants with up to 8 k-steps fit that once the 25 depthwise taps are read
// from LDS at their use instead of living in registers; with two blocks per CU the expand phase of one block (MFMA +
// transcendentals) overlaps the depthwise phase of the other (FMA + LDS) - with one block both phases run in lockstep.)
struct MbxCfgD { int th, tw, xw; };
// output tile and outputs per depthwise unit of the deep kernels: stride 1: 12 x 16 (3x3, 14 x 18 = 252 input pixels) / 8 x 16
// (5x5, 12 x 20 = 240); stride 2 (the first block of a stage, e.g. EfficientNet-B0 block 11: 5x5, 112 -> 672): 4 x 10
// (11 x 23 = 253) / 7 x 8 (3x3, 15 x 17 = 255) - the tiles of mbxb_kernel.  WIDE (stride 1): 20 columns - 8 x 20 (3x3, 10 x 22 =
// 220) / 6 x 20 (5x5, 10 x 24 = 240) for maps whose width 16 does not divide: the 24 x 40 maps of blocks 12-15 at 1280 x 768 take
// 8 whole tiles per image instead of 9 of which 3 are half empty (and guarded output by output).
__host__ __device__ constexpr MbxCfgD mbxd_cfg(int k, int s, bool wide = false) {
  return s == 1 ? (wide ? (k == 3 ? MbxCfgD{8, 20, 10} : MbxCfgD{6, 20, 4}) : (k == 3 ? MbxCfgD{12, 16, 4} : MbxCfgD{8, 16, 8}))
                : (k == 3 ? MbxCfgD{7, 8, 8} : MbxCfgD{4, 10, 5});
}
// the planner (plan.mbx_tile) and the launchers agree on this rule: the wide tile when it covers the map with fewer output slots
__host__ __device__ constexpr long long mbxd_slots(int Ho, int Wo, int k, bool wide) {
  return (long long)((Ho + mbxd_cfg(k, 1, wide).th - 1) / mbxd_cfg(k, 1, wide).th) * mbxd_cfg(k, 1, wide).th *
         ((Wo + mbxd_cfg(k, 1, wide).tw - 1) / mbxd_cfg(k, 1, wide).tw) * mbxd_cfg(k, 1, wide).tw;
}
bool mbxd_wide(int Ho, int Wo, int k, int stride) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("UDA_MBXD_WIDE"); on = e ? atoi(e) : 1; }
  return on && stride == 1 && mbxd_slots(Ho, Wo, k, true) < mbxd_slots(Ho, Wo, k, false);
}

template <int K, int KSF, int PARTS, int S, bool WIDE>
__global__ __launch_bounds__(512, (KSF <= 8 && PARTS != 3) ? 4 : 2) void mbxd_kernel(MbxArgs a) {
  constexpr bool WK_LDS = (K == 5 && (KSF <= 8 || PARTS == 3));     // the 25 taps from LDS at their use, not 25 registers

  constexpr int NW = 8;
  constexpr int NPC = split_np(PARTS);        // pieces per operand (PARTS names the scheme: UDA_SPLIT_*)
  static_assert(!WIDE || S == 1, "wide tiles are a stride-1 variant");
  constexpr int TH = mbxd_cfg(K, S, WIDE).th, TW = mbxd_cfg(K, S, WIDE).tw;
  constexpr int IH = (TH - 1) * S + K, IW = (TW - 1) * S + K;
  constexpr int NP = IH * IW;
  static_assert(NP <= 256, "input tile must fit 8 slices of 32 pixels");
  constexpr int NPP = 256;
  constexpr int NG = NW * 2;                  // depthwise thread groups (32 channels each)
  constexpr int ES = 33;
  constexpr int ETW = mbx_et_w(K, S);         // E transposed, [channel][pixel] (stride 1 only, see mbx_et_w); 32 * CP <= NPP * ES floats
  constexpr bool ET = ETW != 0;
  constexpr int CP = mbx_et_pitch(ETW);
  static_assert(!ET || (IW % ETW == 0 && mbxd_cfg(K, S, WIDE).xw % ETW == 0), "aligned wide accesses of the transposed slab");
  constexpr int XW = mbxd_cfg(K, S, WIDE).xw; // outputs per unit along x
  constexpr int UPR = TW / XW;                // units per output row
  constexpr int NUNIT = TH * UPR;             // 48 (3x3) / 16 (5x5) units over 16 groups; stride 2: 7 / 8 (half of the groups idle)
  static_assert(TW % XW == 0, "units must tile the output tile");
  constexpr int NCOL = (XW - 1) * S + K;
  constexpr int BSLAB = KSF * NPC * 64;     // uint4 per slab of packed expand weights
  extern __shared__ __attribute__((aligned(16))) float dlds[];
  float* E = dlds;                            // [NPP][ES]
  float* red = E + (size_t)NPP * ES;          // [NG][32]
  constexpr int NPAR = (K * K + 2) * 32;      // per slab: depthwise taps [K*K][32] | BN scale | BN shift (host-packed, a.wpar)
  float* par = red + NG * 32;                 // [2][NPAR]
  uint4* Bs = (uint4*)(par + 2 * NPAR);       // [KSF][2 parts][64 lanes]: rewritten between the two barriers of a slab
  float* mks = (float*)(Bs + BSLAB);          // [2][32 * NCH]: dropout keep-scales * -ln2 of this sample row (expand, depthwise)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z / a.ch_groups, b_in = b / a.in_div;
  const int oy0 = blockIdx.y * TH, ox0 = blockIdx.x * TW;
  const int iy0 = oy0 * S - a.pad_t, ix0 = ox0 * S - a.pad_l;
  const float* xin = a.in + (size_t)b_in * a.H * a.W * a.Cin;
  const int NCH = (a.Cmid + 31) >> 5;
  // this block's slabs [chb, che): all of them, or one of ch_groups contiguous ranges (small grids, see MbxArgs)
  const int ch_per = (NCH + a.ch_groups - 1) / a.ch_groups;
  const int chb = (int)(blockIdx.z % a.ch_groups) * ch_per;
  const int che = chb + ch_per < NCH ? chb + ch_per : NCH;
  if (chb >= che) return;                     // (whole block, before any barrier)
  const uint4* Wp = (const uint4*)a.wsplit;
  // Every per-slab operand (packed expand weights, host-packed depthwise taps + BN scale / shift, dropout scales) is
  // requested while the PREVIOUS slab's expand phase ends, i.e. before that slab's output stores are issued: vmcnt
  // retires loads and stores in order, so a load issued after the stores would make its consumer wait for the stores'
  // HBM round trip.  The loads are plain and branch-free so that no wait is placed right behind them.
  // ---- slab 0 operands -> LDS buffers 0
  for (int f = tid; f < BSLAB; f += 512) {
    const int ks = f / (NPC * 64), rest = f - ks * (NPC * 64);     // [ks][part][lane]
    Bs[f] = Wp[(((size_t)ks * NCH + chb) * NPC + (rest >> 6)) * 64 + (rest & 63)];
  }
  for (int f = tid; f < NPAR; f += 512) par[f] = a.wpar[(size_t)chb * NPAR + f];
  for (int f = tid; f < 2 * 32 * NCH; f += 512) {
    const int which = f / (32 * NCH), col = f - which * 32 * NCH;
    const float* m = which ? a.mask1 : a.mask0;
    mks[f] = ((m && col < a.Cmid) ? m[(size_t)b * a.Cmid + col] : 1.f) * UDA_NEG_LN2;
  }

  // ---- this wave's operand fragments: pixel = wave * 32 + li, channels 16 ks + 8 lh .. + 7
  bf16x8 ap[KSF][NPC];
  float amax = 0.f;                           // fp16 pieces: largest operand magnitude this lane has split
  {
    const int p = wave * 32 + li;
    const int iy = iy0 + p / IW, ix = ix0 + p % IW;
    const bool in = (p < NP) && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    const float* px = xin + ((size_t)(in ? iy : 0) * a.W + (in ? ix : 0)) * a.Cin;
#pragma unroll
    for (int ks = 0; ks < KSF; ++ks) {
      const int k = ks * 16 + 8 * lh;
      float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
      if (in && k < a.Cin) {
        v0 = *(const float4*)(px + k);
        v1 = *(const float4*)(px + k + 4);
      }
      if (k == a.Cin) v0.x = in ? 1.f : 0.f;          // the "inside the image" channel that carries the BN shift
      split_parts<PARTS>(v0, v1, ap[ks], amax);
    }
  }
  split_report<PARTS>(amax, a.oor);
  __syncthreads();

  const int c = tid & 31, g = tid >> 5;       // depthwise stage: channel within the slab, thread group
  const size_t tile = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  constexpr int B_PER = (BSLAB + 511) / 512;  // uint4 of the next slab per thread
  // loop-invariant addressing of the depthwise stage (see mbxb_kernel): E window offsets, output offsets relative to the
  // block's uniform output base, and the block-uniform "whole tile" decision for the unguarded path
  constexpr int UPT = (NUNIT + NG - 1) / NG;
  constexpr bool UNIT_GUARD = (NUNIT % NG) != 0;      // stride 2: fewer units than thread groups
  int eoff[UPT];
  unsigned ooff[UPT];
#pragma unroll
  for (int ui = 0; ui < UPT; ++ui) {
    const int u = (UNIT_GUARD && g + NG * ui >= NUNIT) ? 0 : g + NG * ui;
    const int orow = u / UPR, oxs = (u % UPR) * XW;
    eoff[ui] = ET ? c * CP + orow * S * IW + oxs * S : (orow * S * IW + oxs * S) * ES + c;
    ooff[ui] = (unsigned)((orow * a.Wo + oxs) * a.Cmid + c) * 4u;      // bytes
  }
  float* const obase = a.out + (((size_t)b * a.Ho + oy0) * a.Wo + ox0) * a.Cmid;
  const unsigned cm = (unsigned)a.Cmid;
  const bool full = (oy0 + TH <= a.Ho) && (ox0 + TW <= a.Wo);

  for (int ch = chb; ch < che; ++ch) {
    const int col = ch * 32 + c;
    const bool dcol = col < a.Cmid;
    const uint4* bcur = Bs;
    const float mk0c = mks[ch * 32 + c], mk1 = mks[32 * NCH + ch * 32 + c];   // expand-side scale of channel c: applied after the depthwise
    const float* pcur = par + ((ch - chb) & 1) * NPAR;
    // ---- the NEXT slab's operands are requested first: they have the whole expand phase to arrive
    const bool more = ch + 1 < che;
    uint4 nb[B_PER];
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {      // (guarded loads measured 6 % faster than clamped unconditional ones here)
      const int f = tid + 512 * i;
      const int ks = f / (NPC * 64), rest = f - ks * (NPC * 64);
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (more && f < BSLAB) v = Wp[(((size_t)ks * NCH + (ch + 1)) * NPC + (rest >> 6)) * 64 + (rest & 63)];
      nb[i] = v;
    }
    constexpr int P_PER = (NPAR + 511) / 512;
    float np_[P_PER];
    const float* wnext = a.wpar + (size_t)(more ? ch + 1 : ch) * NPAR;
#pragma unroll
    for (int i = 0; i < P_PER; ++i) {
      const int f = tid + 512 * i;
      np_[i] = wnext[f < NPAR ? f : 0];
    }
    // ---- expand: E[p][j] = swish(sum_k X[p][k] We'[k][32 ch + j]) / (-ln 2) for this wave's 32 pixels
    {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KSF; ++ks) {
        bf16x8 bp[NPC];
#pragma unroll
        for (int p_ = 0; p_ < NPC; ++p_) bp[p_] = __builtin_bit_cast(bf16x8, bcur[(ks * NPC + p_) * 64 + lane]);
        acc = mfma_terms<PARTS>(ap[ks], bp, acc);
      }
      if constexpr (ET) {
        et_store_slice<ETW>(E + li * CP + wave * 32 + 4 * lh, acc);
      } else {
        float* ep = E + (size_t)(wave * 32 + 4 * lh) * ES + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) ep[((r & 3) + 8 * (r >> 2)) * ES] = swish_core(acc[r]);
      }
    }
    __syncthreads();
    // the next slab's operands are written to their LDS buffers HERE, ahead of this slab's output stores, so that no
    // later wait has to retire those stores (vmcnt retires in order)
    {
      // (the single weight buffer is free: every wave finished its expand phase before the barrier above)
#pragma unroll
      for (int i = 0; i < B_PER; ++i) {
        const int f = tid + 512 * i;
        if (more && f < BSLAB) Bs[f] = nb[i];
      }
      float* pnext = par + ((ch + 1 - chb) & 1) * NPAR;
#pragma unroll
      for (int i = 0; i < P_PER; ++i) {
        const int f = tid + 512 * i;
        if (more && f < NPAR) pnext[f] = np_[i];
      }
    }
    // ---- depthwise on E for channel 32 ch + c
    float wk[WK_LDS ? 1 : K * K];
    if constexpr (!WK_LDS) {
#pragma unroll
      for (int t = 0; t < K * K; ++t) wk[t] = pcur[t * 32 + c];
    }
    const float sc1 = pcur[K * K * 32 + c] * mk0c, sh1 = pcur[(K * K + 1) * 32 + c];
    float ssum = 0.f;
    float* const ob = obase + ch * 32;            // uniform
    auto dw_units = [&](auto guard) {
      constexpr bool GUARD = decltype(guard)::value;
#pragma unroll
      for (int ui = 0; ui < UPT; ++ui) {
        if constexpr (UNIT_GUARD) {
          if (g + NG * ui >= NUNIT) continue;
        }
        if constexpr (GUARD) {
          if (oy0 + (g + NG * ui) / UPR >= a.Ho) continue;
        }
        float acc[XW];
#pragma unroll
        for (int o = 0; o < XW; ++o) acc[o] = 0.f;
        const float* eu = E + eoff[ui];
        constexpr int KYU = WK_LDS ? 1 : K;     // (a fully unrolled tap-row loop hoists all 25 LDS tap reads back into registers)
#pragma unroll KYU
        for (int ky = 0; ky < K; ++ky) {
          float rowv[NCOL];
          if constexpr (ET) {
            static_assert(!ET || NCOL % ETW == 0, "whole 16- / 8-byte reads");
#pragma unroll
            for (int jj = 0; jj < NCOL / (ET ? ETW : 1); ++jj) et_read<ETW>(rowv, eu + ky * IW, jj);
          } else {
            const float* er = eu + ky * IW * ES;
#pragma unroll
            for (int j = 0; j < NCOL; ++j) rowv[j] = er[j * ES];
          }
#pragma unroll
          for (int kx = 0; kx < K; ++kx) {
            const float w = WK_LDS ? pcur[(ky * K + kx) * 32 + c] : wk[WK_LDS ? 0 : ky * K + kx];
#pragma unroll
            for (int o = 0; o < XW; ++o) acc[o] = fmaf(rowv[o * S + kx], w, acc[o]);
          }
        }
#pragma unroll
        for (int o = 0; o < XW; ++o) {
          bool ok = true;
          if constexpr (GUARD) ok = ox0 + (int)((g + NG * ui) % UPR) * XW + o < a.Wo;
          if (ok) {
            const float v = swish_folded(fmaf(acc[o], sc1, sh1), mk1);
            store_uniform_base(ob + (size_t)o * cm, ooff[ui], v);
            ssum += v;
          }
        }
        // units are scheduled one after the other: hoisting the E reads of all units to the front costs 20-30 registers
        // (and with them a block per CU)
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (dcol) {                  // (lane-varying only in a last, partial slab: Cmid = 144, 240)
      if (full) dw_units(std::false_type());
      else dw_units(std::true_type());
    }
    if (a.se_partial) red[g * 32 + c] = ssum;
    __syncthreads();   // E may be rewritten; red[] and the next slab's operands (written after the first barrier) are complete
    if (a.se_partial && g == 0 && dcol) {
      float t = red[c];
#pragma unroll
      for (int gg = 1; gg < NG; ++gg) t += red[gg * 32 + c];
      a.se_partial[((size_t)b * a.n_tiles + tile) * a.Cmid + col] = t;
    }
  }
}

// ---------------------------------------------------------------- fused MBConv front half, deep blocks, pipelined
// The variants with 13 / 14 k-steps (Cin 192 / 208) hold 104 / 112 VGPRs of operand fragments per wave, so only one
// block fits a CU and in mbxd_kernel its two phases run in lockstep: the matrix pipe idles during the depthwise phase
// and the VALU during the 39 dependent MFMAs of the expand.  Here every wave overlaps them itself: iteration `ch`
// issues the MFMAs of slab ch + 1 in between the LDS reads and FMAs of the depthwise of slab ch (one straight-line
// region; the matrix core runs asynchronously to the VALU), then activates that accumulator into the other E buffer.
// E, the packed weights and the SE sums are double-buffered, the per-slab depthwise block triple-buffered, and ONE
// barrier per slab remains.
#ifdef UDA_MBXP_STAMPS
// diagnostic build (-DUDA_MBXP_STAMPS, A/B libraries only): shader-clock cycles every wave spends in the phases of a slab
// iteration, summed over all waves of all launches and printed by the launcher every few launches.  [0] barrier wait,
// [1] operand requests + taps + first window row, [2] interleaved MFMA / depthwise region, [3] output epilogue + stores,
// [4] activation of the next slab -> E, [5] operand writes to LDS, [6] wave-iterations.  Round 4 (blocks 12-14, 5x5, 13 k-steps,
// 20-column tiles): 6970 cycles per wave and slab = barrier 1046 + requests 1197 + MFMA region 2628 (two waves x 39 MFMAs x 32
// cycles = 2496 on the SIMD's matrix pipe) + epilogue 831 + activation 552 + operand writes 723 (the wait for the slab-ahead
// weight loads, which retire in order behind the previous slab's output stores: moving the writes ahead of the stores moved
// the wait into the MFMA region, 6970 either way; without any output stores 6402).  DESIGN.md 4.6.
__device__ unsigned long long g_mbxp_stamps[8];
static void mbxp_stamp_dump() {       // (the runtime is gone by the time static destructors run)
  unsigned long long h[8] = {};
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_mbxp_stamps), sizeof(h)) != hipSuccess || !h[6]) return;
  const char* nm[6] = {"barrier wait", "requests + first row", "MFMA / depthwise region", "epilogue + stores", "activation -> E", "operand writes"};
  unsigned long long tot = 0;
  for (int i = 0; i < 6; ++i) tot += h[i];
  fprintf(stderr, "[mbxp stamps] %llu wave-iterations, %.0f cycles each:", h[6], (double)tot / (double)h[6]);
  for (int i = 0; i < 6; ++i) fprintf(stderr, "  %s %.0f (%.0f %%)", nm[i], (double)h[i] / (double)h[6], 100.0 * (double)h[i] / (double)tot);
  fprintf(stderr, "\n");
}
#define MBXP_STAMP(i) do { const unsigned long long t_ = clock64(); st_acc[i] += t_ - st_t; st_t = t_; } while (0)
#else
#define MBXP_STAMP(i) do { } while (0)
#endif

template <int K, int KSF, bool WIDE, int SCH>    // SCH: UDA_SPLIT_BF16X2 or UDA_SPLIT_F16X2 (two pieces per operand)
__global__ __launch_bounds__(512, 2) void mbxp_kernel(MbxArgs a) {
  static_assert(split_np(SCH) == 2, "the self-overlapping kernel is laid out for two pieces per operand");
  constexpr int NW = 8;
  constexpr int TH = mbxd_cfg(K, 1, WIDE).th, TW = mbxd_cfg(K, 1, WIDE).tw;
  constexpr int IH = TH + K - 1, IW = TW + K - 1;
  constexpr int NP = IH * IW;
  static_assert(NP <= 256, "input tile must fit 8 slices of 32 pixels");
  constexpr int NPP = 256;
  constexpr int NG = NW * 2;
  constexpr int ES = 33;
  constexpr int ETW = mbx_et_w(K, 1);         // E transposed, [channel][pixel] (see mbx_et_w); 32 * CP <= NPP * ES floats per buffer
  constexpr bool ET = ETW != 0;
  constexpr int CP = mbx_et_pitch(ETW);
  constexpr int XW = mbxd_cfg(K, 1, WIDE).xw;
  static_assert(!ET || (IW % ETW == 0 && XW % ETW == 0), "aligned wide accesses of the transposed slab");
  constexpr int UPR = TW / XW;
  constexpr int NUNIT = TH * UPR;
  constexpr int UPT = (NUNIT + NG - 1) / NG;  // units per thread: 3 (3x3) / 1 (5x5); wide: 1 / 2 (30 units: two thread groups idle in the second round)
  constexpr bool UNIT_GUARD = (NUNIT % NG) != 0;
  constexpr int NCOL = XW + K - 1;
  constexpr int BSLAB = KSF * 2 * 64;
  constexpr int NPAR = (K * K + 2) * 32;
  extern __shared__ __attribute__((aligned(16))) float plds[];
  float* E = plds;                            // [2][NPP][ES]
  float* red = E + (size_t)2 * NPP * ES;      // [2][NG][32]
  float* par = red + 2 * NG * 32;             // [3][NPAR]
  uint4* Bs = (uint4*)(par + 3 * NPAR);       // [2][BSLAB]
  float* mks = (float*)(Bs + 2 * BSLAB);      // [2][32 * NCH]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z / a.ch_groups, b_in = b / a.in_div;
  const int oy0 = blockIdx.y * TH, ox0 = blockIdx.x * TW;
  const int iy0 = oy0 - a.pad_t, ix0 = ox0 - a.pad_l;
  const float* xin = a.in + (size_t)b_in * a.H * a.W * a.Cin;
  const int NCH = (a.Cmid + 31) >> 5;
  // this block's slabs [chb, che) (see mbxd_kernel); buffers rotate with the slab's position in the range, r = ch - chb
  const int ch_per = (NCH + a.ch_groups - 1) / a.ch_groups;
  const int chb = (int)(blockIdx.z % a.ch_groups) * ch_per;
  const int che = chb + ch_per < NCH ? chb + ch_per : NCH;
  if (chb >= che) return;
  const uint4* Wp = (const uint4*)a.wsplit;

  // ---- operands of the first two slabs -> LDS
  for (int f = tid; f < 2 * BSLAB; f += 512) {
    const int sl = f / BSLAB, r = f - sl * BSLAB;
    const int ks = r >> 7, rest = r & 127;
    const int ch = chb + sl < che ? chb + sl : che - 1;
    Bs[f] = Wp[(((size_t)ks * NCH + ch) * 2 + (rest >> 6)) * 64 + (rest & 63)];
  }
  for (int f = tid; f < 2 * NPAR; f += 512) {
    const int sl = f / NPAR;
    par[f] = a.wpar[(size_t)(chb + sl < che ? chb + sl : che - 1) * NPAR + (f - sl * NPAR)];
  }
  for (int f = tid; f < 2 * 32 * NCH; f += 512) {
    const int which = f / (32 * NCH), col = f - which * 32 * NCH;
    const float* m = which ? a.mask1 : a.mask0;
    mks[f] = ((m && col < a.Cmid) ? m[(size_t)b * a.Cmid + col] : 1.f) * UDA_NEG_LN2;
  }
  bf16x8 ah[KSF], al[KSF];
  float amax = 0.f;                           // fp16 pieces: largest operand magnitude this lane has split
  {
    const int p = wave * 32 + li;
    const int iy = iy0 + p / IW, ix = ix0 + p % IW;
    const bool in = (p < NP) && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    const float* px = xin + ((size_t)(in ? iy : 0) * a.W + (in ? ix : 0)) * a.Cin;
#pragma unroll
    for (int ks = 0; ks < KSF; ++ks) {
      const int k = ks * 16 + 8 * lh;
      float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
      if (in && k < a.Cin) {
        v0 = *(const float4*)(px + k);
        v1 = *(const float4*)(px + k + 4);
      }
      if (k == a.Cin) v0.x = in ? 1.f : 0.f;
      bf16x8 pc[2];
      split_parts<SCH>(v0, v1, pc, amax);
      ah[ks] = pc[0];
      al[ks] = pc[1];
    }
  }
  split_report<SCH>(amax, a.oor);
  __syncthreads();

  const int c = tid & 31, g = tid >> 5;
  const size_t tile = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  constexpr int B_PER = (BSLAB + 511) / 512;
  constexpr int P_PER = (NPAR + 511) / 512;
  // loop-invariant addressing of the depthwise stage (see mbxb_kernel)
  int eoff[UPT];
  unsigned ooff[UPT];
#pragma unroll
  for (int ui = 0; ui < UPT; ++ui) {
    const int u = (UNIT_GUARD && g + NG * ui >= NUNIT) ? 0 : g + NG * ui;      // (an idle unit computes unit 0 again and stores nothing)
    const int orow = u / UPR, oxs = (u % UPR) * XW;
    eoff[ui] = ET ? c * CP + orow * IW + oxs : (orow * IW + oxs) * ES + c;
    ooff[ui] = (unsigned)((orow * a.Wo + oxs) * a.Cmid + c) * 4u;      // bytes
  }
  float* const obase = a.out + (((size_t)b * a.Ho + oy0) * a.Wo + ox0) * a.Cmid;
  const unsigned cm = (unsigned)a.Cmid;
  const bool full = (oy0 + TH <= a.Ho) && (ox0 + TW <= a.Wo) && ((a.Cmid & 31) == 0);   // (Cin 192 / 208: whole slabs)
  // activated accumulator -> E buffer (registers 4q .. 4q+3 of a lane = pixels 8q .. 8q+3 of its half of the slice)
  auto e_store = [&](float* Eb, const f32x16& acc) {
    if constexpr (ET) {
      et_store_slice<ETW>(Eb + li * CP + wave * 32 + 4 * lh, acc);
    } else {
      float* ep = Eb + (size_t)(wave * 32 + 4 * lh) * ES + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) ep[((r & 3) + 8 * (r >> 2)) * ES] = swish_core(acc[r]);    // x (-ln 2) x dropout scale: after the depthwise
    }
  };

  // ---- slab 0: expand -> E[0] (nothing to overlap with yet)
  {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KSF; ++ks) {
      const bf16x8 bh = __builtin_bit_cast(bf16x8, Bs[(ks * 2 + 0) * 64 + lane]);
      const bf16x8 bl = __builtin_bit_cast(bf16x8, Bs[(ks * 2 + 1) * 64 + lane]);
      acc = mfma16<SCH>(al[ks], bh, acc);
      acc = mfma16<SCH>(ah[ks], bl, acc);
      acc = mfma16<SCH>(ah[ks], bh, acc);
    }
    e_store(E, acc);
  }

#ifdef UDA_MBXP_STAMPS
  unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_t = clock64();
#endif
  for (int ch = chb; ch < che; ++ch) {
    __syncthreads();      // E[r & 1], Bs[(r + 1) & 1], par[(r + 1) % 3], red[(r - 1) & 1] are complete
    MBXP_STAMP(0);
    const int r = ch - chb;
    const int col = ch * 32 + c;
    const bool dcol = col < a.Cmid;
    const float* Ec = E + (size_t)(r & 1) * NPP * ES;
    float* En = E + (size_t)((r + 1) & 1) * NPP * ES;
    const float* pcur = par + (r % 3) * NPAR;
    const uint4* bnext = Bs + (size_t)((r + 1) & 1) * BSLAB;
    const bool more = ch + 1 < che, more2 = ch + 2 < che;
    if (r > 0 && a.se_partial && g == 0 && (ch - 1) * 32 + c < a.Cmid) {      // SE tile sums of the previous slab
      const float* rp = red + ((r - 1) & 1) * NG * 32;
      float t = rp[c];
#pragma unroll
      for (int gg = 1; gg < NG; ++gg) t += rp[gg * 32 + c];
      a.se_partial[((size_t)b * a.n_tiles + tile) * a.Cmid + (ch - 1) * 32 + c] = t;
    }
    // ---- operands of slab ch + 2 are requested now and written to LDS at the end of the iteration
    uint4 nb[B_PER];
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int f = tid + 512 * i;
      const int ks = f >> 7, rest = f & 127;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (more2 && f < BSLAB) v = Wp[(((size_t)ks * NCH + (ch + 2)) * 2 + (rest >> 6)) * 64 + (rest & 63)];
      nb[i] = v;
    }
    float np_[P_PER];
    const float* wnext = a.wpar + (size_t)(more2 ? ch + 2 : ch) * NPAR;
#pragma unroll
    for (int i = 0; i < P_PER; ++i) {
      const int f = tid + 512 * i;
      np_[i] = wnext[f < NPAR ? f : 0];
    }
    // ---- depthwise of slab ch, with the MFMAs of slab ch + 1 issued in between (one scheduling region)
    float wk[K * K];
#pragma unroll
    for (int t = 0; t < K * K; ++t) wk[t] = pcur[t * 32 + c];
    const float sc1 = pcur[K * K * 32 + c] * mks[ch * 32 + c], sh1 = pcur[(K * K + 1) * 32 + c];
    const float mk1 = mks[32 * NCH + ch * 32 + c];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float dacc[UPT][XW];
#pragma unroll
    for (int ui = 0; ui < UPT; ++ui)
#pragma unroll
      for (int o = 0; o < XW; ++o) dacc[ui][o] = 0.f;
    // The region is laid out by hand, chunk by chunk, and fenced with sched_barrier so that it stays that way (left to
    // itself the scheduler groups all MFMAs in front of all FMAs, and the in-order wave then idles through every one of
    // the dependent MFMAs): one MFMA - 8 passes on the matrix core - then the FMAs, the E reads of the NEXT tap row and
    // the fragment reads of the NEXT k-step that fit under it.
    constexpr int ROWS = UPT * K;                  // (unit, tap row) steps of the depthwise
    constexpr int MPR = (KSF + ROWS - 1) / ROWS;   // k-steps of the expand issued per step
    constexpr int MPS = 3 * MPR;                   // MFMAs per step
    constexpr int NF = K * XW;                     // FMAs per step
    constexpr int FPM = (NF + MPS - 1) / MPS, RPM = (NCOL + MPS - 1) / MPS;
    auto e_row = [&](int st) -> const float* {
      const int ui = st / K, ky = st % K;
      return Ec + eoff[ui] + (ET ? ky * IW : ky * IW * ES);
    };
    float rowv[2][NCOL];
    constexpr int NRD = ET ? NCOL / (ET ? ETW : 1) : 1;      // wide reads per tap row of the transposed slab
    static_assert(!ET || NCOL % (ET ? ETW : 1) == 0, "whole 16- / 8-byte reads");
    {
      const float* er = e_row(0);
      if constexpr (ET) {
#pragma unroll
        for (int jj = 0; jj < NRD; ++jj) et_read<ETW>(rowv[0], er, jj);
      } else {
#pragma unroll
        for (int j = 0; j < NCOL; ++j) rowv[0][j] = er[j * ES];
      }
    }
    bf16x8 bh_c = __builtin_bit_cast(bf16x8, bnext[lane]), bl_c = __builtin_bit_cast(bf16x8, bnext[64 + lane]);
    bf16x8 bh_n = bh_c, bl_n = bl_c;
    __builtin_amdgcn_sched_barrier(0);
    MBXP_STAMP(1);
#pragma unroll
    for (int st = 0; st < ROWS; ++st) {
      const int ui = st / K, ky = st % K;
      const float* ern = e_row(st + 1 < ROWS ? st + 1 : st);
#pragma unroll
      for (int m = 0; m < MPS; ++m) {
        const int ks = st * MPR + m / 3, t = m % 3;
        if (ks < KSF) {
          if (t == 0 && ks + 1 < KSF) {
            bh_n = __builtin_bit_cast(bf16x8, bnext[((ks + 1) * 2 + 0) * 64 + lane]);
            bl_n = __builtin_bit_cast(bf16x8, bnext[((ks + 1) * 2 + 1) * 64 + lane]);
          }
          if (t == 0) acc = mfma16<SCH>(al[ks], bh_c, acc);
          if (t == 1) acc = mfma16<SCH>(ah[ks], bl_c, acc);
          if (t == 2) {
            acc = mfma16<SCH>(ah[ks], bh_c, acc);
            bh_c = bh_n; bl_c = bl_n;
          }
        }
#pragma unroll
        for (int f = m * FPM; f < (m + 1) * FPM; ++f)
          if (f < NF) {
            const int kx = f / XW, o = f % XW;
            dacc[ui][o] = fmaf(rowv[st & 1][o + kx], wk[ky * K + kx], dacc[ui][o]);
          }
        if (st + 1 < ROWS) {
          if constexpr (ET) {      // the next tap row: NRD wide reads, spread over the MFMA gaps of this step
            constexpr int GAP = MPS / NRD > 0 ? MPS / NRD : 1;
            if (m % GAP == 0 && m / GAP < NRD) et_read<ETW>(rowv[(st + 1) & 1], ern, m / GAP);
            static_assert(!ET || (MPS >= NRD), "a gap per wide read");
          } else {
#pragma unroll
            for (int r = m * RPM; r < (m + 1) * RPM; ++r)
              if (r < NCOL) rowv[(st + 1) & 1][r] = ern[r * ES];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // (pin the sums here: otherwise the FMAs are sunk into the guarded store blocks below, behind all the MFMAs)
#pragma unroll
    for (int ui = 0; ui < UPT; ++ui)
#pragma unroll
      for (int o = 0; o < XW; ++o) asm volatile("" : "+v"(dacc[ui][o]));
    MBXP_STAMP(2);
    // ---- outputs of slab ch
    float ssum = 0.f;
    float* const ob = obase + ch * 32;            // uniform
    auto store_units = [&](auto guard) {
      constexpr bool GUARD = decltype(guard)::value;
#pragma unroll
      for (int ui = 0; ui < UPT; ++ui) {
        if constexpr (UNIT_GUARD) {
          if (g + NG * ui >= NUNIT) continue;
        }
        if constexpr (GUARD) {
          if (oy0 + (g + NG * ui) / UPR >= a.Ho) continue;
        }
#pragma unroll
        for (int o = 0; o < XW; ++o) {
          bool ok = true;
          if constexpr (GUARD) ok = ox0 + (int)((g + NG * ui) % UPR) * XW + o < a.Wo;
          if (ok) {
            const float v = swish_folded(fmaf(dacc[ui][o], sc1, sh1), mk1);
            store_uniform_base(ob + (size_t)o * cm, ooff[ui], v);
            ssum += v;
          }
        }
      }
    };
    if (full) store_units(std::false_type());
    else if (dcol) store_units(std::true_type());
    if (a.se_partial) red[((r & 1) * NG + g) * 32 + c] = ssum;
    MBXP_STAMP(3);
    // ---- slab ch + 1: activate -> the other E buffer (its readers, depthwise ch - 1, finished before the barrier above)
    if (more) e_store(En, acc);
    MBXP_STAMP(4);
    // ---- slab ch + 2 operands -> LDS: Bs[r & 1] (its MFMAs were issued one iteration ago), par[(r + 2) % 3]
    {
      uint4* bw = Bs + (size_t)(r & 1) * BSLAB;
#pragma unroll
      for (int i = 0; i < B_PER; ++i) {
        const int f = tid + 512 * i;
        if (more2 && f < BSLAB) bw[f] = nb[i];
      }
      float* pw = par + ((r + 2) % 3) * NPAR;
#pragma unroll
      for (int i = 0; i < P_PER; ++i) {
        const int f = tid + 512 * i;
        if (more2 && f < NPAR) pw[f] = np_[i];
      }
    }
    MBXP_STAMP(5);
  }
#ifdef UDA_MBXP_STAMPS
  if (lane == 0) {
    for (int i = 0; i < 6; ++i) atomicAdd(&g_mbxp_stamps[i], st_acc[i]);
    atomicAdd(&g_mbxp_stamps[6], (unsigned long long)(che - chb));
  }
#endif
  if (a.se_partial) {
    __syncthreads();
    const int lc = (che - 1) * 32 + c;
    if (g == 0 && lc < a.Cmid) {
      const float* rp = red + ((che - 1 - chb) & 1) * NG * 32;
      float t = rp[c];
#pragma unroll
      for (int gg = 1; gg < NG; ++gg) t += rp[gg * 32 + c];
      a.se_partial[((size_t)b * a.n_tiles + tile) * a.Cmid + lc] = t;
    }
  }
}

// Slab groups of a deep fused MBConv launch: a launch of `blocks` blocks on a device that holds per_cu of them per CU is
// split along the channels until it fills the device (at most 16 ways, at least 2 slabs per block: every group re-reads and re-splits
// the input tile, which is what bounds the split).  A batch of 32 images never splits; one image with T = 10 has 60-90 blocks of 36
// slabs each in the last blocks (4 groups), one image under head-only MC 9 blocks (16 groups: fused MBConv time of a one-image serve
// 0.47 -> 0.37 ms against the 4-way limit of round 4; UDA_MBX_SPLIT_MAX / UDA_MBX_SPLIT_SLABS, A/B in one job).  UDA_MBX_SPLIT=0: never.
// The slabs of a tile are independent: the result does not depend on the split.
static int mbx_ch_groups(long long blocks, int per_cu, int n_slabs) {
  static int on = -1, n_cu = 0, gmax = 16, smin = 2;
  if (on < 0) {
    const char* e = getenv("UDA_MBX_SPLIT");
    on = e ? atoi(e) : 1;
    if (const char* m = getenv("UDA_MBX_SPLIT_MAX")) gmax = atoi(m) > 0 ? atoi(m) : 16;
    if (const char* m = getenv("UDA_MBX_SPLIT_SLABS")) smin = atoi(m) > 0 ? atoi(m) : 2;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n_cu = 0;
    (void)hipGetLastError();
  }
  if (!on || n_cu <= 0 || blocks <= 0) return 1;
  long long g = (long long)n_cu * per_cu / blocks;
  if (g > gmax) g = gmax;
  if (g > n_slabs / smin) g = n_slabs / smin;
  if (g < 1) g = 1;
  const long long per = (n_slabs + g - 1) / g;          // slabs per group -> no empty group at the end
  return (int)((n_slabs + per - 1) / per);
}

template <int K, int KSF, bool WIDE, int SCH>
static void launch_mbxp_tw(const MbxArgs& a, int rows, hipStream_t s) {
  constexpr int TH = mbxd_cfg(K, 1, WIDE).th, TW = mbxd_cfg(K, 1, WIDE).tw;
  const size_t lds = ((size_t)2 * 256 * 33 + 2 * 16 * 32 + 3 * (K * K + 2) * 32 + 2 * 32 * ((a.Cmid + 31) / 32)) * sizeof(float) +
                     (size_t)2 * KSF * 2 * 64 * sizeof(uint4);
  static size_t attr_lds = 64 * 1024;
  if (lds > attr_lds) {
    hipFuncSetAttribute((const void*)mbxp_kernel<K, KSF, WIDE, SCH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_lds = lds;
  }
  dim3 grid((a.Wo + TW - 1) / TW, (a.Ho + TH - 1) / TH, rows);
  MbxArgs b = a;
  b.ch_groups = mbx_ch_groups((long long)grid.x * grid.y * rows, 1, (a.Cmid + 31) / 32);
  grid.z = (unsigned)(rows * b.ch_groups);
  hipLaunchKernelGGL((mbxp_kernel<K, KSF, WIDE, SCH>), grid, dim3(512), lds, s, b);
#ifdef UDA_MBXP_STAMPS
  static int n_launch = 0;
  if (K == 5 && ++n_launch % 9 == 0) mbxp_stamp_dump();
#endif
}

template <int K, int KSF, int SCH>
static void launch_mbxp_ts(const MbxArgs& a, int rows, hipStream_t s) {
  if (mbxd_wide(a.Ho, a.Wo, K, 1)) launch_mbxp_tw<K, KSF, true, SCH>(a, rows, s);
  else launch_mbxp_tw<K, KSF, false, SCH>(a, rows, s);
}
template <int K, int KSF, int PARTS, int S> static void launch_mbxd_t(const MbxArgs& a, int rows, hipStream_t s);
template <int K, int KSF>
static void launch_mbxp_t(const MbxArgs& a, int rows, hipStream_t s) {
  if (a.wparts == UDA_SPLIT_F16X2) {
    if constexpr (K == 3) {
      // (the fp16 instances of the 3x3 variant on 16-column tiles do not build: hipcc forms the "uniform" store bases of their
      // stores with vector arithmetic and hands the scalar-base store of store_uniform_base a vector register pair; those maps
      // - none at 1280 x 768, where blocks 12-15 take 20-column tiles - run the two-phase kernel)
      if (!mbxd_wide(a.Ho, a.Wo, K, 1)) { launch_mbxd_t<K, KSF, UDA_SPLIT_F16X2, 1>(a, rows, s); return; }
      launch_mbxp_tw<K, KSF, true, UDA_SPLIT_F16X2>(a, rows, s);
    } else {
      launch_mbxp_ts<K, KSF, UDA_SPLIT_F16X2>(a, rows, s);
    }
    return;
  }
  launch_mbxp_ts<K, KSF, UDA_SPLIT_BF16X2>(a, rows, s);
}

// Dynamic LDS of the fused MBConv launch the executor will make for this op (the formulas of launch_mbxb_t / launch_mbxd_tw /
// launch_mbxp_tw): uda_create checks it against the 160 KB of a gfx950 CU and names the op, instead of a refused launch
// surfacing in the middle of a run.
size_t mbx_lds_bytes(int Cin, int Cmid, int k, int stride, int scheme, int Ho, int Wo) {
  const int npc = uda_split_pieces(scheme), ksf = (Cin + 1 + 15) / 16, nch = (Cmid + 31) / 32;
  const size_t par = (size_t)(k * k + 2) * 32;
  if (Cin <= 48) {        // mbxb_kernel
    const MbxCfgB c = mbxb_cfg(k, stride);
    const int ih = (c.th - 1) * stride + k, iw = (c.tw - 1) * stride + k, npp = (ih * iw + 31) / 32 * 32;
    const int etw = mbx_et_w(k, stride);
    return ((etw ? (size_t)32 * mbx_et_pitch(etw) : (size_t)npp * 32) + 8 * 32 + 2 * par) * sizeof(float) + (size_t)ksf * npc * 64 * sizeof(uint4);
  }
  static int pipe = -1;
  if (pipe < 0) { const char* e = getenv("UDA_MBXP"); pipe = e ? atoi(e) : 1; }
  const bool wide = mbxd_wide(Ho, Wo, k, stride);
  const bool p2 = pipe && stride == 1 && ksf >= 13 && npc == 2 && !(scheme == UDA_SPLIT_F16X2 && k == 3 && !wide);      // mbxp_kernel
  if (p2) return ((size_t)2 * 256 * 33 + 2 * 16 * 32 + 3 * par + 2 * 32 * nch) * sizeof(float) + (size_t)2 * ksf * 2 * 64 * sizeof(uint4);
  return ((size_t)256 * 33 + 16 * 32 + 2 * par + 2 * 32 * nch) * sizeof(float) + (size_t)ksf * npc * 64 * sizeof(uint4);
}

bool mbxd_supported(int Cin, int Cmid, int k, int stride) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("UDA_FUSE_MBXD"); on = e ? atoi(e) : 1; }
  static int s2 = -1;            // UDA_FUSE_MBXD_S2=0: stride-2 deep blocks run unfused (1x1 expand + depthwise), as before round 3
  if (s2 < 0) { const char* e = getenv("UDA_FUSE_MBXD_S2"); s2 = e ? atoi(e) : 1; }
  const int ksf = (Cin + 1 + 15) / 16;
  static int maxksf = -1;
  if (maxksf < 0) { const char* e = getenv("UDA_MBXD_MAXKSF"); maxksf = e ? atoi(e) : 14; }
  return on && (stride == 1 || (stride == 2 && s2)) && (k == 3 || k == 5) && Cin % 8 == 0 && Cin > 48 &&
         (ksf == 6 || ksf == 8 || ksf == 13 || ksf == 14) && ksf <= maxksf && Cmid % 4 == 0;
}

int mbxd_tiles(int Ho, int Wo, int k, int stride) {
  const MbxCfgD c = mbxd_cfg(k, stride, mbxd_wide(Ho, Wo, k, stride));
  return ((Ho + c.th - 1) / c.th) * ((Wo + c.tw - 1) / c.tw);
}

template <int K, int KSF, int PARTS, int S, bool WIDE>
static void launch_mbxd_tw(const MbxArgs& a, int rows, hipStream_t s) {
  constexpr int TH = mbxd_cfg(K, S, WIDE).th, TW = mbxd_cfg(K, S, WIDE).tw;
  const size_t lds = ((size_t)256 * 33 + 16 * 32 + 2 * (K * K + 2) * 32 + 2 * 32 * ((a.Cmid + 31) / 32)) * sizeof(float) +
                     (size_t)KSF * split_np(PARTS) * 64 * sizeof(uint4);
  static size_t attr_lds = 64 * 1024;      // above the default limit the kernel needs an explicit opt-in
  if (lds > attr_lds) {
    hipFuncSetAttribute((const void*)mbxd_kernel<K, KSF, PARTS, S, WIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_lds = lds;
  }
  dim3 grid((a.Wo + TW - 1) / TW, (a.Ho + TH - 1) / TH, rows);
  MbxArgs b = a;
  b.ch_groups = mbx_ch_groups((long long)grid.x * grid.y * rows, (KSF <= 8 && PARTS != 3) ? 2 : 1, (a.Cmid + 31) / 32);
  grid.z = (unsigned)(rows * b.ch_groups);
  hipLaunchKernelGGL((mbxd_kernel<K, KSF, PARTS, S, WIDE>), grid, dim3(512), lds, s, b);
}

template <int K, int KSF, int PARTS, int S>
static void launch_mbxd_t(const MbxArgs& a, int rows, hipStream_t s) {
  if constexpr (S == 1) {
    if (mbxd_wide(a.Ho, a.Wo, K, 1)) { launch_mbxd_tw<K, KSF, PARTS, 1, true>(a, rows, s); return; }
  }
  launch_mbxd_tw<K, KSF, PARTS, S, false>(a, rows, s);
}

template <int PARTS, int S>
static void launch_mbxd_p(const MbxArgs& a, int rows, int k, int ksf, hipStream_t s) {
  if (k == 3) {
    switch (ksf) {
      case 6: launch_mbxd_t<3, 6, PARTS, S>(a, rows, s); break;
      case 8: launch_mbxd_t<3, 8, PARTS, S>(a, rows, s); break;
      case 13: launch_mbxd_t<3, 13, PARTS, S>(a, rows, s); break;
      default: launch_mbxd_t<3, 14, PARTS, S>(a, rows, s); break;
    }
  } else {
    switch (ksf) {
      case 6: launch_mbxd_t<5, 6, PARTS, S>(a, rows, s); break;
      case 8: launch_mbxd_t<5, 8, PARTS, S>(a, rows, s); break;
      case 13: launch_mbxd_t<5, 13, PARTS, S>(a, rows, s); break;
      default: launch_mbxd_t<5, 14, PARTS, S>(a, rows, s); break;
    }
  }
}

void launch_mbxd(const MbxArgs& a, int rows, int k, int stride, hipStream_t s) {
  const int ksf = (a.Cin + 1 + 15) / 16;
  if (stride == 2) {                // the first block of a stage: the two-phase kernel on the stride-2 tiles
    if (a.wparts == UDA_SPLIT_BF16X3) launch_mbxd_p<3, 2>(a, rows, k, ksf, s);
    else if (a.wparts == UDA_SPLIT_F16X2) launch_mbxd_p<4, 2>(a, rows, k, ksf, s);
    else launch_mbxd_p<2, 2>(a, rows, k, ksf, s);
    return;
  }
  static int pipe = -1;
  if (pipe < 0) { const char* e = getenv("UDA_MBXP"); pipe = e ? atoi(e) : 1; }
  if (a.wparts == UDA_SPLIT_BF16X3) {     // six cross terms: three pieces per operand, the two-phase kernel
    launch_mbxd_p<3, 1>(a, rows, k, ksf, s);
    return;
  }
  if (pipe && ksf >= 13) {          // one block per CU anyway: the self-overlapping variant
    if (k == 3) { if (ksf == 13) launch_mbxp_t<3, 13>(a, rows, s); else launch_mbxp_t<3, 14>(a, rows, s); }
    else { if (ksf == 13) launch_mbxp_t<5, 13>(a, rows, s); else launch_mbxp_t<5, 14>(a, rows, s); }
    return;
  }
  if (a.wparts == UDA_SPLIT_F16X2) launch_mbxd_p<4, 1>(a, rows, k, ksf, s);
  else launch_mbxd_p<2, 1>(a, rows, k, ksf, s);
}

// depthwise-side operands of the fused kernels, one contiguous block per 32-channel slab:
// [slab][K*K taps | BN scale * -log2(e) | BN shift * -log2(e)][32 channels], zero beyond Cmid
size_t mbx_par_floats(int Cmid, int k) { return (size_t)((Cmid + 31) / 32) * (k * k + 2) * 32; }
void mbx_pack_params(const float* wd, const float* sc1, const float* sh1, int Cmid, int k, float* out) {
  const int nch = (Cmid + 31) / 32, rows = k * k + 2;
  const float L = -1.4426950408889634f;
  for (int ch = 0; ch < nch; ++ch)
    for (int r = 0; r < rows; ++r)
      for (int j = 0; j < 32; ++j) {
        const int col = ch * 32 + j;
        float v = 0.f;
        if (col < Cmid) v = r < k * k ? wd[(size_t)r * Cmid + col] : (r == k * k ? sc1[col] * L : sh1[col] * L);
        out[((size_t)ch * rows + r) * 32 + j] = v;
      }
}

// expand kernel [Cin][Cmid] times the BN scale, plus the BN shift as row Cin -> packed split-bf16 fragments
size_t mbxb_packed_elems(int Cin, int Cmid, int scheme) { return pwb_packed_elems(Cin + 1, Cmid, scheme); }
// perm16: rows 0..15 stored in the k order of an accumulator tile used as the A operand (FUSE0): slot 8 h + j holds
// channel (j & 3) + 8 (j >> 2) + 4 h
void mbxb_pack_weights(const float* we, const float* sc0, const float* sh0, int Cin, int Cmid, uint16_t* out, bool perm16, int scheme, float* stats) {
  float* w = (float*)malloc((size_t)(Cin + 1) * Cmid * sizeof(float));
  // the GEMM delivers y = -log2(e) * BN(x W): see swish_folded
  const float L = -1.4426950408889634f;
  for (int k = 0; k < Cin; ++k)
    for (int n = 0; n < Cmid; ++n) w[(size_t)k * Cmid + n] = we[(size_t)k * Cmid + n] * sc0[n] * L;
  for (int n = 0; n < Cmid; ++n) w[(size_t)Cin * Cmid + n] = sh0[n] * L;
  if (perm16 && Cin == 16) {
    float* t = (float*)malloc((size_t)16 * Cmid * sizeof(float));
    memcpy(t, w, (size_t)16 * Cmid * sizeof(float));
    for (int h = 0; h < 2; ++h)
      for (int j = 0; j < 8; ++j)
        memcpy(w + (size_t)(8 * h + j) * Cmid, t + (size_t)((j & 3) + 8 * (j >> 2) + 4 * h) * Cmid, (size_t)Cmid * sizeof(float));
    free(t);
  }
  if (stats) {
    double mx = 0.0, s2 = 0.0;
    const size_t n = (size_t)(Cin + 1) * Cmid;
    for (size_t i = 0; i < n; ++i) { const double v = fabs((double)w[i]); if (v > mx) mx = v; s2 += v * v; }
    stats[0] = (float)mx;
    stats[1] = (float)sqrt(s2 / (double)(n ? n : 1));
  }
  // (no power-of-two pre-scale here: the expand accumulator feeds the swish directly, there is no free place to undo one)
  pwb_pack_weights(w, Cin + 1, Cmid, scheme, out);
  free(w);
}

// FUSE0 operands: W0^T [32 projected channels (16 real)][32 input channels] times the BN scale, then the BN shift [32]
void mbxb_pack_proj(const float* w0 /*[c0][cout]*/, const float* sc, const float* sh, int c0, int cout, float* out /*[32*32 + 32]*/) {
  for (int r = 0; r < 32; ++r)
    for (int k = 0; k < 32; ++k) out[r * 32 + k] = (r < cout && k < c0) ? w0[(size_t)k * cout + r] * sc[r] : 0.f;
  for (int r = 0; r < 32; ++r) out[32 * 32 + r] = r < cout ? sh[r] : 0.f;
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

// IEEE binary16, round to nearest even, subnormals kept (what v_cvt_pk_f16_f32 does on the device); finite inputs
static inline uint16_t f32_to_f16_rne(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
  const uint32_t a = u & 0x7FFFFFFFu;
  if (a >= 0x47800000u) return (uint16_t)(sign | 0x7C00u);               // >= 65536 (or inf / NaN): infinity
  if (a < 0x38800000u) {                                                  // below 2^-14: subnormal (or zero)
    if (a < 0x33000000u) return sign;                                     // below 2^-25: rounds to zero
    const int e = (int)(a >> 23);                                         // biased float exponent, 102 .. 112
    const uint32_t m = (a & 0x7FFFFFu) | 0x800000u;                       // 24-bit significand
    const int shift = 126 - e;                                            // value = m * 2^(e - 150); units of 2^-24 -> m >> shift
    const uint32_t q = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    uint32_t r = q;
    if (rem > half || (rem == half && (q & 1u))) ++r;
    return (uint16_t)(sign | r);                                          // (a carry into 0x400 is the smallest normal: correct)
  }
  uint32_t r = a + 0xFFFu + ((a >> 13) & 1u);                             // round the 13 dropped bits to nearest even
  r = ((r >> 13) - (112u << 10));                                         // rebias 127 -> 15
  if (r >= 0x7C00u) r = 0x7C00u;                                          // 65520 and above round to infinity
  return (uint16_t)(sign | r);
}
static inline float f16_to_f32(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  const uint32_t e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
  float x;
  if (e == 0) {
    x = ldexpf((float)m, -24);
    uint32_t u; memcpy(&u, &x, 4); u |= sign; memcpy(&x, &u, 4);
    return x;
  }
  const uint32_t u = sign | (e == 31 ? 0x7F800000u | (m << 13) : ((e + 112u) << 23) | (m << 13));
  memcpy(&x, &u, 4);
  return x;
}

size_t pwb_packed_elems(int K, int N, int scheme) {
  return (size_t)((K + 15) / 16) * ((N + 31) / 32) * uda_split_pieces(scheme) * 64 * 8;
}

float split_weight_scale(const float* w, size_t n) {
  float mx = 0.f;
  for (size_t i = 0; i < n; ++i) { const float v = fabsf(w[i]); if (v > mx) mx = v; }
  if (!(mx > 0.f) || !(mx < 3.0e38f)) return 1.0f;
  int e = 0;
  frexpf(mx, &e);                 // mx = f * 2^e, f in [0.5, 1)  ->  mx * 2^(14 - e) in [2^13, 2^14)
  return ldexpf(1.0f, 14 - e);
}

// w [K][N] float32 -> fragments [k-step][32-col tile][part][lane][8] bf16 of v_mfma_f32_32x32x16_bf16's B operand:
// lane l holds B[16 s + 8 (l >> 5) + e][32 j + (l & 31)], e = 0..7
void pwb_pack_weights(const float* w, int K, int N, int scheme, uint16_t* out, float scale) {
  const int KS = (K + 15) / 16, NTL = (N + 31) / 32, parts = uda_split_pieces(scheme);
  const bool half = scheme == UDA_SPLIT_F16X2;
  for (int s = 0; s < KS; ++s)
    for (int j = 0; j < NTL; ++j)
      for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 8; ++e) {
          const int k = 16 * s + 8 * (l >> 5) + e, n = 32 * j + (l & 31);
          float r = (k < K && n < N) ? w[(size_t)k * N + n] * scale : 0.f;
          for (int p = 0; p < parts; ++p) {
            const uint16_t h = half ? f32_to_f16_rne(r) : f32_to_bf16_rne(r);
            out[((((size_t)s * NTL + j) * parts + p) * 64 + l) * 8 + e] = h;
            r -= half ? f16_to_f32(h) : bf16_to_f32(h);
          }
        }
}

}  // namespace uda
