// Fused separable convolution on the bf16 matrix cores (split-precision products as in kernels_pwb.hip).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "mfma_common.h"
#include "fuse_sample.h"

namespace uda {

// ---------------------------------------------------------------- fused separable convolution
// SeparableConv2D of the BiFPN nodes and of the class / box heads (efficientdet_keras.py:207-227,421-446,
// 584-626): depthwise 3x3 stride 1 (TF SAME, no bias / BN / activation) feeding the 1x1 convolution
// (+ bias, BN, swish, MC-dropout keep-scale).  The depthwise result never goes to HBM: the block computes it
// for its 8 x 16 pixel tile and all C channels on the VALU (3-row sliding window per thread, every input element
// fetched once per block column), splits it into bf16 pieces and writes it straight into the LDS image the MFMA
// stage reads as its A operand.  One barrier between the two stages; epilogue as in pwb_kernel.
// Six-term products (PARTS = 3, UDA_PW_TERMS=6): three bf16 images would be 55 KB (two blocks per CU); there the depthwise
// result goes to LDS as float32 (35 KB: three blocks per CU, as the three-term kernel) and a wave splits the 8 values of a
// fragment into its three pieces when it loads them - every wave owns its 32 pixel rows, so an element is still split once.
constexpr int SEP_TH = 8, SEP_TW = 16;

// TIN > 0 (round 5): the input is shared by the T = in_div samples of an image and carries a DEFERRED dropout site -
// keep-scales mask_in[sample row][channel] that the producing op did not apply (plan.py: the first layer of a head under
// head-only MC dropout).  A per-channel factor commutes with the depthwise conv, so the block computes the depthwise result
// of its tile ONCE (TIN = units per thread held in registers: 1 up to 64 channels, 2 up to 128), then per sample: scale,
// split, 1x1 on the matrix cores, epilogue (with this op's own site), store.  The producer's output stays one row per image
// (1 / T of the bytes), this op reads 1 / T of its input and does the depthwise arithmetic once per image instead of T times.
template <int NT, int PARTS, int OCC, int TIN = 0>     // NT = 32-column tiles of the 1x1 output handled per block (each wave: all of them)
__global__ __launch_bounds__(256, OCC) void sep_kernel(SepMulti m) {
  constexpr int BM = SEP_TH * SEP_TW;      // 128 pixels = 4 MFMA row tiles, one per wave
  // the problem of this block (uniform): one conv, or one of the pyramid levels of a head layer launched together
  SepArgs a = m.one;
  // block -> (sample row b, tile bx): consecutive block ids go round the 8 XCDs, so sample row b's tiles (of all levels) are
  // given to XCD b mod 8 in order - a tile's halo is then fetched into ONE L2 (rows beyond the last multiple of 8: plain order)
  int bx, b;
  {
    const int L = blockIdx.x, rows8 = m.rows & ~7;
    if (L < m.tiles * rows8) {
      const int j = L >> 3;
      b = (j / m.tiles) * 8 + (L & 7);
      bx = j % m.tiles;
    } else {
      const int l2 = L - m.tiles * rows8;
      b = rows8 + l2 / m.tiles;
      bx = l2 % m.tiles;
    }
  }
  if (m.n_lv > 0) {
    int l = 0;
    while (l + 1 < m.n_lv && bx >= m.tile0[l + 1]) ++l;
    bx -= m.tile0[l];
    const SepLevel& L = m.lv[l];
    a.in = L.in; a.out = L.out; a.wd = L.wd; a.wsplit = L.wsplit;
    a.bias = L.bias; a.bn_scale = L.bn_scale; a.bn_shift = L.bn_shift; a.mask = L.mask; a.mask_in = L.mask_in;
    a.H = L.H; a.W = L.W; a.wunscale = L.wunscale;
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char slds[];
  const int C = a.C, C4 = C >> 2;
  const int KS = (C + 15) >> 4;            // MFMA k-steps
  constexpr int NPC = split_np(PARTS);     // pieces per operand (PARTS names the scheme: UDA_SPLIT_*)
  constexpr bool F32A = PARTS == 3;        // A image kept as float32, split at the fragment loads
  float amax = 0.f;                        // fp16 pieces: largest operand magnitude this lane has split
  // bytes per A image row: KS * 16 bf16 (float32) + 16 pad (conflict-free ds_read_b128: 36- / 68-dword pitch at C = 64)
  const int arow = F32A ? KS * 64 + 16 : KS * 32 + 16;
  unsigned char* As = slds;                // [NPC][BM][arow] pieces, or [BM][arow] float32
  uint4* Bs = (uint4*)(slds + (size_t)(F32A ? 1 : NPC) * BM * arow);   // [KS][NT][NPC][64 lanes] x 16 B
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int b_in = TIN ? b : b / a.in_div;       // (TIN: the grid runs over images)
  const int tiles_x = (a.W + SEP_TW - 1) / SEP_TW;
  const int ty = bx / tiles_x, tx = bx - ty * tiles_x;
  const int oy0 = ty * SEP_TH, ox0 = tx * SEP_TW;
  const int nt0 = blockIdx.y * NT, n0 = nt0 * 32;
  const int NTL = (a.Cout + 31) >> 5;

  // ---- B: all k-steps of this block's column tiles, requested first (in flight during the depthwise stage)
  const uint4* Wp = (const uint4*)a.wsplit;
  const int b_total = KS * NT * NPC * 64;
  constexpr int B_MAX = 8;                 // uint4 per thread: K <= 128, NT <= 4, PARTS = 2 -> 8 * 4 * 2 * 64 / 256 = 16 (two rounds)
  for (int f0 = tid; f0 < b_total; f0 += 256 * B_MAX) {
    uint4 rb[B_MAX];
#pragma unroll
    for (int i = 0; i < B_MAX; ++i) {
      const int f = f0 + 256 * i;
      int q = f >> 6;
      const int part = q % NPC; q /= NPC;
      const int nt = q % NT, ks = q / NT;
      rb[i] = make_uint4(0u, 0u, 0u, 0u);
      if (f < b_total && nt0 + nt < NTL) rb[i] = Wp[(((size_t)ks * NTL + (nt0 + nt)) * NPC + part) * 64 + (f & 63)];
    }
#pragma unroll
    for (int i = 0; i < B_MAX; ++i) {
      const int f = f0 + 256 * i;
      if (f < b_total) Bs[f] = rb[i];
    }
  }

  // ---- depthwise 3x3: unit = (channel quad q, tile column x); 8 output rows with a 3-row register window
  const float* inb = a.in + (size_t)b_in * a.H * a.W * C;
  // split a depthwise value into its pieces -> A image row m = ry * 16 + x, channels 4q .. 4q + 3
  auto put_a = [&](float4 acc, int m, int q) {
    if constexpr (F32A) {
      *(float4*)(As + (size_t)m * arow + q * 16) = acc;
    } else {
      float r0 = acc.x, r1 = acc.y, r2 = acc.z, r3 = acc.w;
      split_track<PARTS>(amax, r0, r1);
      split_track<PARTS>(amax, r2, r3);
#pragma unroll
      for (int p = 0; p < NPC; ++p) {
        const unsigned u0 = pack_piece<PARTS>(r0, r1), u1 = pack_piece<PARTS>(r2, r3);
        *(uint2*)(As + (size_t)(p * BM + m) * arow + q * 8) = make_uint2(u0, u1);
        if (p + 1 < NPC) {
          r0 -= piece_lo<PARTS>(u0); r1 -= piece_hi<PARTS>(u0);
          r2 -= piece_lo<PARTS>(u1); r3 -= piece_hi<PARTS>(u1);
        }
      }
    }
  };
  auto dw_unit = [&](int u, auto&& sink) {       // sink(ry, value): the 8 output rows of unit u, one at a time
    const int q = u % C4, x = u / C4;
    float4 wk[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wk[t] = *(const float4*)(a.wd + (size_t)t * C + 4 * q);
    const int gx = ox0 + x;
    // all (8 + 2) x 3 input quads of the unit are requested up front: 30 independent 16-byte loads in flight,
    // one exposed memory latency per unit instead of one per output row
    float4 win[SEP_TH + 2][3];
#pragma unroll
    for (int r = 0; r < SEP_TH + 2; ++r) {
      const int iy = oy0 - 1 + r;
      const bool rowok = iy >= 0 && iy < a.H;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int ix = gx - 1 + j;
        win[r][j] = (rowok && ix >= 0 && ix < a.W) ? *(const float4*)(inb + ((size_t)iy * a.W + ix) * C + 4 * q)
                                                   : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int ry = 0; ry < SEP_TH; ++ry) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float4 v = win[ry + ky][kx];
          const float4 w = wk[ky * 3 + kx];
          acc.x = fmaf(v.x, w.x, acc.x);
          acc.y = fmaf(v.y, w.y, acc.y);
          acc.z = fmaf(v.z, w.z, acc.z);
          acc.w = fmaf(v.w, w.w, acc.w);
        }
      sink(ry, acc);
    }
  };
  float4 dwv[TIN ? TIN : 1][SEP_TH];        // TIN: the depthwise result of this thread's units, kept for the T samples
  if constexpr (TIN) {
#pragma unroll
    for (int ui = 0; ui < TIN; ++ui) {
      const int u = tid + 256 * ui;
      if (u < SEP_TW * C4) dw_unit(u, [&](int ry, float4 v) { dwv[ui][ry] = v; });
    }
  } else {
    for (int u = tid; u < SEP_TW * C4; u += 256) {
      const int q = u % C4, x = u / C4;
      dw_unit(u, [&](int ry, float4 v) { put_a(v, ry * SEP_TW + x, q); });      // (split row by row: nothing stays live)
    }
  }
  const int n_samp = TIN ? a.in_div : 1;
  for (int ts = 0; ts < n_samp; ++ts) {
  const int bo = TIN ? b * n_samp + ts : b;   // output sample row
  if constexpr (TIN) {
    // this sample's A image: the shared depthwise result times the deferred keep-scale of its channels
#pragma unroll
    for (int ui = 0; ui < TIN; ++ui) {
      const int u = tid + 256 * ui;
      if (u < SEP_TW * C4) {
        const int q = u % C4, x = u / C4;
        const float4 mi = *(const float4*)(a.mask_in + (size_t)bo * C + 4 * q);
#pragma unroll
        for (int ry = 0; ry < SEP_TH; ++ry) {
          float4 v = dwv[ui][ry];
          v.x *= mi.x; v.y *= mi.y; v.z *= mi.z; v.w *= mi.w;
          put_a(v, ry * SEP_TW + x, q);
        }
      }
    }
  }
  // channels beyond C inside the last k-step (C % 16 == 8): zero them once
  if (C & 15) {
    if constexpr (F32A) {
      for (int e = tid; e < 2 * BM; e += 256) *(uint4*)(As + (size_t)(e >> 1) * arow + (C >> 3) * 32 + (e & 1) * 16) = make_uint4(0u, 0u, 0u, 0u);
    } else {
      for (int e = tid; e < NPC * BM; e += 256) *(uint4*)(As + (size_t)e * arow + (C >> 3) * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
  }
  if (ts + 1 == n_samp) split_report<PARTS>(amax, a.oor);
  __syncthreads();

  // ---- 1x1 on the matrix cores: wave w owns pixel rows [32 w, 32 w + 32) and all NT column tiles
  f32x16 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  for (int ks = 0; ks < KS; ++ks) {
    bf16x8 af[NPC];
    if constexpr (F32A) {
      const unsigned char* ap = As + (size_t)(wave * 32 + li) * arow + ks * 64 + lh * 32;     // 8 channels of pixel row li
      const float4 v0 = *(const float4*)ap, v1 = *(const float4*)(ap + 16);
      split_parts<PARTS>(v0, v1, af);
    } else {
#pragma unroll
      for (int p = 0; p < NPC; ++p)
        af[p] = *(const bf16x8*)(As + (size_t)(p * BM + wave * 32 + li) * arow + ks * 32 + lh * 16);
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      bf16x8 bf[NPC];
#pragma unroll
      for (int p = 0; p < NPC; ++p) bf[p] = __builtin_bit_cast(bf16x8, Bs[((ks * NT + n) * NPC + p) * 64 + lane]);
      acc[n] = mfma_terms<PARTS>(af, bf, acc[n]);
    }
  }
  __syncthreads();      // the staging tile below aliases the A image (TIN: the launcher checked that it ends before the B image)

  // tile row m -> output pixel
  auto pixel_of = [&](int m, size_t& pix) -> bool {
    const int y = oy0 + (m >> 4), x = ox0 + (m & 15);
    pix = (size_t)y * a.W + x;
    return y < a.H && x < a.W;
  };
  const size_t out_base = (size_t)bo * a.H * a.W;
  const float un = a.wunscale;             // the packed weights carry a power-of-two factor 1 / un (fp16 pieces; else 1)

  if ((a.Cout & 3) != 0) {
    // Output rows that are not a multiple of 16 bytes (class head: 9 * 7 = 63 channels = 252 bytes per pixel): no aligned
    // float4 stores.  Writing the accumulator layout directly (a lane holds one column of 16 rows: 128-byte pieces at 252-byte
    // pitch) leaves every cache line partially written per store - the counters showed 1.41x the algorithmic traffic for
    // class-predict, an extra read of the output's own size.  The finished values (bias / BN / activation / dropout applied
    // in the accumulator layout, where the column is the lane) are staged per wave as packed rows [32][Cout] and leave in
    // memory order: the 16 pixels of a tile row are one contiguous run of 16 * Cout floats, written 64 lanes x 4 bytes at a time.
    float* st = (float*)slds + wave * 32 * (NT * 32);      // [32 rows][<= NT * 32 columns] packed (the A / B images are dead)
    const int cw = a.Cout - n0 < NT * 32 ? a.Cout - n0 : NT * 32;      // columns of this block
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int col = n0 + n * 32 + li;
      if (col >= a.Cout) continue;
      const float bias = a.bias ? a.bias[col] : 0.f;
      const float sc = a.bn_scale ? a.bn_scale[col] : 1.f;
      const float sh = a.bn_scale ? a.bn_shift[col] : 0.f;
      const float mk = a.mask ? a.mask[(size_t)bo * a.Cout + col] : 1.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = fmaf(fmaf(acc[n][r], un, bias), sc, sh);
        if (a.act == UDA_ACT_SWISH) v = swishf_b(v);
        else if (a.act >= UDA_ACT_RELU) v = act_relu_family(v, a.act);
        v *= mk;
        st[((r & 3) + 8 * (r >> 2) + 4 * lh) * cw + n * 32 + li] = v;
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (cw == a.Cout) {
      // this wave's rows: tile rows 2 wave and 2 wave + 1 (16 pixels each), each a contiguous run of 16 * Cout floats
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int y = oy0 + 2 * wave + half;
        if (y >= a.H) continue;
        const int npx = a.W - ox0 < 16 ? a.W - ox0 : 16;
        float* dst = a.out + (out_base + (size_t)y * a.W + ox0) * a.Cout;
        const float* src = st + half * 16 * cw;
        for (int i = lane; i < npx * cw; i += 64) dst[i] = src[i];
      }
    } else {                                                 // several column blocks: rows are not contiguous per block
      for (int i = lane; i < 32 * cw; i += 64) {
        const int row = i / cw, cc = i - row * cw;
        size_t pix;
        if (pixel_of(wave * 32 + row, pix)) a.out[(out_base + pix) * a.Cout + n0 + cc] = st[i];
      }
    }
    if constexpr (!TIN) return;
    __syncthreads();          // (the next sample's A image is written over the packed rows)
    continue;
  }

  float* stg = (float*)slds + wave * 32 * PWB_STG;
  const int rrow = lane >> 4, c4 = lane & 15;
#pragma unroll
  for (int p = 0; p < (NT + 1) / 2; ++p) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int n = 2 * p + q;
      if (n < NT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * lh) * PWB_STG + q * 32 + li] = acc[n][r];
      }
    }
    __syncthreads();
    const int col = n0 + 2 * p * 32 + 4 * c4;
    const bool colok = (col < a.Cout) && (2 * p * 32 + 4 * c4 < NT * 32);
    // parameters and staged values first (unconditional: dead lanes read a clamped, valid column), the arithmetic, then the
    // eight stores back to back - see pwb_kernel's epilogue: with the loads under the lane's bounds condition every store
    // was preceded by s_waitcnt vmcnt(0), one store in flight per wave
    const int colc = colok ? col : 0;
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f), sc = make_float4(1.f, 1.f, 1.f, 1.f);
    float4 sh = make_float4(0.f, 0.f, 0.f, 0.f), mk = make_float4(1.f, 1.f, 1.f, 1.f);
    if (a.bias) bias = *(const float4*)(a.bias + colc);
    if (a.bn_scale) {
      sc = *(const float4*)(a.bn_scale + colc);
      sh = *(const float4*)(a.bn_shift + colc);
    }
    if (a.mask) mk = *(const float4*)(a.mask + (size_t)bo * a.Cout + colc);
    float4 v[8];
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
      v[it] = t;
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      size_t pix;
      if (pixel_of(wave * 32 + it * 4 + rrow, pix) && colok) *(float4*)(a.out + (out_base + pix) * a.Cout + col) = v[it];
    }
    __syncthreads();
  }
  }       // samples
}

// ---------------------------------------------------------------- separable convolution on an LDS-staged input tile
// sepf_kernel: the same arithmetic as sep_kernel (same depthwise FMA order, same split, same MFMA order: bit-identical
// outputs), organised around an input tile in LDS instead of a depthwise image in LDS, so that the INPUT can be something
// that is computed rather than read - the BiFPN fusion act(sum_i w_i resample_i(in_i)) of the node the separable conv
// belongs to (FIN; efficientdet_keras.py:90-136 + 207-227): the fused tensor never goes to HBM, fuse_kernel's launch, its
// write and this kernel's read of it disappear.
//   tile      16 x 16 outputs per block of four waves (halo 18 x 18: 1.27x instead of 1.41x for 8 x 16), staged PC = 16 or 32
//             channels at a time as [324 pixels][PC + 4] float32 (26 / 47 KB; the k-steps of a pass accumulate into the same
//             accumulators).  16-channel passes (three blocks per CU) for plain and identity + nearest-up inputs; pooled inputs
//             read a 2x larger map, where 64-byte pieces of a pixel doubled the traffic: 32-channel passes (sepf_cfg)
//   staging   U elements (pixel, channel quad) per thread with ALL their loads in flight before the first use: branch-free
//             samplers per mode signature (sepf_stage); a generic one-element-at-a-time loop for anything else
//   depthwise a wave owns four tile rows = two MFMA row tiles; lane (li, lh) computes pixel li of row tile lh, EIGHT channels
//             at a time - the same eight channels on every lane, so the 72 taps of a channel group are wave-uniform (scalar
//             loads, scalar operands of the FMAs: no LDS or vector-memory traffic for the taps); its window arrives in 18
//             ds_read_b128 (pitch PC + 4 dwords: the 16 pixels of a tile row start on 16 different bank quads)
//   fragments two channel groups make one k-step: lane halves trade them with v_permlane32_swap (upper half of group 0 <->
//             lower half of group 1) and each operand register is then the A fragment of row tile 0 / row tile 1 - no LDS
//             round trip of the depthwise result, no second barrier
//   blocks    given to XCDs by sample row (a tile's halo is fetched into ONE L2); epilogue per wave through its own staging
//             rows, loads first, eight stores back to back (see pwb_kernel)
// What bounds it: bytes in flight per CU - requesting the next pass ahead of the current pass's matrix stage was
// built and removed: that stage is an order of magnitude shorter than the memory latency, and the registers cost the third block.
constexpr int SF_T = 16, SF_TP = SF_T + 2, SF_NPX = SF_TP * SF_TP;      // (channels per pass PC, tile pitch PC + 4: template)

__device__ __forceinline__ void swap_halves(bf16x8& a, bf16x8& b) {      // a[32..63] <-> b[0..31], per 32-bit register
  uint4 ua = __builtin_bit_cast(uint4, a), ub = __builtin_bit_cast(uint4, b);
  auto r0 = __builtin_amdgcn_permlane32_swap(ua.x, ub.x, false, false);
  auto r1 = __builtin_amdgcn_permlane32_swap(ua.y, ub.y, false, false);
  auto r2 = __builtin_amdgcn_permlane32_swap(ua.z, ub.z, false, false);
  auto r3 = __builtin_amdgcn_permlane32_swap(ua.w, ub.w, false, false);
  a = __builtin_bit_cast(bf16x8, make_uint4(r0[0], r1[0], r2[0], r3[0]));
  b = __builtin_bit_cast(bf16x8, make_uint4(r0[1], r1[1], r2[1], r3[1]));
}

// One resampled input of the fusion at the (clamped, in-map) output position (y, x), mode known at compile time, loads
// unconditional: out-of-range pool taps are read at a clamped position and replaced by -inf, which is what skipping them
// does to a running maximum (fuse_sample) - same taps in the same order, same bits.
template <int MODE>
__device__ __forceinline__ float4 fuse_sample_t(const FuseArgs& a, int i, int b, int y, int x, int c4) {
  const int bi = b / a.in_div[i];
  const float* base = a.in[i] + (size_t)bi * a.Hi[i] * a.Wi[i] * a.C + c4 * 4;
  if constexpr (MODE == UDA_RS_NONE) {
    return *(const float4*)(base + ((size_t)y * a.Wi[i] + x) * a.C);
  } else if constexpr (MODE == UDA_RS_NEAREST_UP) {
    int sy = (int)floorf((float)y * a.sy[i]);
    int sx = (int)floorf((float)x * a.sx[i]);
    sy = min(sy, a.Hi[i] - 1);
    sx = min(sx, a.Wi[i] - 1);
    return *(const float4*)(base + ((size_t)sy * a.Wi[i] + sx) * a.C);
  } else {                                   // 3 x 3 / stride 2 max pool (the launcher checked pk == 3, ps == 2)
    const int y0 = y * 2 - a.ppt[i], x0 = x * 2 - a.ppl[i];
    float4 t[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = min(max(y0 + ky, 0), a.Hi[i] - 1), ix = min(max(x0 + kx, 0), a.Wi[i] - 1);
        t[ky * 3 + kx] = *(const float4*)(base + ((size_t)iy * a.Wi[i] + ix) * a.C);
      }
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const bool ok = y0 + ky >= 0 && y0 + ky < a.Hi[i] && x0 + kx >= 0 && x0 + kx < a.Wi[i];
        const float4 v = t[ky * 3 + kx];
        m.x = fmaxf(m.x, ok ? v.x : -INFINITY); m.y = fmaxf(m.y, ok ? v.y : -INFINITY);
        m.z = fmaxf(m.z, ok ? v.z : -INFINITY); m.w = fmaxf(m.w, ok ? v.w : -INFINITY);
      }
    return m;
  }
}

// fuse_value (fuse_sample.h) for a compile-time list of modes (M2 < 0: two inputs), swish on
template <int M0, int M1, int M2>
__device__ __forceinline__ float4 fuse_value_t(const FuseArgs& a, int b, int y, int x, int c4) {
  float4 s = fuse_sample_t<M0>(a, 0, b, y, x, c4);
  s.x *= a.wgt[0]; s.y *= a.wgt[0]; s.z *= a.wgt[0]; s.w *= a.wgt[0];
  {
    const float4 v = fuse_sample_t<M1>(a, 1, b, y, x, c4);
    s.x = fmaf(v.x, a.wgt[1], s.x); s.y = fmaf(v.y, a.wgt[1], s.y);
    s.z = fmaf(v.z, a.wgt[1], s.z); s.w = fmaf(v.w, a.wgt[1], s.w);
  }
  if constexpr (M2 >= 0) {
    const float4 v = fuse_sample_t<M2>(a, 2, b, y, x, c4);
    s.x = fmaf(v.x, a.wgt[2], s.x); s.y = fmaf(v.y, a.wgt[2], s.y);
    s.z = fmaf(v.z, a.wgt[2], s.z); s.w = fmaf(v.w, a.wgt[2], s.w);
  }
  s.x = fuse_swish(s.x); s.y = fuse_swish(s.y); s.z = fuse_swish(s.z); s.w = fuse_swish(s.w);
  return s;
}

// mode signatures with a batched staging path (everything else: the generic, one-element-at-a-time loop)
enum { SF_SIG_GENERIC = 0, SF_SIG_PLAIN = 1, SF_SIG_NU = 2, SF_SIG_NNP = 3, SF_SIG_NP = 4 };

// Stage one 32-channel pass of the 18 x 18 input tile: U elements (pixel, channel quad) per thread and round with ALL their
// loads in flight before the first use - the block has one memory latency per round, not one per element.
template <int SIG, int U, int PC>
__device__ __forceinline__ void sepf_stage(const SepArgs& a, const FuseArgs& f, float* T, const float* inb, int b, int oy0, int ox0,
                                           int c0, int pc4, int tid) {
  constexpr int SF_PITCH = PC + 4, FULL4 = PC / 4;
  const int H = a.H, W = a.W, C = a.C;
  const int ne = SF_NPX * pc4;
  for (int e0 = tid; e0 < ne; e0 += 256 * U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + 256 * u < ne ? e0 + 256 * u : tid;       // (a dead slot repeats a live element: valid addresses)
      const int px = pc4 == FULL4 ? e / FULL4 : e / pc4, q = e - px * pc4;
      const int sy = px / SF_TP, sx = px - sy * SF_TP;
      const int y = oy0 - 1 + sy, x = ox0 - 1 + sx;
      const bool inside = y >= 0 && y < H && x >= 0 && x < W;
      const int yc = min(max(y, 0), H - 1), xc = min(max(x, 0), W - 1);
      const int c4 = (c0 >> 2) + q;
      float4 t;
      if constexpr (SIG == SF_SIG_PLAIN) t = *(const float4*)(inb + ((size_t)yc * W + xc) * C + 4 * c4);
      else if constexpr (SIG == SF_SIG_NU) t = fuse_value_t<UDA_RS_NONE, UDA_RS_NEAREST_UP, -1>(f, b, yc, xc, c4);
      else if constexpr (SIG == SF_SIG_NNP) t = fuse_value_t<UDA_RS_NONE, UDA_RS_NONE, UDA_RS_MAXPOOL>(f, b, yc, xc, c4);
      else t = fuse_value_t<UDA_RS_NONE, UDA_RS_MAXPOOL, -1>(f, b, yc, xc, c4);
      v[u] = inside ? t : make_float4(0.f, 0.f, 0.f, 0.f);     // SAME padding of the depthwise conv: zeros outside the map
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + 256 * u;
      if (e < ne) {
        const int px = pc4 == FULL4 ? e / FULL4 : e / pc4, q = e - px * pc4;
        *(float4*)(T + px * SF_PITCH + 4 * q) = v[u];
      }
    }
  }
}

#ifdef UDA_SEPF_STAMPS
// diagnostic build (-DUDA_SEPF_STAMPS, A/B libraries only): shader-clock cycles every wave of the LARGE launches (maps of at
// least 96 x 160) spends in the phases of a block: [0] weight fragments + staging of a pass (requests, wait, fusion, LDS
// writes), [1] barrier behind it, [2] depthwise + matrix stage, [3] barrier in front of the next pass, [4] epilogue + stores,
// [5] everything up to the first pass, [6] blocks
__device__ unsigned long long g_sepf_stamps[64 * 8];      // 64 copies (block id mod 64): the atomics of a launch do not queue on one line
static void sepf_stamp_dump() {
  unsigned long long h[8] = {}, all[64 * 8] = {};
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(all, HIP_SYMBOL(g_sepf_stamps), sizeof(all)) != hipSuccess) return;
  for (int i = 0; i < 64 * 8; ++i) h[i & 7] += all[i];
  if (!h[6]) return;
  const char* nm[6] = {"staging", "barrier after staging", "depthwise + matrix", "barrier before staging", "epilogue", "prologue"};
  unsigned long long tot = 0;
  for (int i = 0; i < 6; ++i) tot += h[i];
  fprintf(stderr, "[sepf stamps] %llu waves, %.0f cycles = %.2f us (100 MHz wall clock) each:", h[6], (double)tot / (double)h[6], (double)h[7] / (double)h[6] / 100.0);
  for (int i = 0; i < 6; ++i) fprintf(stderr, "  %s %.0f (%.0f %%)", nm[i], (double)h[i] / (double)h[6], 100.0 * (double)h[i] / (double)tot);
  fprintf(stderr, "\n");
}
#define SEPF_STAMP(i) do { const unsigned long long t_ = clock64(); st_acc[i] += t_ - st_t; st_t = t_; } while (0)
#else
#define SEPF_STAMP(i) do { } while (0)
#endif

// PC = channels per pass: 16 (three blocks per CU) | 32 (two).  Three and four column tiles (88 / 112 / 128 output channels:
// D1 ... D3) hold 96 / 128 accumulator registers per lane: at the 168 registers of three blocks per CU the compiler spilled
// 50-150 of them and the scratch traffic was 3.4-4.5x the kernel's algorithmic bytes (profiles/r05_d2_per_op.txt, round 5:
// the D2 BiFPN nodes at 1.2 TB/s algorithmic with 5.4 TB/s on the counters) - those shapes take two blocks per CU.
template <int NT, int PARTS, bool FIN, int PC>
__global__ __launch_bounds__(256, (PC == 16 && NT <= 2) ? 3 : 2) void sepf_kernel(SepArgs a, FuseArgs f, int sig, int rows) {
  extern __shared__ __attribute__((aligned(16))) unsigned char slds[];
  constexpr int NPC = split_np(PARTS);
  constexpr int SF_PC = PC, SF_PITCH = PC + 4;
  float* T = (float*)slds;                                              // [324][PC + 4] staged input tile, one channel pass
  uint4* Bs = (uint4*)(slds + (size_t)SF_NPX * SF_PITCH * 4);           // [PC / 16 k-steps][NT][NPC][64 lanes] x 16 B
  const int C = a.C, H = a.H, W = a.W;
  const int NTL = (a.Cout + 31) >> 5;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int tiles_x = (W + SF_T - 1) / SF_T;
  // Block -> (sample row b, tile): consecutive block ids go round the 8 XCDs (each with its own L2), so the tiles of one
  // sample row are given to ONE XCD - sample rows b = xcd (mod 8), tiles in order - and the halo a tile shares with its
  // neighbours is fetched into that L2 once (rows beyond the last multiple of 8: plain order).
  int b, tile;
  {
    const int ntile = tiles_x * ((H + SF_T - 1) / SF_T);
    const int L = blockIdx.x, rows8 = rows & ~7;
    if (L < ntile * rows8) {
      const int xcd = L & 7, j = L >> 3;
      b = (j / ntile) * 8 + xcd;
      tile = j % ntile;
    } else {
      const int l2 = L - ntile * rows8;
      b = rows8 + l2 / ntile;
      tile = l2 % ntile;
    }
  }
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int oy0 = ty * SF_T, ox0 = tx * SF_T;
  // this lane's output pixel inside the tile = top-left corner of its 3 x 3 window in staged coordinates
  const int pr = 4 * wave + 2 * lh + (li >> 4), pcx = li & 15;
  const float* tbase = T + (pr * SF_TP + pcx) * SF_PITCH;
  const uint4* Wp = (const uint4*)a.wsplit;
  const float* inb = FIN ? nullptr : a.in + (size_t)(b / a.in_div) * H * W * C;

  f32x16 acc[2][NT];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][n][r] = 0.f;
  float amax = 0.f;

#ifdef UDA_SEPF_STAMPS
  unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_t = clock64();
  const unsigned long long st_w0 = wall_clock64();
#endif
  const int npass = (C + SF_PC - 1) / SF_PC;
  for (int ps = 0; ps < npass; ++ps) {
    const int c0 = ps * SF_PC;
    const int pch = C - c0 < SF_PC ? C - c0 : SF_PC;      // channels of this pass (a multiple of 8)
    const int pc4 = pch >> 2;
    const int nks = (pch + 15) >> 4;
    if (ps) SEPF_STAMP(2); else SEPF_STAMP(5);
    if (ps) __syncthreads();                              // every wave is done with the previous pass's tile and fragments
    SEPF_STAMP(3);
    for (int i = tid; i < nks * NT * NPC * 64; i += 256) {
      int q = i >> 6;
      const int part = q % NPC; q /= NPC;
      const int nt = q % NT, ksl = q / NT;
      Bs[i] = nt < NTL ? Wp[(((size_t)(ps * (PC / 16) + ksl) * NTL + nt) * NPC + part) * 64 + (i & 63)] : make_uint4(0u, 0u, 0u, 0u);
    }
    if constexpr (!FIN) {
      sepf_stage<SF_SIG_PLAIN, PC == 16 ? 6 : 11, PC>(a, f, T, inb, b, oy0, ox0, c0, pc4, tid);
    } else if (sig == SF_SIG_NU) {
      // (identity + nearest-up nodes run 16-channel passes, sepf_pc: the 32-channel instance of this branch only exists for
      // UDA_SEPF_PC=32 A/B runs and must not set the register count of the kernel the pooled nodes use - 11 elements of two
      // inputs in flight were 88 registers, the pooled nodes' kernel spilled 50-115 of them)
      sepf_stage<SF_SIG_NU, PC == 16 ? 6 : 2, PC>(a, f, T, inb, b, oy0, ox0, c0, pc4, tid);
    } else if (sig == SF_SIG_NNP) {
      sepf_stage<SF_SIG_NNP, (PC == 16 || NT > 2) ? 1 : 3, PC>(a, f, T, inb, b, oy0, ox0, c0, pc4, tid);      // (11 loads per element: one element in flight beside 96+ accumulators)
    } else if (sig == SF_SIG_NP) {
      sepf_stage<SF_SIG_NP, (PC == 16 || NT > 2) ? 1 : 3, PC>(a, f, T, inb, b, oy0, ox0, c0, pc4, tid);
    } else {
      for (int e = tid; e < SF_NPX * pc4; e += 256) {
        const int px = e / pc4, q = e - px * pc4;
        const int sy = px / SF_TP, sx = px - sy * SF_TP;
        const int y = oy0 - 1 + sy, x = ox0 - 1 + sx;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (y >= 0 && y < H && x >= 0 && x < W) v = fuse_value(f, b, y, x, (c0 >> 2) + q);
        *(float4*)(T + px * SF_PITCH + 4 * q) = v;
      }
    }
    SEPF_STAMP(0);
    __syncthreads();
    SEPF_STAMP(1);

    for (int ksl = 0; ksl < nks; ++ksl) {
      bf16x8 fr[2][NPC];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int cl = ksl * 16 + g * 8;                  // channel offset inside the pass (uniform)
        float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0;
        if (cl < pch) {
          const float* wt = a.wd + c0 + cl;               // uniform address: scalar loads
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
              const float* p = tbase + (ky * SF_TP + kx) * SF_PITCH + cl;
              const float4 v0 = *(const float4*)p, v1 = *(const float4*)(p + 4);
              const float* w = wt + (size_t)(ky * 3 + kx) * C;
              r0.x = fmaf(v0.x, w[0], r0.x); r0.y = fmaf(v0.y, w[1], r0.y);
              r0.z = fmaf(v0.z, w[2], r0.z); r0.w = fmaf(v0.w, w[3], r0.w);
              r1.x = fmaf(v1.x, w[4], r1.x); r1.y = fmaf(v1.y, w[5], r1.y);
              r1.z = fmaf(v1.z, w[6], r1.z); r1.w = fmaf(v1.w, w[7], r1.w);
            }
        }
        split_parts<PARTS>(r0, r1, fr[g], amax);
      }
#pragma unroll
      for (int p = 0; p < NPC; ++p) swap_halves(fr[0][p], fr[1][p]);    // fr[t] = A fragment of row tile t
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        bf16x8 bf[NPC];
#pragma unroll
        for (int p = 0; p < NPC; ++p) bf[p] = __builtin_bit_cast(bf16x8, Bs[((ksl * NT + n) * NPC + p) * 64 + lane]);
        acc[0][n] = mfma_terms<PARTS>(fr[0], bf, acc[0][n]);
        acc[1][n] = mfma_terms<PARTS>(fr[1], bf, acc[1][n]);
      }
    }
  }
  split_report<PARTS>(amax, a.oor);
  SEPF_STAMP(2);
  __syncthreads();      // the staging rows below alias the tile
  SEPF_STAMP(3);

  // epilogue: per wave, 32 x 64 values at a time through its own staging rows, out as float4 along the channels
  float* stg = (float*)slds + wave * 32 * PWB_STG;
  const size_t out_base = (size_t)b * H * W;
  const float un = a.wunscale;
  const int rrow = lane >> 4, c4 = lane & 15;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int p = 0; p < (NT + 1) / 2; ++p) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int n = 2 * p + q;
        if (n < NT) {
#pragma unroll
          for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * lh) * PWB_STG + q * 32 + li] = acc[t][n][r];
        }
      }
      __builtin_amdgcn_wave_barrier();
      const int col = 2 * p * 32 + 4 * c4;
      const bool colok = col < a.Cout && col < NT * 32;
      const int colc = colok ? col : 0;            // (loads unconditional, stores back to back: see pwb_kernel's epilogue)
      float4 bias = make_float4(0.f, 0.f, 0.f, 0.f), sc = make_float4(1.f, 1.f, 1.f, 1.f);
      float4 sh = make_float4(0.f, 0.f, 0.f, 0.f), mk = make_float4(1.f, 1.f, 1.f, 1.f);
      if (a.bias) bias = *(const float4*)(a.bias + colc);
      if (a.bn_scale) {
        sc = *(const float4*)(a.bn_scale + colc);
        sh = *(const float4*)(a.bn_shift + colc);
      }
      if (a.mask) mk = *(const float4*)(a.mask + (size_t)b * a.Cout + colc);
      float4 v[8];
#pragma unroll
      for (int it = 0; it < 8; ++it) v[it] = *(const float4*)(stg + (it * 4 + rrow) * PWB_STG + 4 * c4);
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        float4 w = v[it];
        w.x = fmaf(fmaf(w.x, un, bias.x), sc.x, sh.x);
        w.y = fmaf(fmaf(w.y, un, bias.y), sc.y, sh.y);
        w.z = fmaf(fmaf(w.z, un, bias.z), sc.z, sh.z);
        w.w = fmaf(fmaf(w.w, un, bias.w), sc.w, sh.w);
        if (a.act == UDA_ACT_SWISH) {
          w.x = swishf_b(w.x); w.y = swishf_b(w.y); w.z = swishf_b(w.z); w.w = swishf_b(w.w);
        } else if (a.act >= UDA_ACT_RELU) {
          w.x = act_relu_family(w.x, a.act); w.y = act_relu_family(w.y, a.act); w.z = act_relu_family(w.z, a.act); w.w = act_relu_family(w.w, a.act);
        }
        w.x *= mk.x; w.y *= mk.y; w.z *= mk.z; w.w *= mk.w;
        v[it] = w;
      }
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = it * 4 + rrow;                                  // row of the 32-pixel row tile
        const int y = oy0 + 4 * wave + 2 * t + (row >> 4), x = ox0 + (row & 15);
        if (colok && y < H && x < W) *(float4*)(a.out + (out_base + (size_t)y * W + x) * a.Cout + col) = v[it];
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
#ifdef UDA_SEPF_STAMPS
  SEPF_STAMP(4);
  if (tid == 0 && a.H * a.W >= 96 * 160) {      // wave 0 speaks for the block
    unsigned long long* g = g_sepf_stamps + (blockIdx.x & 63) * 8;
    for (int i = 0; i < 6; ++i) atomicAdd(&g[i], st_acc[i]);
    atomicAdd(&g[6], 1ull);
    atomicAdd(&g[7], wall_clock64() - st_w0);
  }
#endif
}

// channels per pass per staging signature: plain and identity + nearest-up inputs run 16-channel passes; pooled inputs read a
// 2x larger map, where 64-byte pieces of a pixel doubled the traffic (half-used 128-byte lines): 32-channel passes.
// UDA_SEPF_PC = 16 | 32 forces one (A/B).
static int sepf_pc(int sig) {
  static int epc = -1;
  if (epc < 0) { const char* e = getenv("UDA_SEPF_PC"); epc = e ? atoi(e) : 0; }
  return epc == 16 || epc == 32 ? epc : ((sig == SF_SIG_PLAIN || sig == SF_SIG_NU) ? 16 : 32);
}

size_t sepf_lds_bytes(int C, int Cout, int scheme) {      // (the larger of the configurations)
  const int nt = (Cout + 31) / 32, npc = uda_split_pieces(scheme);
  size_t lds = (size_t)SF_NPX * 36 * 4 + (size_t)2 * nt * npc * 1024;
  const size_t stg = (size_t)4 * 32 * PWB_STG * 4;
  return lds < stg ? stg : lds;
}

bool sepf_supported(int C, int Cout, int scheme) {
  return scheme != UDA_SPLIT_NONE && C % 8 == 0 && C >= 16 && C <= 128 && Cout % 4 == 0 && Cout >= 4 && Cout <= 128 &&
         sepf_lds_bytes(C, Cout, scheme) <= 80 * 1024;
}

template <int NT, bool FIN>
static void launch_sepf_nt(const SepArgs& a, const FuseArgs& f, int rows, hipStream_t s) {
  const dim3 grid(((a.W + SF_T - 1) / SF_T) * ((a.H + SF_T - 1) / SF_T) * rows);
  // which batched staging path fits the fusion (uniform switch inside the kernel)
  int sig = SF_SIG_GENERIC;
  if (!FIN) sig = SF_SIG_PLAIN;
  else if (f.act == UDA_ACT_SWISH && !getenv("UDA_SEPF_GENERIC")) {
    auto pool_ok = [&](int i) { return f.mode[i] == UDA_RS_MAXPOOL && f.pk[i] == 3 && f.ps[i] == 2; };
    if (f.n_in == 2 && f.mode[0] == UDA_RS_NONE && f.mode[1] == UDA_RS_NEAREST_UP) sig = SF_SIG_NU;
    else if (f.n_in == 3 && f.mode[0] == UDA_RS_NONE && f.mode[1] == UDA_RS_NONE && pool_ok(2)) sig = SF_SIG_NNP;
    else if (f.n_in == 2 && f.mode[0] == UDA_RS_NONE && pool_ok(1)) sig = SF_SIG_NP;
  }
  const int pc = sepf_pc(sig);
  const int npc = uda_split_pieces(a.wparts);
  size_t lds = (size_t)SF_NPX * (pc + 4) * 4 + (size_t)(pc / 16) * NT * npc * 1024;
  const size_t stg = (size_t)4 * 32 * PWB_STG * 4;
  if (lds < stg) lds = stg;
  auto go = [&](auto kern) {
    static size_t attr_lds = 64 * 1024;
    if (lds > attr_lds) {
      hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_lds = lds;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a, f, sig, rows);
#ifdef UDA_SEPF_STAMPS
    static int n_launch = 0;
    if (a.H * a.W >= 96 * 160 && ++n_launch % 9 == 0) sepf_stamp_dump();
#endif
  };
  auto by_scheme = [&](auto pcv) {
    constexpr int PC = decltype(pcv)::value;
    if (a.wparts == UDA_SPLIT_BF16X3) go(sepf_kernel<NT, 3, FIN, PC>);
    else if (a.wparts == UDA_SPLIT_F16X2) go(sepf_kernel<NT, 4, FIN, PC>);
    else go(sepf_kernel<NT, 2, FIN, PC>);
  };
  if (pc == 16) by_scheme(std::integral_constant<int, 16>{});
  else by_scheme(std::integral_constant<int, 32>{});
}

// fused != nullptr: the conv's input is the BiFPN fusion described by *fused (a.in unused)
void launch_sepf(const SepArgs& a, const FuseArgs* fused, int rows, hipStream_t s) {
  const int ntl = (a.Cout + 31) / 32;
  FuseArgs f{};
  if (fused) f = *fused;
  auto pick = [&](auto fin) {
    constexpr bool FIN = decltype(fin)::value;
    if (ntl == 1) launch_sepf_nt<1, FIN>(a, f, rows, s);
    else if (ntl == 2) launch_sepf_nt<2, FIN>(a, f, rows, s);
    else if (ntl == 3) launch_sepf_nt<3, FIN>(a, f, rows, s);
    else launch_sepf_nt<4, FIN>(a, f, rows, s);
  };
  if (fused) pick(std::true_type{}); else pick(std::false_type{});
}

bool sep_supported(int C, int Cout) { return C % 8 == 0 && C >= 16 && C <= 128 && Cout >= 1; }

// Deferred-input mode (sep_kernel TIN): the per-sample epilogue staging must end before the weight image (it may only alias the
// A image, which is rebuilt per sample), all columns in one block, two to four column tiles.
bool sep_tin_supported(int C, int Cout, int scheme) {
  if (scheme == UDA_SPLIT_NONE || !sep_supported(C, Cout) || C < 64) return false;
  const int KS = (C + 15) / 16, ntl = (Cout + 31) / 32, npc = uda_split_pieces(scheme);
  if (ntl < 2 || ntl > 4) return false;
  const size_t a_img = scheme == UDA_SPLIT_BF16X3 ? (size_t)128 * (KS * 64 + 16) : (size_t)npc * 128 * (KS * 32 + 16);
  const size_t stg = (Cout & 3) ? (size_t)4 * 32 * ntl * 32 * 4 : (size_t)4 * 32 * PWB_STG * 4;
  if (stg > a_img) return false;
  return a_img + (size_t)KS * ntl * npc * 1024 <= 120 * 1024;      // (the launcher would split the columns over several blocks otherwise)
}

static int sep_tiles(int H, int W) { return ((W + SEP_TW - 1) / SEP_TW) * ((H + SEP_TH - 1) / SEP_TH); }

template <int NT>
static void launch_sep_nt(const SepMulti& m, int rows, int gy, hipStream_t s) {
  const SepArgs& a = m.one;
  const int KS = (a.C + 15) / 16;
  const int npc = uda_split_pieces(a.wparts);
  const size_t a_img = a.wparts == UDA_SPLIT_BF16X3 ? (size_t)128 * (KS * 64 + 16) : (size_t)npc * 128 * (KS * 32 + 16);   // float32 image | pieces
  size_t lds = a_img + (size_t)KS * NT * npc * 1024;
  const size_t stg = (a.Cout & 3) ? (size_t)4 * 32 * NT * 32 * 4 : (size_t)4 * 32 * PWB_STG * 4;     // epilogue staging (packed rows | float4 tiles)
  if (lds < stg) lds = stg;
  SepMulti mm = m;
  mm.tiles = m.n_lv > 0 ? m.tile0[m.n_lv] : sep_tiles(a.H, a.W);
  static const bool remap = !(getenv("UDA_SEP_REMAP") && atoi(getenv("UDA_SEP_REMAP")) == 0);     // (0: plain block order, A/B)
  // deferred-input mode: one block serves the in_div samples of an image - the grid runs over images
  const int tin = a.mask_in ? (a.C <= 64 ? 1 : 2) : 0;
  if (tin) rows /= a.in_div;
  mm.rows = remap ? rows : 0;
  const dim3 grid(mm.tiles * rows, gy);
  auto go = [&](auto kern) {
    static size_t attr_lds = 64 * 1024;
    if (lds > attr_lds) {
      hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_lds = lds;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, mm);
  };
  static int occ = -1;
  if (occ < 0) { const char* e = getenv("UDA_SEP_OCC"); occ = e ? atoi(e) : 3; }   // 3 blocks per CU: measured 14 % faster than 2
  if (tin) {
    // (head layers: C = F, Cout = F | 9 C | 36 / 72: two to four column tiles; 32 / 64 more registers than the plain kernel)
    if constexpr (NT >= 2) {
      if (tin == 1) {
        // (two blocks per CU: the 32 registers of the kept depthwise result spill at the 168 of three)
        if (a.wparts == UDA_SPLIT_BF16X3) go(sep_kernel<NT, 3, 2, 1>);
        else if (a.wparts == UDA_SPLIT_F16X2) go(sep_kernel<NT, 4, 2, 1>);
        else go(sep_kernel<NT, 2, 2, 1>);
      } else {
        if (a.wparts == UDA_SPLIT_BF16X3) go(sep_kernel<NT, 3, 2, 2>);
        else if (a.wparts == UDA_SPLIT_F16X2) go(sep_kernel<NT, 4, 2, 2>);
        else go(sep_kernel<NT, 2, 2, 2>);
      }
    }
    return;
  }
  if (a.wparts == UDA_SPLIT_BF16X3) { if (occ >= 3 && NT <= 2) go(sep_kernel<NT, 3, 3>); else go(sep_kernel<NT, 3, 2>); }
  else if (a.wparts == UDA_SPLIT_F16X2) { if (occ >= 3 && NT <= 2) go(sep_kernel<NT, 4, 3>); else go(sep_kernel<NT, 4, 2>); }
  else if (occ >= 3 && NT <= 2) go(sep_kernel<NT, 2, 3>);
  else go(sep_kernel<NT, 2, 2>);
}

// Dynamic LDS of the launch launch_sep_any will make for a C -> Cout separable conv (see launch_sep_nt): checked by uda_create
size_t sep_lds_bytes(int C, int Cout, int scheme) {
  const int KS = (C + 15) / 16, ntl = (Cout + 31) / 32, npc = uda_split_pieces(scheme);
  const size_t a_img = scheme == UDA_SPLIT_BF16X3 ? (size_t)128 * (KS * 64 + 16) : (size_t)npc * 128 * (KS * 32 + 16);
  int nt = ntl <= 4 ? ntl : 3;
  if (ntl > 2 && a_img + (size_t)KS * (ntl < 4 ? ntl : 3) * npc * 1024 > 120 * 1024) nt = 2;
  const size_t lds = a_img + (size_t)KS * nt * npc * 1024;
  const size_t stg = (Cout & 3) ? (size_t)4 * 32 * nt * 32 * 4 : (size_t)4 * 32 * PWB_STG * 4;
  return lds < stg ? stg : lds;
}

static void launch_sep_any(const SepMulti& m, int rows, hipStream_t s) {
  const int ntl = (m.one.Cout + 31) / 32;
  // three pieces per operand (UDA_PW_TERMS=6) at 112 channels (D2's BiFPN): the A image (128 x 464 B) plus the weight
  // fragments of four column tiles (7 x 4 x 3 KB) are 142 KB - two column tiles per block there, more column blocks
  {
    const int KS = (m.one.C + 15) / 16;
    const int npc = uda_split_pieces(m.one.wparts);
    const size_t a_img = m.one.wparts == UDA_SPLIT_BF16X3 ? (size_t)128 * (KS * 64 + 16) : (size_t)npc * 128 * (KS * 32 + 16);
    const size_t need = a_img + (size_t)KS * (ntl < 4 ? ntl : 3) * npc * 1024;
    if (ntl > 2 && need > 120 * 1024) { launch_sep_nt<2>(m, rows, (ntl + 1) / 2, s); return; }
  }
  // all columns in one block when they fit four 32-column tiles, else blocks of three
  if (ntl == 1) launch_sep_nt<1>(m, rows, 1, s);
  else if (ntl == 2) launch_sep_nt<2>(m, rows, 1, s);
  else if (ntl == 3) launch_sep_nt<3>(m, rows, 1, s);
  else if (ntl == 4) launch_sep_nt<4>(m, rows, 1, s);
  else launch_sep_nt<3>(m, rows, (ntl + 2) / 3, s);
}

void launch_sep(const SepArgs& a, int rows, hipStream_t s) {
  SepMulti m{};
  m.one = a;
  m.n_lv = 0;
  launch_sep_any(m, rows, s);
}

void launch_sep_multi(const SepArgs& common, const SepLevel* lv, int n_lv, int rows, hipStream_t s) {
  SepMulti m{};
  m.one = common;
  m.n_lv = n_lv;
  int t = 0;
  for (int i = 0; i < n_lv; ++i) {
    m.lv[i] = lv[i];
    m.tile0[i] = t;
    t += sep_tiles(lv[i].H, lv[i].W);
  }
  m.tile0[n_lv] = t;
  launch_sep_any(m, rows, s);
}

}  // namespace uda
