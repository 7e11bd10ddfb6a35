// Fused separable convolution on the bf16 matrix cores (split-precision products as in kernels_pwb.hip).
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "mfma_common.h"

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

template <int NT, int PARTS, int OCC>     // NT = 32-column tiles of the 1x1 output handled per block (each wave: all of them)
__global__ __launch_bounds__(256, OCC) void sep_kernel(SepMulti m) {
  constexpr int BM = SEP_TH * SEP_TW;      // 128 pixels = 4 MFMA row tiles, one per wave
  // the problem of this block (uniform): one conv, or one of the pyramid levels of a head layer launched together
  SepArgs a = m.one;
  int bx = blockIdx.x;
  if (m.n_lv > 0) {
    int l = 0;
    while (l + 1 < m.n_lv && bx >= m.tile0[l + 1]) ++l;
    bx -= m.tile0[l];
    const SepLevel& L = m.lv[l];
    a.in = L.in; a.out = L.out; a.wd = L.wd; a.wsplit = L.wsplit;
    a.bias = L.bias; a.bn_scale = L.bn_scale; a.bn_shift = L.bn_shift; a.mask = L.mask;
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
  const int b = blockIdx.z, b_in = b / a.in_div;
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
  for (int u = tid; u < SEP_TW * C4; u += 256) {
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
      // split into bf16 pieces -> A image row m = ry * 16 + x, channels 4q .. 4q + 3
      const int m = ry * SEP_TW + x;
      if constexpr (F32A) {
        *(float4*)(As + (size_t)m * arow + q * 16) = acc;
        continue;
      }
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
  }
  // channels beyond C inside the last k-step (C % 16 == 8): zero them once
  if (C & 15) {
    if constexpr (F32A) {
      for (int e = tid; e < 2 * BM; e += 256) *(uint4*)(As + (size_t)(e >> 1) * arow + (C >> 3) * 32 + (e & 1) * 16) = make_uint4(0u, 0u, 0u, 0u);
    } else {
      for (int e = tid; e < NPC * BM; e += 256) *(uint4*)(As + (size_t)e * arow + (C >> 3) * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
  }
  split_report<PARTS>(amax, a.oor);
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
  __syncthreads();      // the staging tile below aliases the A / B images

  // tile row m -> output pixel
  auto pixel_of = [&](int m, size_t& pix) -> bool {
    const int y = oy0 + (m >> 4), x = ox0 + (m & 15);
    pix = (size_t)y * a.W + x;
    return y < a.H && x < a.W;
  };
  const size_t out_base = (size_t)b * a.H * a.W;
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
      const float mk = a.mask ? a.mask[(size_t)b * a.Cout + col] : 1.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = fmaf(fmaf(acc[n][r], un, bias), sc, sh);
        if (a.act == UDA_ACT_SWISH) v = swishf_b(v);
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
    return;
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
      size_t pix;
      if (colok && pixel_of(wave * 32 + row, pix)) {
        float4 v = *(const float4*)(stg + row * PWB_STG + 4 * c4);
        v.x = fmaf(fmaf(v.x, un, bias.x), sc.x, sh.x);
        v.y = fmaf(fmaf(v.y, un, bias.y), sc.y, sh.y);
        v.z = fmaf(fmaf(v.z, un, bias.z), sc.z, sh.z);
        v.w = fmaf(fmaf(v.w, un, bias.w), sc.w, sh.w);
        if (a.act == UDA_ACT_SWISH) {
          v.x = swishf_b(v.x); v.y = swishf_b(v.y); v.z = swishf_b(v.z); v.w = swishf_b(v.w);
        }
        v.x *= mk.x; v.y *= mk.y; v.z *= mk.z; v.w *= mk.w;
        *(float4*)(a.out + (out_base + pix) * a.Cout + col) = v;
      }
    }
    __syncthreads();
  }
}

bool sep_supported(int C, int Cout) { return C % 8 == 0 && C >= 16 && C <= 128 && Cout >= 1; }

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
  const dim3 grid(m.n_lv > 0 ? m.tile0[m.n_lv] : sep_tiles(a.H, a.W), gy, rows);
  auto go = [&](auto kern) {
    static size_t attr_lds = 64 * 1024;
    if (lds > attr_lds) {
      hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_lds = lds;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, m);
  };
  static int occ = -1;
  if (occ < 0) { const char* e = getenv("UDA_SEP_OCC"); occ = e ? atoi(e) : 3; }   // 3 blocks per CU: measured 14 % faster than 2
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
