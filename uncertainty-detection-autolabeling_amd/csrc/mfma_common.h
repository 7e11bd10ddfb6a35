// Device helpers shared by the split-bf16 MFMA kernels (kernels_pwb.hip: 1x1 convolutions and the fused MBConv front
// halves; kernels_sep.hip: the fused separable convolutions).  The two files are separate translation units because
// they want different compiler settings: see the Makefile (EXTRA_kernels_pwb).
#pragma once
#include "uda_internal.h"

namespace uda {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float sigmoidf_b(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float swishf_b(float x) { return x * sigmoidf_b(x); }

// swish with the exponent scale folded into the producer: y = -log2(e) * x comes out of the GEMM / BN (weights,
// shift and BN scale are pre-multiplied on the host or when they are staged), k = -keep_scale / log2(e):
//   x * sigmoid(x) * keep_scale = y * k / (1 + 2^y)        (v_exp, v_add, v_rcp, 2 v_mul: 5 VALU instead of 7)
constexpr float UDA_NEG_LN2 = -0.6931471805599453f;
__device__ __forceinline__ float swish_folded(float y, float k) {
  return (y * k) * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y));
}

// the same without the constant factor: y / (1 + 2^y) = swish(x) / (-ln 2) for y = -log2(e) x.  The fused MBConv kernels
// store THIS as the expanded activation and fold (-ln 2) x (dropout keep-scale of the channel) into the BN scale that
// follows the depthwise convolution (a per-channel factor commutes with a depthwise convolution): one multiply less per
// expanded element.
__device__ __forceinline__ float swish_core(float y) { return y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y)); }

// Store through a UNIFORM base (scalar register pair) plus a 32-bit per-lane byte offset: global_store_dword v, v, s[..].
// The base is made opaque so that the address is not re-associated into per-lane 64-bit pointers (which the compiler
// then hoists out of the slab loop: 2 registers per output, spilled at the occupancy these kernels need).
typedef __attribute__((address_space(1))) char uda_gchar;
typedef __attribute__((address_space(1))) float uda_gfloat;
__device__ __forceinline__ void store_uniform_base(float* base, unsigned byte_off, float v) {
  // (written as one instruction: left to the compiler the scalar base is copied into a register pair per lane and the
  // 64-bit add comes back.  vmcnt stays conservative: an outstanding store the compiler does not know of only makes a
  // later counted wait cover more operations, never fewer - memory operations retire in order.)
  uda_gchar* g = (uda_gchar*)base;
  asm volatile("global_store_dword %0, %1, %2" : : "v"(byte_off), "v"(v), "s"(g) : "memory");
}

// two floats -> packed bf16 pair (round to nearest even; v_cvt_pk_bf16_f32), element 0 in the low half
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf16_lo_f32(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf16_hi_f32(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

// ---------------------------------------------------------------- split schemes
// The PARTS template parameter of the split-precision kernels (= the `wparts` field of their argument blocks) names the
// scheme (uda_internal.h: UDA_SPLIT_*): 2 = two bf16 pieces / three cross terms, 3 = three bf16 pieces / six cross terms,
// 4 = two FP16 pieces / three cross terms on v_mfma_f32_32x32x16_f16.  fp16 carries 11 significant bits per piece, so two
// pieces hold 22 bits and a0 b0 + a0 b1 + a1 b0 is good to ~2^-22 per product (the six bf16 terms: ~2^-24; float32 itself
// accumulates with 2^-24 per addition) at the matrix-core cost of the three-term bf16 scheme.  The price is fp16's
// exponent range: an operand above 65504 overflows (tracked per lane and reported through the launch's range flag, see
// split_track / split_report - never silently), one below 2^-3 has a subnormal low piece, i.e. an ABSOLUTE resolution of
// 2^-25 instead of a relative one of 2^-22 (the matrix cores honour fp16 subnormals: tools/micro/f16_split_probe.hip).
// Weights are pre-scaled by a power of two on the host where the accumulator has a free place to undo it (1x1 / separable
// convolutions: PwArgs::wunscale), so their low pieces are always normal numbers; the depthwise result that feeds a separable
// conv's 1x1 is scaled up the same way (pre-scaled taps, uda_api.hip).  Measured on the heads of D0 / D2 against the float32
// CPU oracle (tools/head_error_probe.py): 2.4e-7 / 3.1e-7 relative rms - the float32 floor (three bf16 pieces: 2.3e-7 / 3.3e-7;
// two bf16 pieces: 5.9e-6 / 5.1e-6).
__host__ __device__ constexpr int split_np(int scheme) { return scheme == UDA_SPLIT_BF16X3 ? 3 : 2; }   // pieces per operand

// two floats -> one packed piece pair (round to nearest even), element 0 in the low half
template <int SCH>
__device__ __forceinline__ unsigned pack_piece(float a, float b) {
  const f32x2 v = {a, b};
  if constexpr (SCH == UDA_SPLIT_F16X2) return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));   // v_cvt_pk_f16_f32
  else return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));                                   // v_cvt_pk_bf16_f32
}
template <int SCH>
__device__ __forceinline__ float piece_lo(unsigned p) {
  if constexpr (SCH == UDA_SPLIT_F16X2) return (float)__builtin_bit_cast(f16x2, p)[0];
  else return __uint_as_float(p << 16);
}
template <int SCH>
__device__ __forceinline__ float piece_hi(unsigned p) {
  if constexpr (SCH == UDA_SPLIT_F16X2) return (float)__builtin_bit_cast(f16x2, p)[1];
  else return __uint_as_float(p & 0xffff0000u);
}
// fragments travel as bf16x8 (16 bytes) whatever the scheme; the matrix instruction reinterprets them
template <int SCH>
__device__ __forceinline__ f32x16 mfma16(const bf16x8& a, const bf16x8& b, f32x16 acc) {
  if constexpr (SCH == UDA_SPLIT_F16X2)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}
// fp16 range tracking: the largest operand magnitude a lane has split (one v_max3_f32 per pair; nothing for bf16, whose
// exponent range is float32's)
constexpr float UDA_F16_MAX = 65504.0f;
template <int SCH>
__device__ __forceinline__ void split_track(float& amax, float a, float b) {
  // (written as the one instruction it is: fmaxf() first quiets each operand with a v_max_f32 x, |x|, |x| of its own -
  // seven instructions and 16 more live registers per four values in pwb_kernel, which then spilled at three blocks per CU)
  if constexpr (SCH == UDA_SPLIT_F16X2) asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(amax) : "v"(a), "v"(b));
}
template <int SCH>
__device__ __forceinline__ void split_report(float amax, unsigned* flag) {
  if constexpr (SCH == UDA_SPLIT_F16X2) {
    // wave-uniform branch (one ballot): nothing in the kernel's hot path becomes control-dependent on a divergent condition
    if (flag && __builtin_amdgcn_ballot_w64(!(amax <= UDA_F16_MAX)) != 0ull) atomicOr(flag, 1u);
  }
}

// 8 consecutive channels of a pixel -> the pieces of scheme SCH (piece p = round-to-nearest of what pieces 0..p-1 left
// over); amax: see split_track
template <int SCH>
__device__ __forceinline__ void split_parts(const float4& v0, const float4& v1, bf16x8* out, float& amax) {
  constexpr int NPC = split_np(SCH);
  float r0 = v0.x, r1 = v0.y, r2 = v0.z, r3 = v0.w, r4 = v1.x, r5 = v1.y, r6 = v1.z, r7 = v1.w;
  split_track<SCH>(amax, r0, r1); split_track<SCH>(amax, r2, r3); split_track<SCH>(amax, r4, r5); split_track<SCH>(amax, r6, r7);
#pragma unroll
  for (int p = 0; p < NPC; ++p) {
    const unsigned u0 = pack_piece<SCH>(r0, r1), u1 = pack_piece<SCH>(r2, r3), u2 = pack_piece<SCH>(r4, r5), u3 = pack_piece<SCH>(r6, r7);
    out[p] = __builtin_bit_cast(bf16x8, make_uint4(u0, u1, u2, u3));
    if (p + 1 < NPC) {
      r0 -= piece_lo<SCH>(u0); r1 -= piece_hi<SCH>(u0); r2 -= piece_lo<SCH>(u1); r3 -= piece_hi<SCH>(u1);
      r4 -= piece_lo<SCH>(u2); r5 -= piece_hi<SCH>(u2); r6 -= piece_lo<SCH>(u3); r7 -= piece_hi<SCH>(u3);
    }
  }
}
template <int SCH>
__device__ __forceinline__ void split_parts(const float4& v0, const float4& v1, bf16x8* out) {
  float unused = 0.f;
  split_parts<SCH>(v0, v1, out, unused);
}
// the significant cross terms of (sum a[i]) x (sum b[j]), smallest first: two pieces -> a1 b0 + a0 b1 + a0 b0 (bf16: ~2^-17
// per product, fp16: ~2^-22), three bf16 pieces -> + a2 b0 + a0 b2 + a1 b1 in front (~2^-24)
template <int SCH>
__device__ __forceinline__ f32x16 mfma_terms(const bf16x8* a, const bf16x8* b, f32x16 acc) {
  if constexpr (SCH == UDA_SPLIT_BF16X3) {
    acc = mfma16<SCH>(a[2], b[0], acc);
    acc = mfma16<SCH>(a[0], b[2], acc);
    acc = mfma16<SCH>(a[1], b[1], acc);
  }
  acc = mfma16<SCH>(a[1], b[0], acc);
  acc = mfma16<SCH>(a[0], b[1], acc);
  acc = mfma16<SCH>(a[0], b[0], acc);
  return acc;
}

constexpr int PWB_BK = 32;         // k per staged chunk = 2 MFMA k-steps of 16
constexpr int PWB_AROW = 80;       // bytes per A image row: 32 bf16 + 16 pad
constexpr int PWB_STG = 68;        // epilogue staging row stride (floats)

}  // namespace uda
