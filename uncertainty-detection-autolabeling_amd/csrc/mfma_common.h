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

constexpr int PWB_BK = 32;         // k per staged chunk = 2 MFMA k-steps of 16
constexpr int PWB_AROW = 80;       // bytes per A image row: 32 bf16 + 16 pad
constexpr int PWB_STG = 68;        // epilogue staging row stride (floats)

}  // namespace uda
