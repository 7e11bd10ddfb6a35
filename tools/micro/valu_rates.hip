// Micro-benchmark (gfx950): issue rate of v_exp_f32 / v_rcp_f32 / v_mul_f32 / v_pk_mul_f32 / v_fma_f32 per SIMD.
// Every wave runs ITER rounds of 16 independent chains of the instruction; 8 waves per SIMD hide the latency.
// Prints cycles per wave64 instruction per SIMD (4 = full rate for f32).  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, float seed) {
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = seed + 0.001f * (float)(threadIdx.x + i);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
      if (OP == 1) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
      if (OP == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(seed));
      if (OP == 3) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(seed));
      if (OP == 7) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i]) : "v"(seed), "v"(v[(i + 5) & 15]));
      if (OP == 5) asm volatile("v_sqrt_f32 %0, %0" : "+v"(v[i]));
      if (OP == 6) asm volatile("v_log_f32 %0, %0" : "+v"(v[i]));
    }
    if (OP == 8) {      // v_pk_fma_f32: acc pair += a pair * broadcast weight pair
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 p = {v[i], v[i + 1]};
        const f2 q = {seed, seed};
        const f2 r = {v[(i + 6) & 15], v[(i + 7) & 15]};
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p) : "v"(q), "v"(r));
        v[i] = p.x; v[i + 1] = p.y;
      }
    }
    if (OP == 4) {
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 p = {v[i], v[i + 1]};
        const f2 q = {seed, seed};
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(q));
        v[i] = p.x; v[i + 1] = p.y;
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
static double run(const char* name, int per_round, float* d_out, int cus, double ghz) {
  const int iters = 4096, blocks = cus * 8;      // 8 blocks x 4 waves per CU = 8 waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 64, 1.0001f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1.0001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: 8 waves x iters x per_round instructions
  const double inst = 8.0 * iters * per_round;
  const double cyc = ms * 1e-3 * ghz * 1e9 / inst;
  printf("%-14s %8.3f ms  %6.2f cycles per wave64 instruction per SIMD (at %.2f GHz)\n", name, ms, cyc, ghz);
  return cyc;
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const double ghz = p.clockRate * 1e-6;
  printf("%s: %d CUs, %.2f GHz\n", p.gcnArchName, cus, ghz);
  float* d_out;
  hipMalloc(&d_out, (size_t)cus * 8 * 256 * sizeof(float));
  run<2>("v_mul_f32", 16, d_out, cus, ghz);
  run<3>("v_fma_f32", 16, d_out, cus, ghz);
  run<4>("v_pk_mul_f32", 8, d_out, cus, ghz);
  run<7>("v_fmac_f32", 16, d_out, cus, ghz);
  run<8>("v_pk_fma_f32", 8, d_out, cus, ghz);
  run<0>("v_exp_f32", 16, d_out, cus, ghz);
  run<1>("v_rcp_f32", 16, d_out, cus, ghz);
  run<5>("v_sqrt_f32", 16, d_out, cus, ghz);
  run<6>("v_log_f32", 16, d_out, cus, ghz);
  hipFree(d_out);
  return 0;
}
