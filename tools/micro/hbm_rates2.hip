// Micro-benchmark (gfx950): how the access SHAPE of a streaming kernel moves its rate - loads in flight per thread,
// temporal hint, grid-stride vs block-contiguous spans, workgroups per CU.  Context for the HBM-bound kernels (1x1
// projections, separable convs, BiFPN fusion) of DESIGN.md section 4.3.   Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float vf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load(const float4* p) { const vf4 v = __builtin_nontemporal_load((const vf4*)p); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void nt_store(float4 v, float4* p) { vf4 w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, (vf4*)p); }

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? nt_load(a + i + u * stride) : a[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) { if (NT) nt_store(v[u], b + i + u * stride); else b[i + u * stride] = v[u]; }
  }
  for (; i < n; i += stride) b[i] = a[i];
}
// every block streams one contiguous span (as a tile-per-block kernel does), U loads in flight per thread
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy_span(const float4* __restrict__ a, float4* __restrict__ b, size_t n, size_t span) {
  const size_t lo = (size_t)blockIdx.x * span, hi = lo + span < n ? lo + span : n;
  for (size_t i = lo + threadIdx.x; i < hi; i += 256 * U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const size_t j = i + 256 * u; if (j < hi) v[u] = NT ? nt_load(a + j) : a[j]; }
#pragma unroll
    for (int u = 0; u < U; ++u) { const size_t j = i + 256 * u; if (j < hi) { if (NT) nt_store(v[u], b + j); else b[j] = v[u]; } }
  }
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ a, float4* out, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? nt_load(a + i + u * stride) : a[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
  }
  if (s.x == 1.2345f) out[0] = s;
}
template <bool NT>
__global__ __launch_bounds__(256) void k_write(float4* __restrict__ a, size_t n) {
  const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { if (NT) nt_store(v, a + i); else a[i] = v; }
}

template <class F>
static void timeit(const char* name, double bytes, F&& launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-46s %8.3f ms  %7.1f GB/s\n", name, ms / 5, bytes / (ms / 5 * 1e-3) * 1e-9);
}

int main() {
  const size_t n = (size_t)1 << 26;          // float4 elements per GiB
  float4 *a, *b;
  hipMalloc(&a, 2 * n * sizeof(float4));
  hipMalloc(&b, 2 * n * sizeof(float4));
  hipMemset(a, 0, 2 * n * sizeof(float4));
  hipMemset(b, 0, 2 * n * sizeof(float4));
  const size_t m = 2 * n;
  char nm[128];
  for (int g : {256 * 4, 256 * 8, 256 * 16, 256 * 32}) {
    snprintf(nm, sizeof nm, "copy U1 grid %d", g);
    timeit(nm, 2.0 * m * 16, [&] { hipLaunchKernelGGL((k_copy<1, false>), dim3(g), dim3(256), 0, 0, a, b, m); });
    snprintf(nm, sizeof nm, "copy U4 grid %d", g);
    timeit(nm, 2.0 * m * 16, [&] { hipLaunchKernelGGL((k_copy<4, false>), dim3(g), dim3(256), 0, 0, a, b, m); });
    snprintf(nm, sizeof nm, "copy U8 grid %d", g);
    timeit(nm, 2.0 * m * 16, [&] { hipLaunchKernelGGL((k_copy<8, false>), dim3(g), dim3(256), 0, 0, a, b, m); });
    snprintf(nm, sizeof nm, "copy U4 nontemporal grid %d", g);
    timeit(nm, 2.0 * m * 16, [&] { hipLaunchKernelGGL((k_copy<4, true>), dim3(g), dim3(256), 0, 0, a, b, m); });
  }
  for (size_t span : {(size_t)2048, (size_t)8192, (size_t)32768}) {        // float4 per block: 32 KB, 128 KB, 512 KB
    const unsigned g = (unsigned)((m + span - 1) / span);
    snprintf(nm, sizeof nm, "copy span %zu KB U4 (%u blocks)", span * 16 / 1024, g);
    timeit(nm, 2.0 * m * 16, [&] { hipLaunchKernelGGL((k_copy_span<4, false>), dim3(g), dim3(256), 0, 0, a, b, m, span); });
    snprintf(nm, sizeof nm, "copy span %zu KB U8 nt (%u blocks)", span * 16 / 1024, g);
    timeit(nm, 2.0 * m * 16, [&] { hipLaunchKernelGGL((k_copy_span<8, true>), dim3(g), dim3(256), 0, 0, a, b, m, span); });
  }
  timeit("read U1", 1.0 * m * 16, [&] { hipLaunchKernelGGL((k_read<1, false>), dim3(8192), dim3(256), 0, 0, a, b, m); });
  timeit("read U4", 1.0 * m * 16, [&] { hipLaunchKernelGGL((k_read<4, false>), dim3(8192), dim3(256), 0, 0, a, b, m); });
  timeit("read U8", 1.0 * m * 16, [&] { hipLaunchKernelGGL((k_read<8, false>), dim3(8192), dim3(256), 0, 0, a, b, m); });
  timeit("read U8 nontemporal", 1.0 * m * 16, [&] { hipLaunchKernelGGL((k_read<8, true>), dim3(8192), dim3(256), 0, 0, a, b, m); });
  timeit("write", 1.0 * m * 16, [&] { hipLaunchKernelGGL((k_write<false>), dim3(8192), dim3(256), 0, 0, b, m); });
  timeit("write nontemporal", 1.0 * m * 16, [&] { hipLaunchKernelGGL((k_write<true>), dim3(8192), dim3(256), 0, 0, b, m); });
  hipFree(a); hipFree(b);
  return 0;
}
