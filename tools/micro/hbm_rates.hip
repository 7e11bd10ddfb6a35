// Micro-benchmark (gfx950): what a plain streaming kernel reaches on this box - read-only, write-only, copy and a
// 3:1 read:write mix (the shape of the projection 1x1 convs), 16-byte accesses, 4 GiB working set (no cache reuse).
// The numbers put the HBM-bound kernels of DESIGN.md section 4.3 in context: the 8 TB/s of the data sheet is not
// what any kernel sees.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ a, float4* out, size_t n) {
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float4 v = a[i];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  if (s.x == 1.2345f) out[0] = s;
}
__global__ __launch_bounds__(256) void k_write(float4* __restrict__ a, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    a[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_mix31(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {   // reads 3n, writes n
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float4 x = a[i], y = a[n + i], z = a[2 * n + i];
    b[i] = make_float4(x.x + y.x + z.x, x.y + y.y + z.y, x.z + y.z + z.z, x.w + y.w + z.w);
  }
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
  printf("%-22s %8.3f ms per pass  %7.1f GB/s\n", name, ms / 5, bytes / (ms / 5 * 1e-3) * 1e-9);
}

int main() {
  const size_t n = (size_t)1 << 26;          // float4 elements per GiB
  float4 *a, *b;
  hipMalloc(&a, 3 * n * sizeof(float4));     // 3 GiB
  hipMalloc(&b, n * sizeof(float4));         // 1 GiB
  hipMemset(a, 0, 3 * n * sizeof(float4));
  hipMemset(b, 0, n * sizeof(float4));
  const int grid = 256 * 32;
  timeit("read 3 GiB", 3.0 * n * 16, [&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, b, 3 * n); });
  timeit("write 1 GiB", 1.0 * n * 16, [&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, b, n); });
  timeit("copy 1 GiB -> 1 GiB", 2.0 * n * 16, [&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); });
  timeit("read 3 GiB, write 1", 4.0 * n * 16, [&] { hipLaunchKernelGGL(k_mix31, dim3(grid), dim3(256), 0, 0, a, b, n); });
  hipFree(a); hipFree(b);
  return 0;
}
