// Probe (GPU box): the DPP wave reduction of kernels_post.hip (wave_max_key) against a shuffle reduction on random keys.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/dpp_max_probe.hip -o /tmp/dpp_probe && /tmp/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  unsigned t;
  t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); v = t > v ? t : v;   // row_shr:1
  t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); v = t > v ? t : v;   // row_shr:2
  t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xe, false); v = t > v ? t : v;   // row_shr:4
  t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xc, false); v = t > v ? t : v;   // row_shr:8
  t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false); v = t > v ? t : v;   // row_bcast:15
  t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false); v = t > v ? t : v;   // row_bcast:31
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned long long wave_max_key(unsigned long long k) {
  const unsigned hi = (unsigned)(k >> 32), lo = (unsigned)k;
  const unsigned mh = wave_max_u32(hi);
  const unsigned ml = wave_max_u32(hi == mh ? lo : 0u);
  return ((unsigned long long)mh << 32) | ml;
}
__global__ void probe(const unsigned long long* in, unsigned long long* out_dpp, unsigned long long* out_shfl) {
  unsigned long long v = in[blockIdx.x * 64 + threadIdx.x];
  const unsigned long long d = wave_max_key(v);
  for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
  if (threadIdx.x == 0) { out_dpp[blockIdx.x] = d; out_shfl[blockIdx.x] = v; }
  if (threadIdx.x == 37 && d != v) out_dpp[blockIdx.x] = ~0ull;     // every lane must hold the result
}
int main() {
  const int N = 4096;
  std::vector<unsigned long long> h(N * 64);
  srand(7);
  for (auto& x : h) {
    x = ((unsigned long long)(rand() % 5 == 0 ? 0 : (0x80000000u | (rand() & 0xff))) << 32) | (unsigned)rand();   // many equal high words, zeros
  }
  unsigned long long *d, *a, *b;
  hipMalloc(&d, h.size() * 8); hipMalloc(&a, N * 8); hipMalloc(&b, N * 8);
  hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  probe<<<N, 64>>>(d, a, b);
  std::vector<unsigned long long> ha(N), hb(N);
  hipMemcpy(ha.data(), a, N * 8, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), b, N * 8, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < N; ++i) {
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) m = h[i * 64 + l] > m ? h[i * 64 + l] : m;
    if (ha[i] != m || hb[i] != m) { if (bad < 5) printf("wave %d: dpp %llx shfl %llx host %llx\n", i, ha[i], hb[i], m); ++bad; }
  }
  printf("dpp wave max: %d of %d waves differ\n", bad, N);
  return bad != 0;
}
