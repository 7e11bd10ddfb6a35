// Probe (GPU box): what ONE grid-wide exchange of the cooperative NMS costs when nothing else happens - the protocol of
// coop_exchange_top (kernels_post.hip): every block stores one 64-bit word (relaxed, agent scope), wave 0 of every block polls
// the problem's words until none is 0.  STEPS exchanges back to back on fresh slots, `bpi` blocks of 1024 threads.
//   spread : blocks 0 .. bpi-1 of the grid (dealt round-robin over the 8 XCDs)
//   one XCD: a grid of 8 x bpi blocks in which only the blocks with blockIdx % 8 == 0 take part (the others leave at once)
//   scope  : agent-scope atomics (what the kernel uses) / the same loads with sc0 only (may hit the XCD's L2; ONLY valid if the
//            blocks really share an XCD - the probe counts time-outs and checks XCC_ID)
//   hipcc --offload-arch=gfx950 -O2 tools/micro/exchange_probe.hip -o /tmp/exchange_probe && /tmp/exchange_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int STEPS = 400;

template <int MODE>      // 0: agent-scope atomics, 1: store agent scope, poll with sc0 loads, 2: store sc0 sc1 / poll sc0 (asm)
__global__ __launch_bounds__(1024) void probe(unsigned long long* slots, int bpi, int stride, int* err, unsigned* xcc, long long* cyc) {
  if (blockIdx.x % stride != 0) return;
  const int blk = blockIdx.x / stride;
  if (blk >= bpi) return;
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  if (threadIdx.x == 0) xcc[blk] = id & 0xf;
  const long long t0 = wall_clock64();
  bool dead = false;
  for (int s = 0; s < STEPS && !dead; ++s) {
    unsigned long long* sl = slots + (size_t)s * 64;
    __syncthreads();
    if (threadIdx.x < 64) {
      const int lane = threadIdx.x;
      if (lane == 0) __hip_atomic_store(&sl[blk], 0x8000000000000000ull + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (lane < bpi) {
        unsigned long long v = 0;
        unsigned spins = 0;
        for (;;) {
          if (MODE == 0) v = __hip_atomic_load(&sl[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else {
            const unsigned long long* p = &sl[lane];
            asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
          }
          if (v != 0ull) break;
          if (++spins > (1u << 20)) { *err = 1; dead = true; break; }
        }
      }
    }
    dead = __syncthreads_or(dead ? 1 : 0) != 0;
  }
  if (threadIdx.x == 0) cyc[blk] = wall_clock64() - t0;
}

template <int MODE>
static void run(const char* name, int bpi, int stride) {
  unsigned long long* slots; int* err; unsigned* xcc; long long* cyc;
  hipMalloc(&slots, (size_t)STEPS * 64 * 8); hipMalloc(&err, 4); hipMalloc(&xcc, 64 * 4); hipMalloc(&cyc, 64 * 8);
  float best = 1e9f;
  std::vector<unsigned> hx(64); std::vector<long long> hc(64); int herr = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(slots, 0, (size_t)STEPS * 64 * 8); hipMemset(err, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(bpi * stride), dim3(1024), 0, 0, slots, bpi, stride, err, xcc, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  hipMemcpy(hx.data(), xcc, 64 * 4, hipMemcpyDeviceToHost); hipMemcpy(hc.data(), cyc, 64 * 8, hipMemcpyDeviceToHost);
  hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
  unsigned seen = 0; for (int i = 0; i < bpi; ++i) seen |= 1u << hx[i];
  printf("%-34s bpi %2d: %7.2f us per exchange (kernel %.3f ms, block 0: %.2f us by wall_clock64 at 100 MHz), XCC ids seen 0x%02x, time-out %d\n",
         name, bpi, best * 1e3f / STEPS, best, hc[0] * 0.01 / STEPS, seen, herr);
  hipFree(slots); hipFree(err); hipFree(xcc); hipFree(cyc);
}

int main() {
  for (int bpi : {8, 23, 45}) {
    run<0>("spread, agent scope", bpi, 1);
    run<0>("one XCD, agent scope", bpi > 32 ? 32 : bpi, 8);
    run<1>("one XCD, sc0 polls", bpi > 32 ? 32 : bpi, 8);
  }
  return 0;
}
