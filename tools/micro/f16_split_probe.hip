// Probe for the two-piece fp16 split contraction (round 4):
//   1. does v_mfma_f32_32x32x16_f16 honour fp16 SUBNORMAL operands (or flush them)?
//   2. error of a K-deep dot product against float64 for: bf16 x2 pieces (3 terms), bf16 x3 pieces (6 terms),
//      fp16 x2 pieces (3 terms), on realistic operand magnitudes (activations ~N(0,1), weights ~N(0, sw)).
// build: hipcc --offload-arch=gfx950 -O3 -o f16_split_probe f16_split_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));

// mode 0: bf16 x2 (3 terms), 1: bf16 x3 (6 terms), 2: fp16 x2 (3 terms), 3: fp16 x1
// A [32][K] row-major, B [K][32]; C [32][32].  One wave.
template <int MODE>
__global__ void dot_kernel(const float* A, const float* B, float* C, int K) {
  const int lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
  f16v acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k0 = 0; k0 < K; k0 += 16) {
    float a[8], b[8];
    for (int e = 0; e < 8; ++e) {
      a[e] = A[(size_t)li * K + k0 + 8 * lh + e];
      b[e] = B[(size_t)(k0 + 8 * lh + e) * 32 + li];
    }
    if constexpr (MODE <= 1) {
      constexpr int P = MODE == 1 ? 3 : 2;
      b8 ap[P], bp[P];
      for (int p = 0; p < P; ++p)
        for (int e = 0; e < 8; ++e) {
          ap[p][e] = (__bf16)a[e]; a[e] -= (float)ap[p][e];
          bp[p][e] = (__bf16)b[e]; b[e] -= (float)bp[p][e];
        }
      if constexpr (P == 3) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[2], bp[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], bp[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1], bp[1], acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[1], bp[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], bp[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0], bp[0], acc, 0, 0, 0);
    } else {
      h8 ap[2], bp[2];
      for (int p = 0; p < 2; ++p)
        for (int e = 0; e < 8; ++e) {
          ap[p][e] = (_Float16)a[e]; a[e] -= (float)ap[p][e];
          bp[p][e] = (_Float16)b[e]; b[e] -= (float)bp[p][e];
        }
      if constexpr (MODE == 2) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ap[1], bp[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ap[0], bp[1], acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ap[0], bp[0], acc, 0, 0, 0);
    }
  }
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
    C[row * 32 + li] = acc[r];
  }
}

static double nrand() {
  const double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0);
  return sqrt(-2.0 * log(u1)) * cos(2.0 * M_PI * u2);
}

template <int MODE>
static void run(const char* name, const std::vector<float>& A, const std::vector<float>& B, int K, const std::vector<double>& ref) {
  float *dA, *dB, *dC;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 32 * 32 * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(dot_kernel<MODE>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
  std::vector<float> C(32 * 32);
  hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
  double mx = 0, emax = 0, e2 = 0, r2 = 0;
  for (int i = 0; i < 32 * 32; ++i) {
    mx = fmax(mx, fabs(ref[i]));
    const double e = fabs(C[i] - ref[i]);
    emax = fmax(emax, e); e2 += e * e; r2 += ref[i] * ref[i];
  }
  printf("  %-22s max|err|/max|ref| = %.3e   rms err / rms ref = %.3e\n", name, emax / mx, sqrt(e2 / r2));
  hipFree(dA); hipFree(dB); hipFree(dC);
}

int main() {
  // ---- 1. subnormal operands
  {
    const int K = 16;
    std::vector<float> A(32 * K, 0.f), B(K * 32, 0.f);
    A[0] = ldexpf(1.f, -20);      // fp16 subnormal (normal range ends at 2^-14)
    B[0] = 1024.f;
    A[1 * K + 0] = ldexpf(1.f, -24);   // smallest fp16 subnormal
    std::vector<double> ref(32 * 32, 0.0);
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 32 * 32 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(dot_kernel<3>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
    std::vector<float> C(32 * 32);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    printf("subnormal probe: 2^-20 * 1024 = %.6e (expected %.6e); 2^-24 * 1024 = %.6e (expected %.6e)\n", C[0], ldexp(1.0, -10), C[32], ldexp(1.0, -14));
    hipFree(dA); hipFree(dB); hipFree(dC);
  }
  // ---- 2. accuracy on realistic magnitudes
  srand(1234);
  const int Ks[3] = {96, 672, 1152};
  const double sws[4] = {0.3, 0.04, 0.005, 1e-4};
  const double sas[3] = {1.0, 0.05, 30.0};
  for (int ki = 0; ki < 3; ++ki)
    for (int wi = 0; wi < 4; ++wi)
      for (int ai = 0; ai < 3; ++ai) {
        const int K = Ks[ki];
        std::vector<float> A(32 * K), B(K * 32);
        for (auto& v : A) { const double x = nrand() * sas[ai]; v = (float)(x / (1.0 + exp(-x))); }   // swish-shaped activations
        for (auto& v : B) v = (float)(nrand() * sws[wi]);
        std::vector<double> ref(32 * 32, 0.0);
        for (int i = 0; i < 32; ++i)
          for (int j = 0; j < 32; ++j) {
            double s = 0;
            for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * (double)B[k * 32 + j];
            ref[i * 32 + j] = s;
          }
        // float32 accumulation of exact products, sequential: what a plain f32 FMA loop gives
        double mx = 0, emax = 0;
        for (int i = 0; i < 32; ++i)
          for (int j = 0; j < 32; ++j) {
            float s = 0;
            for (int k = 0; k < K; ++k) s = fmaf(A[i * K + k], B[k * 32 + j], s);
            mx = fmax(mx, fabs(ref[i * 32 + j])); emax = fmax(emax, fabs(s - ref[i * 32 + j]));
          }
        printf("K = %d, weights ~ N(0, %g), activations swish(N(0, %g)):   [f32 fma loop: %.3e]\n", K, sws[wi], sas[ai], emax / mx);
        run<0>("bf16 x2 (3 terms)", A, B, K, ref);
        run<1>("bf16 x3 (6 terms)", A, B, K, ref);
        run<2>("fp16 x2 (3 terms)", A, B, K, ref);
      }
  return 0;
}
