#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
static inline double exp_neg(double x) {
  const double LOG2E = 1.4426950408889634074, LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
  double k = rint(x * LOG2E);
  double r = fma(-k, LN2_HI, x);
  r = fma(-k, LN2_LO, r);
  double p = 1.0 / 6227020800.0;                 // 1/13!
  p = fma(p, r, 1.0 / 479001600.0);
  p = fma(p, r, 1.0 / 39916800.0);
  p = fma(p, r, 1.0 / 3628800.0);
  p = fma(p, r, 1.0 / 362880.0);
  p = fma(p, r, 1.0 / 40320.0);
  p = fma(p, r, 1.0 / 5040.0);
  p = fma(p, r, 1.0 / 720.0);
  p = fma(p, r, 1.0 / 120.0);
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)k);
}
int main(int argc, char** argv) {
  uint64_t s = 88172645463325252ull; long n = atol(argv[1]); long bad = 0, badd = 0; double maxrel = 0;
  for (long i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    float u = (float)((s >> 40) * (1.0 / 16777216.0));      // [0,1)
    float scale = (i & 3) == 0 ? -100.f : ((i & 3) == 1 ? -2.f : ((i & 3) == 2 ? -0.5f : -8.f));
    float e = scale * u * u;
    if (e == 0.f) continue;
    double a = exp((double)e), b = exp_neg((double)e);
    float fa = (float)a, fb = (float)b;
    if (memcmp(&fa, &fb, 4)) { ++bad; if (bad < 5) printf("e=%.9g lib=%.17g mine=%.17g\n", e, a, b); }
    if (a != b) ++badd;
    double rel = fabs(a - b) / a; if (rel > maxrel) maxrel = rel;
  }
  printf("n=%ld float mismatches=%ld double mismatches=%ld max rel=%.3g\n", n, bad, badd, maxrel);
  return 0;
}
