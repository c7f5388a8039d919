/* Host check of the two division-free forms the gather kernels use (csrc/elem_kernels.h: scale_elem, norm_elem_rcp /
 * norm_elem_rcp255).  Test infrastructure only: built and run by tests/test_exact_division.py.
 *   mode 0: y / 255 in float32 == q + fma(-255, q, y) * r with q = y * r, r = RN(1 / 255), for EVERY float 0 <= y <= 2^22
 *           (the form is odd in y, so the negative half follows)
 *   mode 1: ((double)x - mu) / sd == two FMA refinement steps on d * r, r = RN(1 / sd) (an IEEE division), over n random and
 *           adversarial (x, mu, sd); also counts how often ONE step already agrees
 * prints "<checked> <mismatches> [<mismatches after one step>]" */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint64_t s[2] = {0x9E3779B97F4A7C15ull, 0xD1B54A32D192ED03ull};
static uint64_t rnd(void) {   /* xorshift128+ */
  uint64_t a = s[0], b = s[1];
  s[0] = b;
  a ^= a << 23;
  s[1] = a ^ b ^ (a >> 17) ^ (b >> 26);
  return s[1] + b;
}
static double u01(void) { return (double)(rnd() >> 11) * (1.0 / 9007199254740992.0); }

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  if (mode == 0) {
    const float r = 1.f / 255.f;
    uint64_t bad = 0, n = 0;
    for (uint32_t bits = 0; bits <= 0x4A800000u; ++bits) {      /* 0 .. 4194304.0f */
      float y;
      memcpy(&y, &bits, 4);
      const float q = y * r;
      const float got = fmaf(fmaf(-255.f, q, y), r, q);
      const float want = y / 255.f;
      if (memcmp(&got, &want, 4) != 0) ++bad;
      ++n;
    }
    printf("%llu %llu\n", (unsigned long long)n, (unsigned long long)bad);
    return 0;
  }
  const long n = argc > 2 ? atol(argv[2]) : 1000000;
  uint64_t bad2 = 0, bad1 = 0;
  for (long i = 0; i < n; ++i) {
    float x;
    double mu, sd;
    const int kind = (int)(rnd() % 6);
    if (kind == 0) {               /* byte colours against running statistics of byte colours */
      x = (float)(rnd() % 256); mu = 255.0 * u01(); sd = sqrt(6000.0 * u01() + 1e-8);
    } else if (kind == 1) {        /* depth in metres, tight distributions */
      x = (float)u01(); mu = u01(); sd = sqrt(1e-4 * u01() + 1e-8);
    } else if (kind == 2) {        /* deviation at its floor sqrt(eps) */
      x = (float)(200.0 * u01() - 100.0); mu = 200.0 * u01() - 100.0; sd = sqrt(1e-8);
    } else if (kind == 3) {        /* quotients next to ties: d = k * sd * (1 +- tiny) */
      sd = exp(20.0 * u01() - 10.0); mu = 0.0;
      x = (float)((double)(rnd() % 4096) * sd * (1.0 + (u01() - 0.5) * 1e-15));
    } else if (kind == 4) {        /* wide dynamic range */
      x = (float)((u01() - 0.5) * exp(40.0 * u01() - 20.0)); mu = (u01() - 0.5) * exp(40.0 * u01() - 20.0); sd = exp(30.0 * u01() - 15.0);
    } else {                       /* random mantissas of sd in one binade, x - mu exact small integers */
      uint64_t m = (rnd() & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
      memcpy(&sd, &m, 8);
      x = (float)(rnd() % 1000); mu = (double)(rnd() % 1000);
    }
    const double d = (double)x - mu;
    const double want = d / sd;
    const double r = 1.0 / sd;
    double q = d * r;
    double e = fma(-sd, q, d);
    q = fma(e, r, q);
    if (memcmp(&q, &want, 8) != 0) ++bad1;
    e = fma(-sd, q, d);
    q = fma(e, r, q);
    if (memcmp(&q, &want, 8) != 0) ++bad2;
  }
  printf("%ld %llu %llu\n", n, (unsigned long long)bad2, (unsigned long long)bad1);
  return 0;
}
