// Host check of glenet_amd/csrc/glx_libm.h (the device sinf / cosf / atanf / atan2f) against the installed libm:
// the header compiles unchanged for the host, so this tests the statements the kernels run.
//   g++ -O2 -ffp-contract=off -mfma -fopenmp -I glenet_amd/csrc tools/libm_check.cpp -o /tmp/libm_check -lm
//   /tmp/libm_check            # all finite floats for sinf / cosf / atanf, 4e8 pairs for atan2f  (~1 min on 8 cores)
//   /tmp/libm_check quick      # every 4099th float, 2e6 pairs (the CPU test suite runs this)
// Prints one JSON line; exit code 1 when any result differs in any bit.
#include <stdio.h>
#include <stdlib.h>

#include "glx_libm.h"

static inline bool same(float a, float b) {
  return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b);
}

int main(int argc, char** argv) {
  const bool quick = argc > 1 && !strcmp(argv[1], "quick");
  const uint32_t step = quick ? 4099u : 1u;
  long bad_sin = 0, bad_cos = 0, bad_atan = 0, bad_atan2 = 0, n1 = 0;
#pragma omp parallel for reduction(+ : bad_sin, bad_cos, bad_atan, n1) schedule(static)
  for (uint32_t u = 0; u < 0x7f800000u; u += step)
    for (int sg = 0; sg < 2; ++sg) {
      const float f = __uint_as_float(u | ((uint32_t)sg << 31));
      bad_sin += !same(sinf(f), glxm::sinf_(f));
      bad_cos += !same(cosf(f), glxm::cosf_(f));
      bad_atan += !same(atanf(f), glxm::atanf_(f));
      ++n1;
    }
  const long pairs = quick ? 2000000L : 400000000L;
#pragma omp parallel for reduction(+ : bad_atan2) schedule(static)
  for (long i = 0; i < pairs; ++i) {
    uint64_t st = 88172645463325252ULL + 0x9E3779B97F4A7C15ULL * (uint64_t)(i + 1);   // splitmix64 of the index
    st = (st ^ (st >> 30)) * 0xBF58476D1CE4E5B9ULL;
    st = (st ^ (st >> 27)) * 0x94D049BB133111EBULL;
    st ^= st >> 31;
    const uint32_t a = (uint32_t)st, b = (uint32_t)(st >> 32);
    float y, x;
    if (i & 1) {               // any two finite bit patterns
      y = __uint_as_float(a); x = __uint_as_float(b);
      if (!(fabsf(x) < INFINITY) || !(fabsf(y) < INFINITY)) continue;
    } else {                   // the polygon sort's range: differences of coordinates, a few metres
      y = (float)(int32_t)a * (8.0f / 1073741824.0f); x = (float)(int32_t)b * (8.0f / 1073741824.0f);
    }
    bad_atan2 += !same(atan2f(y, x), glxm::atan2f_(y, x));
  }
  printf("{\"values_1d\": %ld, \"pairs_atan2f\": %ld, \"sinf_diff\": %ld, \"cosf_diff\": %ld, \"atanf_diff\": %ld, "
         "\"atan2f_diff\": %ld}\n", n1, pairs, bad_sin, bad_cos, bad_atan, bad_atan2);
  return (bad_sin | bad_cos | bad_atan | bad_atan2) ? 1 : 0;
}
