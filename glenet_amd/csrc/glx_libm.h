// sinf / cosf / atan2f with the RESULT BITS of the host C library the reference's CPU path calls
// (glibc 2.35, x86-64, the FMA-dispatched sinf / cosf and the generic atanf / atan2f) -- device code.
//
// Why: NMS keep masks must be bit-identical to the reference (BASELINE north_star), and `iou > thresh` flips on one
// ulp.  The rotated-overlap arithmetic is restated operation for operation in glx_iou_nms.hip; what is left is the
// three libm calls of iou3d_cpu.cpp (cos / sin of the heading at :83,:141-142, atan2 of the polygon sort at :30).
// The device's own cosf / sinf / atan2f (ocml) are different ~1 ulp routines, and double-precision evaluation rounded
// to float (rounds 1-2) is *correctly* rounded where glibc is not -- either way ~1 % of the overlaps differed in the
// last bit.  These routines follow glibc's published algorithms instead:
//   * sinf / cosf: sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, s_sincosf.h (Szabolcs Nagy's double-precision
//     polynomial kernels): quadrant reduction by multiplication with 2/pi * 2^24, |x| >= 120 through the 192-bit
//     2/pi table, two degree-7/8 polynomials evaluated in double, ONE rounding to float.  x86-64 glibc dispatches the
//     build of that file with FMA contraction (sysdeps/x86_64/fpu/multiarch/s_sinf-fma.c) on every CPU that has FMA,
//     so the a*b+c steps below are written as fma().
//   * atan2f / atanf: sysdeps/ieee754/flt-32/e_atan2f.c, s_atanf.c (the fdlibm float routines: four reduction
//     intervals, an 11-term odd/even split polynomial in float, no contraction).
// Pinned, not assumed: tools/libm_check.cpp evaluates the same statements on the host against the installed libm --
// sinf / cosf / atanf over ALL finite floats (2 x 2 139 095 040 values each) and atan2f on 4e8 pairs: zero differing
// bits (recorded in DESIGN.md section 4); tests/test_libm_gpu.py checks the device build against libm on the GPU box.
// (A host CPU without FMA would run glibc's non-FMA build, which differs from this one on 34 of 2.2e9 arguments
// below 120.)
//
// The header compiles for the device (hipcc) and, unchanged, for the host (g++ -x c++): the exhaustive host check
// therefore tests THIS text, not a copy of it.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define GLXM_FN __device__ __forceinline__
#else
#define GLXM_FN static inline
static inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
#endif

namespace glxm {

GLXM_FN uint32_t top12(float x) { return (__float_as_uint(x) >> 20) & 0x7ffu; }

// one of the two kernels: n even -> sine polynomial in x, n odd -> cosine polynomial in x2; `neg` selects the
// table with negated cosine coefficients (quadrants 2, 3)
GLXM_FN float sincos_poly(double x, double x2, bool neg, int n) {
  const double sg = neg ? -1.0 : 1.0;
  if ((n & 1) == 0) {
    const double s1c = -0x1.555545995a603p-3, s2c = 0x1.1107605230bc4p-7, s3c = -0x1.994eb3774cf24p-13;
    const double x3 = x * x2;
    const double s1 = fma(x2, s3c, s2c);
    const double x7 = x3 * x2;
    const double s = fma(x3, s1c, x);
    return (float)fma(x7, s1, s);
  }
  const double c0 = sg * 0x1p0, c1c = sg * -0x1.ffffffd0c621cp-2, c2c = sg * 0x1.55553e1068f19p-5,
               c3c = sg * -0x1.6c087e89a359dp-10, c4c = sg * 0x1.99343027bf8c3p-16;
  const double x4 = x2 * x2;
  const double c2 = fma(x2, c4c, c3c);
  const double c1 = fma(x2, c1c, c0);
  const double x6 = x4 * x2;
  const double c = fma(x4, c2c, c1);
  return (float)fma(x6, c2, c);
}

// |x| < 120: n = round(x * 2/pi) through a 2^24-scaled float-to-int conversion, x - n * pi/2 in double
GLXM_FN double reduce_fast(double x, int* np) {
  const double r = x * 0x1.45F306DC9C883p+23;
  const int n = ((int32_t)r + 0x800000) >> 24;
  *np = n;
  return fma(-(double)n, 0x1.921FB54442D18p0, x);
}

// |x| >= 120: 96 bits of x * 2/pi from the sliding table of 2/pi's bits
GLXM_FN double reduce_large(uint32_t xi, int* np) {
  const uint32_t inv_pio4[24] = {0xa2u,       0xa2f9u,     0xa2f983u,   0xa2f9836eu, 0xf9836e4eu, 0x836e4e44u,
                                 0x6e4e4415u, 0x4e441529u, 0x441529fcu, 0x1529fc27u, 0x29fc2757u, 0xfc2757d1u,
                                 0x2757d1f5u, 0x57d1f534u, 0xd1f534ddu, 0xf534ddc0u, 0x34ddc0dbu, 0xddc0db62u,
                                 0xc0db6295u, 0xdb629599u, 0x6295993cu, 0x95993c43u, 0x993c4390u, 0x3c439041u};
  const uint32_t* arr = &inv_pio4[(xi >> 26) & 15];
  const int shift = (xi >> 23) & 7;
  xi = (xi & 0xffffffu) | 0x800000u;
  xi <<= shift;
  uint64_t res0 = (uint32_t)(xi * arr[0]);
  const uint64_t res1 = (uint64_t)xi * arr[4];
  const uint64_t res2 = (uint64_t)xi * arr[8];
  res0 = (res2 >> 32) | (res0 << 32);
  res0 += res1;
  const uint64_t n = (res0 + (1ull << 61)) >> 62;
  res0 -= n << 62;
  const double x = (double)(int64_t)res0;
  *np = (int)n;
  return x * 0x1.921FB54442D18p-62;
}

GLXM_FN double quadrant_sign(int n) { return ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0; }

template <int COS>
GLXM_FN float sincosf_impl(float y) {
  double x = (double)y;
  int n;
  if (top12(y) < top12(0x1.921FB6p-1f)) {          // |y| < pi/4
    const double x2 = x * x;
    if (top12(y) < top12(0x1p-12f)) return COS ? 1.0f : y;
    return sincos_poly(x, x2, false, COS);
  }
  if (top12(y) < top12(120.0f)) {
    x = reduce_fast(x, &n);
    const double s = quadrant_sign(n);
    return sincos_poly(x * s, x * x, (n & 2) != 0, n ^ COS);
  }
  if (top12(y) < top12(__uint_as_float(0x7f800000u))) {
    const uint32_t xi = __float_as_uint(y);
    const int sign = (int)(xi >> 31);
    x = reduce_large(xi, &n);
    const double s = quadrant_sign(n + sign);
    return sincos_poly(x * s, x * x, ((n + sign) & 2) != 0, n ^ COS);
  }
  return y - y;                                     // inf / NaN -> NaN
}

GLXM_FN float sinf_(float y) { return sincosf_impl<0>(y); }
GLXM_FN float cosf_(float y) { return sincosf_impl<1>(y); }

GLXM_FN float atanf_(float x) {
  const float hi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
  const float lo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
  const int32_t hx = (int32_t)__float_as_uint(x), ix = hx & 0x7fffffff;
  int id;
  if (ix >= 0x4c000000) {                           // |x| >= 2^25
    if (ix > 0x7f800000) return x + x;
    return hx > 0 ? hi[3] + lo[3] : -hi[3] - lo[3];
  }
  if (ix < 0x3ee00000) {                            // |x| < 7/16
    if (ix < 0x31000000) return x;                  // |x| < 2^-29
    id = -1;
  } else {
    x = fabsf(x);
    if (ix < 0x3f980000) {
      if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }
      else { id = 1; x = (x - 1.0f) / (x + 1.0f); }
    } else {
      if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
      else { id = 3; x = -1.0f / x; }
    }
  }
  const float z = x * x, w = z * z;
  const float s1 = z * (3.3333334327e-01f + w * (1.4285714924e-01f + w * (9.0908870101e-02f +
                   w * (6.6610731184e-02f + w * (4.9768779427e-02f + w * 1.6285819933e-02f)))));
  const float s2 = w * (-2.0000000298e-01f + w * (-1.1111110449e-01f + w * (-7.6918758452e-02f +
                   w * (-5.8335702866e-02f + w * -3.6531571299e-02f))));
  if (id < 0) return x - x * (s1 + s2);
  const float r = hi[id] - ((x * (s1 + s2) - lo[id]) - x);
  return hx < 0 ? -r : r;
}

GLXM_FN float atan2f_(float y, float x) {
  const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f,
              pi_lo = -8.7422776573e-08f;
  const int32_t hx = (int32_t)__float_as_uint(x), ix = hx & 0x7fffffff;
  const int32_t hy = (int32_t)__float_as_uint(y), iy = hy & 0x7fffffff;
  if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
  if (hx == 0x3f800000) return atanf_(y);
  const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
  if (iy == 0) {
    if (m < 2) return y;
    return m == 2 ? pi + tiny : -pi - tiny;
  }
  if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  if (ix == 0x7f800000) {
    if (iy == 0x7f800000) {
      switch (m) {
        case 0: return pi_o_4 + tiny;
        case 1: return -pi_o_4 - tiny;
        case 2: return 3.0f * pi_o_4 + tiny;
        default: return -3.0f * pi_o_4 - tiny;
      }
    }
    switch (m) {
      case 0: return 0.0f;
      case 1: return -0.0f;
      case 2: return pi + tiny;
      default: return -pi - tiny;
    }
  }
  if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  const int k = (iy - ix) >> 23;
  float z;
  if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
  else if (hx < 0 && k < -60) z = 0.0f;
  else z = atanf_(fabsf(y / x));
  switch (m) {
    case 0: return z;
    case 1: return __uint_as_float(__float_as_uint(z) ^ 0x80000000u);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
  }
}

}  // namespace glxm
