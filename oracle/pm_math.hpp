// pm_math.hpp — portable, bit-reproducible double-precision exp / log / tanh.
//
// The reference's NoiseAgent / MomentumAgent price their orders with f64 transcendentals from the platform
// libm (rand_distr LogNormal -> exp, Ziggurat tail -> ln, momentum -> tanh; ref crates/step_sim/src/agents/
// common.rs:104-141, momentum_agent.rs:156).  libm results differ in the last ulp between platforms (host glibc
// vs. device OCML), so the agents here use these self-contained routines built only from IEEE-754 +, -, *, /
// and integer bit manipulation, with contraction into FMA disabled: the SAME source gives bit-identical results
// under g++ on the host and hipcc on gfx950.  Accuracy: <= 2 ulp vs. libm on the ranges used (tests/test_pm_math.py).
// A copy of this file lives under oracle/ (test infrastructure never includes product headers and vice versa;
// tests/test_pm_math.py checks the two copies are identical).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PM_FN __host__ __device__ inline
#else
#define PM_FN inline
#endif

#if defined(__clang__)
#define PM_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define PM_NO_CONTRACT
#endif

namespace pm {

PM_FN double from_bits(uint64_t b) { return __builtin_bit_cast(double, b); }
PM_FN uint64_t to_bits(double d) { return __builtin_bit_cast(uint64_t, d); }
PM_FN double fabs_(double x) { return from_bits(to_bits(x) & 0x7FFFFFFFFFFFFFFFull); }

// 2^k for -1022 <= k <= 1023
PM_FN double pow2i(int k) { return from_bits(static_cast<uint64_t>(1023 + k) << 52); }

// floor for |x| < 2^51 (exact): truncate toward zero, step down for negative non-integers
PM_FN double floor_(double x) {
  PM_NO_CONTRACT
  if (!(fabs_(x) < 4503599627370496.0)) return x;  // already integral (or NaN/inf)
  const double t = static_cast<double>(static_cast<int64_t>(x));
  return (t > x) ? t - 1.0 : t;
}
PM_FN double ceil_(double x) {
  PM_NO_CONTRACT
  if (!(fabs_(x) < 4503599627370496.0)) return x;
  const double t = static_cast<double>(static_cast<int64_t>(x));
  return (t < x) ? t + 1.0 : t;
}

PM_FN double exp(double x) {
  PM_NO_CONTRACT
  if (x != x) return x;
  if (x > 709.782712893384) return from_bits(0x7FF0000000000000ull);  // +inf
  if (x < -745.2) return 0.0;
  // x = k ln2 + r, |r| <= ln2/2, Cody-Waite two-constant reduction
  const double kf = floor_(x * 1.4426950408889634074 + 0.5);
  const int k = static_cast<int>(kf);
  const double r = (x - kf * 6.93147180369123816490e-01) - kf * 1.90821492927058770002e-10;
  // exp(r) = sum r^n / n!, n <= 13 (|r| <= 0.347: truncation < 1e-17)
  double p = 1.0 / 6227020800.0;
  p = p * r + 1.0 / 479001600.0;
  p = p * r + 1.0 / 39916800.0;
  p = p * r + 1.0 / 3628800.0;
  p = p * r + 1.0 / 362880.0;
  p = p * r + 1.0 / 40320.0;
  p = p * r + 1.0 / 5040.0;
  p = p * r + 1.0 / 720.0;
  p = p * r + 1.0 / 120.0;
  p = p * r + 1.0 / 24.0;
  p = p * r + 1.0 / 6.0;
  p = p * r + 0.5;
  p = p * r + 1.0;
  p = p * r + 1.0;
  // scale by 2^k in two steps so that subnormal results and k = 1024 stay representable
  const int k1 = k / 2, k2 = k - k1;
  return p * pow2i(k1) * pow2i(k2);
}

// natural logarithm of a finite x > 0 (subnormals are scaled up first)
PM_FN double log(double x) {
  PM_NO_CONTRACT
  if (x != x || x < 0.0) return from_bits(0x7FF8000000000000ull);
  if (x == 0.0) return from_bits(0xFFF0000000000000ull);
  if (x == from_bits(0x7FF0000000000000ull)) return x;
  int e = 0;
  uint64_t b = to_bits(x);
  if ((b >> 52) == 0) {  // subnormal
    x = x * 18014398509481984.0;  // 2^54
    b = to_bits(x);
    e = -54;
  }
  e += static_cast<int>(b >> 52) - 1023;
  double m = from_bits((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);  // [1, 2)
  if (m > 1.4142135623730951) {  // keep m in [sqrt(1/2), sqrt(2))
    m = m * 0.5;
    e += 1;
  }
  const double s = (m - 1.0) / (m + 1.0);  // |s| <= 0.1716
  const double z = s * s;
  // log(m) = 2 atanh(s) = 2 s (1 + z/3 + z^2/5 + ... + z^13/27)
  double q = 1.0 / 27.0;
  q = q * z + 1.0 / 25.0;
  q = q * z + 1.0 / 23.0;
  q = q * z + 1.0 / 21.0;
  q = q * z + 1.0 / 19.0;
  q = q * z + 1.0 / 17.0;
  q = q * z + 1.0 / 15.0;
  q = q * z + 1.0 / 13.0;
  q = q * z + 1.0 / 11.0;
  q = q * z + 1.0 / 9.0;
  q = q * z + 1.0 / 7.0;
  q = q * z + 1.0 / 5.0;
  q = q * z + 1.0 / 3.0;
  q = q * z + 1.0;
  const double lm = 2.0 * s * q;
  const double ef = static_cast<double>(e);
  return ef * 6.93147180369123816490e-01 + (lm + ef * 1.90821492927058770002e-10);
}

PM_FN double tanh(double x) {
  PM_NO_CONTRACT
  if (x != x) return x;
  const double ax = fabs_(x);
  double r;
  if (ax > 22.0) {
    r = 1.0;
  } else if (ax < 0.125) {
    // x - x^3/3 + 2x^5/15 - 17x^7/315 + 62x^9/2835 - 1382x^11/155925 + 21844x^13/6081075 - 929569x^15/638512875
    const double z = ax * ax;
    double q = -929569.0 / 638512875.0;
    q = q * z + 21844.0 / 6081075.0;
    q = q * z - 1382.0 / 155925.0;
    q = q * z + 62.0 / 2835.0;
    q = q * z - 17.0 / 315.0;
    q = q * z + 2.0 / 15.0;
    q = q * z - 1.0 / 3.0;
    q = q * z + 1.0;
    r = ax * q;
  } else {
    const double t = exp(2.0 * ax);
    r = (t - 1.0) / (t + 1.0);
  }
  return (x < 0.0) ? -r : r;
}

}  // namespace pm
