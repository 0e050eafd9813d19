// cornell_moe_amd/csrc/fastmath.hpp -- FP64 exp / sqrt for the covariance inner loops on gfx950.
//
// The inner loops evaluate exp(-a) and sqrt(s) for a, s >= 0 only, with arguments that never approach the overflow /
// denormal ranges, so the generic library sequences (which spend ~1/3 of their instructions on range checks and input
// scaling) are replaced by straight-line code:
//   exp_nonpos(x), x <= 0 : n = rint(x log2 e); r = x - n ln2 (two-term Cody-Waite, fma); degree-11 polynomial
//                           (Chebyshev interpolant of e^r on |r| <= ln2/2, max rel. error 4.2e-18 in exact arithmetic);
//                           v_ldexp_f64.  17 VALU instructions.  Underflows to 0 / denormals gracefully through ldexp.
//   sqrt_nonneg(s), s >= 0: v_rsq_f64 seed + one coupled Newton step + two Heron corrections -- the same refinement
//                           hipcc emits for sqrt() minus the input scaling; s is clamped at 1e-300 so s == 0 gives 1e-150
//                           (indistinguishable from 0 for every use here: it only enters 1 + a + a^2/3 and exp(-a)).
// Accuracy is checked on the GPU against numpy in tests/test_gpu_parity.py::test_fastmath (<= 2 ulp).
#pragma once
#include <hip/hip_runtime.h>

namespace moe {

__device__ __forceinline__ double exp_nonpos(double x) {
  const double n = __builtin_rint(x * 1.4426950408889634074);
  double r = fma(n, -0.6931471805599453094, x);
  r = fma(n, -2.3190468138462995584e-17, r);
  double p = 2.5110049204818658e-08;
  p = fma(p, r, 2.763265472252779e-07);
  p = fma(p, r, 2.755724088722987e-06);
  p = fma(p, r, 2.4801485441561313e-05);
  p = fma(p, r, 0.00019841269890076403);
  p = fma(p, r, 0.0013888888952352863);
  p = fma(p, r, 0.008333333333319589);
  p = fma(p, r, 0.04166666666648795);
  p = fma(p, r, 0.1666666666666668);
  p = fma(p, r, 0.5000000000000019);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, static_cast<int>(n));
}

__device__ __forceinline__ double sqrt_nonneg(double s) {
  s = fmax(s, 1.0e-300);
  const double y = __builtin_amdgcn_rsq(s);
  double g = s * y;
  double h = 0.5 * y;
  const double e = fma(-h, g, 0.5);
  g = fma(g, e, g);
  h = fma(h, e, h);
  double d = fma(-g, g, s);
  g = fma(d, h, g);
  d = fma(-g, g, s);
  g = fma(d, h, g);
  return g;
}

}  // namespace moe
