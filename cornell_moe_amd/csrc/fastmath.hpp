// cornell_moe_amd/csrc/fastmath.hpp -- FP64 exp / sqrt for the covariance inner loops on gfx950.
//
// The inner loops evaluate exp(-a) and sqrt(s) for a, s >= 0 only, with arguments that never approach the overflow /
// denormal ranges, so the generic library sequences (which spend ~1/3 of their instructions on range checks and input
// scaling) are replaced by straight-line code:
//   exp_nonpos(x), x <= 0 : n = rint(x log2 e); r = x - n ln2 (two-term Cody-Waite, fma); degree-11 polynomial
//                           (Chebyshev interpolant of e^r on |r| <= ln2/2, max rel. error 4.2e-18 in exact arithmetic);
//                           v_ldexp_f64.  17 VALU instructions.  Underflows to 0 / denormals gracefully through ldexp.
//   sqrt_nonneg(s), s >= 0: v_rsq_f64 seed + one coupled Newton step + one Heron correction -- the refinement
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

// Table-driven variant for the hottest loop (kg_mc.hpp): x = k ln2/32 + r, |r| <= ln2/64, e^x = 2^(k>>5) * T[k&31] * e^r
// with T[j] = 2^(j/32) (correctly rounded, staged in LDS by the caller) and a degree-6 polynomial for e^r (truncation
// 3.5e-18).  k is extracted with the 1.5*2^52 magic-number trick (no v_rndne / v_cvt).  12 FP64 + 3 integer VALU
// instructions + one conflict-free ds_read_b64, against 17 FP64 for exp_nonpos.  Valid for -1e9 < x <= 0; <= 1.5 ulp.
__device__ __forceinline__ double exp_nonpos_tab(double x, const double* __restrict__ tab32) {
  const double kMagic = 6755399441055744.0;  // 1.5 * 2^52
  const double t = fma(x, 46.16624130844683, kMagic);
  const double kf = t - kMagic;
  double r = fma(kf, -0.02166084938653512, x);
  r = fma(kf, -5.9631716539705866e-12, r);
  const int k = __double2loint(t);  // low mantissa word of t = k (two's complement)
  const double T = tab32[k & 31];
  const double r2 = r * r;
  double q = 1.0 / 720.0;
  q = fma(q, r, 1.0 / 120.0);
  q = fma(q, r, 1.0 / 24.0);
  q = fma(q, r, 1.0 / 6.0);
  q = fma(q, r, 0.5);
  const double sx = fma(r2, q, r);  // e^r - 1
  return ldexp(fma(T, sx, T), k >> 5);
}

// sqrt(s) for s > 0 (no clamp; callers guarantee s >= 1e-300): v_rsq_f64 seed, one coupled Newton step, one Heron
// correction.  Measured correctly rounded (max 0.500 ulp) over 4e6 arguments in [1e-12, 630] (tools/mathcheck.hip); a
// second Heron correction changes nothing.
__device__ __forceinline__ double sqrt_pos(double s) {
  const double y = __builtin_amdgcn_rsq(s);
  double g = s * y;
  double h = 0.5 * y;
  const double e = fma(-h, g, 0.5);
  g = fma(g, e, g);
  h = fma(h, e, h);
  const double d = fma(-g, g, s);
  return fma(d, h, g);
}

__device__ __forceinline__ double sqrt_nonneg(double s) { return sqrt_pos(fmax(s, 1.0e-300)); }

// 2^(j/32), j = 0..31, correctly rounded: the table exp_nonpos_tab expects (callers copy it into LDS).
__device__ __constant__ const double kExp2Tab32[32] = {
    1.0,
    1.0218971486541166,
    1.0442737824274138,
    1.0671404006768237,
    1.0905077326652577,
    1.1143867425958924,
    1.1387886347566916,
    1.1637248587775775,
    1.189207115002721,
    1.215247359980469,
    1.241857812073484,
    1.2690509571917332,
    1.2968395546510096,
    1.3252366431597413,
    1.3542555469368927,
    1.383909881963832,
    1.4142135623730951,
    1.4451808069770467,
    1.4768261459394993,
    1.5091644275934228,
    1.5422108254079407,
    1.5759808451078865,
    1.6104903319492543,
    1.645755478153965,
    1.681792830507429,
    1.718619298122478,
    1.7562521603732995,
    1.7947090750031072,
    1.8340080864093424,
    1.8741676341103,
    1.9152065613971474,
    1.9571441241754002,
};

}  // namespace moe
