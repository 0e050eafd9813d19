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

// Table-driven variant for the hottest loop (kg_mc.hpp): x = k ln2/64 + r, |r| <= ln2/128, e^x = 2^(k>>6) * T[k&63] * e^r
// with T[j] = 2^(j/64) (correctly rounded, staged in LDS by the caller) and a degree-5 polynomial for e^r - 1 (truncation
// r^6/720 <= 3.5e-17).  k is extracted with the 1.5*2^52 magic-number trick (no v_rndne / v_cvt); the reduction is one fma
// (the product is exact inside it).  10 FP64 + 3 integer VALU instructions + one ds_read_b64, against 17 FP64 for exp_nonpos.  (A 512-entry table with a degree-4 polynomial -- one fma less, 1.00 ulp --
// measured no faster: its lookups collide on LDS banks where the 64-entry table's mostly broadcast; and it costs 3.5 KB.)  <= 1.5 ulp (tools/mathcheck.hip).  Valid for -2.3e7 < x <= 0 (k must fit
// 32 bits: beyond that the exponent wraps and the result can be anything, including inf) -- callers bound the argument:
// sqrt(5) r <= 1e7 by construction of the tables (kg_mc.hpp to_frame / kTableExtent), -r2/2 by an explicit fmax.
__device__ __forceinline__ double exp_nonpos_tab(double x, const double* __restrict__ tab64) {
  const double kMagic = 6755399441055744.0;  // 1.5 * 2^52
  const double t = fma(x, 92.33248261689366, kMagic);  // 64 / ln2
  const double kf = t - kMagic;
  // (one-term reduction: ln2/64 as a double is 3.3e-17 relative off, i.e. r is off by |x| 3.3e-17 and e^x by the same RELATIVE
  //  amount -- |x| e^x 3.3e-17 <= 1.3e-17 absolute for x <= 0, a tenth of an ulp of the O(1) covariances it feeds; the lo term
  //  of the hi/lo split that used to follow cost one FP64 instruction per covariance entry)
  const double r = fma(kf, -0.010830424696249145, x);  // ln2 / 64
  const int k = __double2loint(t);  // low mantissa word of t = k (two's complement)
  const double T = tab64[k & 63];
  const double r2 = r * r;
  double q = 1.0 / 120.0;
  q = fma(q, r, 1.0 / 24.0);
  q = fma(q, r, 1.0 / 6.0);
  q = fma(q, r, 0.5);
  const double sx = fma(r2, q, r);  // e^r - 1
  return ldexp(fma(T, sx, T), k >> 6);
}

// sqrt(s) for s > 0 (no clamp; callers guarantee s >= 1e-300): v_rsq_f64 seed, one Newton step, one Heron
// correction (7 instructions).  Measured correctly rounded (max 0.500 ulp) over 4e6 arguments in [1e-12, 630] (tools/mathcheck.hip); a
// second Heron correction changes nothing.
__device__ __forceinline__ double sqrt_pos(double s) {
  const double y = __builtin_amdgcn_rsq(s);
  double g = s * y;
  double h = 0.5 * y;
  const double e = fma(-h, g, 0.5);
  g = fma(g, e, g);
  // (h is not refined: it only scales the final correction d ~ 2^-45 g, where its seed accuracy of ~2^-22 is plenty)
  const double d = fma(-g, g, s);
  return fma(d, h, g);
}

__device__ __forceinline__ double sqrt_nonneg(double s) { return sqrt_pos(fmax(s, 1.0e-300)); }

// The same without the coupled Newton step: v_rsq_f64 seed (relative error d <= 5e-8) and ONE Heron correction, which leaves
// -1.5 d^2: at most 36 ulp = 4e-15 relative (tools/mathcheck.hip, "sqrt seed+heron"), two FP64 instructions less (1 + 4 instead of
// 1 + 6).  For the Monte-Carlo kernels' distance -> covariance chain (r4): a relative error e in a = sqrt(5) r moves the Matern
// covariance by a e e^-a (...) <= 2e-15 relative -- the size of the rounding the chain's other ~25 operations add up to.
__device__ __forceinline__ double sqrt_pos_fast(double s) {
  const double y = __builtin_amdgcn_rsq(s);
  const double g = s * y;
  const double d = fma(-g, g, s);
  return fma(d, 0.5 * y, g);
}

// 2^(j/64), j = 0..63, correctly rounded: the table exp_nonpos_tab expects (callers copy it into LDS).
__device__ __constant__ const double kExp2Tab64[64] = {
    1.0,
    1.0108892860517005,
    1.0218971486541166,
    1.0330248790212284,
    1.0442737824274138,
    1.0556451783605572,
    1.0671404006768237,
    1.0787607977571199,
    1.0905077326652577,
    1.102382583307841,
    1.1143867425958924,
    1.1265216186082418,
    1.1387886347566916,
    1.1511892299529827,
    1.1637248587775775,
    1.1763969916502812,
    1.189207115002721,
    1.202156731452703,
    1.215247359980469,
    1.22848053610687,
    1.241857812073484,
    1.255380757024691,
    1.2690509571917332,
    1.2828700160787783,
    1.2968395546510096,
    1.3109612115247644,
    1.3252366431597413,
    1.339667524053303,
    1.3542555469368927,
    1.3690024229745905,
    1.383909881963832,
    1.3989796725383112,
    1.4142135623730951,
    1.42961333839197,
    1.4451808069770467,
    1.460917794180647,
    1.4768261459394993,
    1.4929077282912648,
    1.5091644275934228,
    1.5255981507445384,
    1.5422108254079407,
    1.559004400237837,
    1.5759808451078865,
    1.593142151342267,
    1.6104903319492543,
    1.6280274218573478,
    1.645755478153965,
    1.6636765803267364,
    1.681792830507429,
    1.7001063537185235,
    1.718619298122478,
    1.7373338352737062,
    1.7562521603732995,
    1.7753764925265212,
    1.7947090750031072,
    1.8142521755003989,
    1.8340080864093424,
    1.8539791250833855,
    1.8741676341103,
    1.8945759815869656,
    1.9152065613971474,
    1.9360617934922943,
    1.9571441241754002,
    1.978456026387951,
};

}  // namespace moe
