// cornell_moe_amd/csrc/device_cov.hpp -- device-side covariance scalar functions (gfx950).
//
// For both kernels of the reference (gpp_covariance.cpp:121-234 SquareExponential, :339-459 MaternNu2p5) every
// derivative block is a polynomial in the scaled differences times three radial scalars of r^2 = sum (x1-x2)^2 / l^2:
//   base   = cov[0,0]
//   first  = coefficient of the first-derivative entries   (SE: k,        Matern: 5/3 alpha e^-a (a+1))
//   second = coefficient of the u_m v_n Hessian product    (SE: k,        Matern: 25/3 alpha e^-a)
//   third  = coefficient used by d(Hessian)/dx             (SE: k,        Matern: second * sqrt5 / sqrt(r2), 0 at r2 = 0)
// with a = sqrt(5 r2).
#pragma once
#include <hip/hip_runtime.h>

#include "fastmath.hpp"
#include "kernels.hpp"

namespace moe {

struct Radial {
  double base, first, second, third;
};

__host__ __device__ __forceinline__ Radial radial_scalars(int type, double alpha, double r2) {
  Radial r;
#if defined(__HIP_DEVICE_COMPILE__)
#define MOE_EXP_NONPOS(x) exp_nonpos(x)
#define MOE_SQRT_NONNEG(x) sqrt_nonneg(x)
#else  // host side of the same formulas (host_math.hip)
#define MOE_EXP_NONPOS(x) exp(x)
#define MOE_SQRT_NONNEG(x) sqrt(x)
#endif
  if (type == MOE_COV_SQUARE_EXPONENTIAL) {
    const double k = alpha * MOE_EXP_NONPOS(-0.5 * r2);
    r.base = k;
    r.first = k;
    r.second = k;
    r.third = k;
  } else {
    const double s = MOE_SQRT_NONNEG(r2);
    const double a = 2.236067977499789696409173668731276235 * s;
    const double e = MOE_EXP_NONPOS(-a);
    r.base = alpha * e * (1.0 + a + (5.0 / 3.0) * r2);
    r.first = (5.0 / 3.0) * alpha * e * (a + 1.0);
    r.second = (25.0 / 3.0) * alpha * e;
    r.third = (r2 > 0.0) ? r.second * 2.236067977499789696409173668731276235 / s : 0.0;
  }
#undef MOE_EXP_NONPOS
#undef MOE_SQRT_NONNEG
  return r;
}

// cov(p1, p2)[a, b]; a indexes p1's observation (0 = value, 1+m = d/dx_{d1[m]}), b likewise for p2.
// diff[k] = p1[k] - p2[k]  (so u_m = -diff[d1m] / l^2, v_n = +diff[d2n] / l^2).
// (`Diff`: anything indexable -- the register array of the streaming kernels, or PointDiff below for the small m x m algebra)
template <class Diff>
__host__ __device__ __forceinline__ double cov_entry_g(const CovParams& cp, const Radial& rd, const Diff& diff, int a, int b,
                                              const DerivList& d1, const DerivList& d2) {
  if (a == 0 && b == 0) return rd.base;
  if (b == 0) {
    const int i1 = d1.idx[a - 1];
    return rd.first * (-diff[i1] * cp.inv_l2[i1]);
  }
  if (a == 0) {
    const int i2 = d2.idx[b - 1];
    return rd.first * (diff[i2] * cp.inv_l2[i2]);
  }
  const int i1 = d1.idx[a - 1], i2 = d2.idx[b - 1];
  const double u = -diff[i1] * cp.inv_l2[i1], v = diff[i2] * cp.inv_l2[i2];
  double val = u * v * rd.second;
  if (i1 == i2) val += rd.first * cp.inv_l2[i2];
  return val;
}

template <int DP>
__host__ __device__ __forceinline__ double cov_entry(const CovParams& cp, const Radial& rd, const double (&diff)[DP], int a, int b,
                                            const DerivList& d1, const DerivList& d2) {
  return cov_entry_g(cp, rd, diff, a, b, d1, d2);
}

// d cov(p1, p2)[a, b] / d p1_dd   (GradCovariance, gpp_covariance.cpp:171-234 / 389-459)
template <class Diff>
__host__ __device__ __forceinline__ double grad_cov_entry_g(const CovParams& cp, const Radial& rd, const Diff& diff, int a,
                                                   int b, int dd, const DerivList& d1, const DerivList& d2) {
  const double di = -diff[dd] * cp.inv_l2[dd];  // (p2 - p1) / l^2
  if (a == 0 && b == 0) return di * rd.first;
  if (b == 0) {
    const int i1 = d1.idx[a - 1];
    double val = rd.second * di * (-diff[i1] * cp.inv_l2[i1]);
    if (dd == i1) val -= rd.first * cp.inv_l2[i1];
    return val;
  }
  if (a == 0) {
    const int i2 = d2.idx[b - 1];
    double val = rd.second * di * (diff[i2] * cp.inv_l2[i2]);
    if (dd == i2) val += rd.first * cp.inv_l2[i2];
    return val;
  }
  const int i1 = d1.idx[a - 1], i2 = d2.idx[b - 1];
  const double u = -diff[i1] * cp.inv_l2[i1], v = diff[i2] * cp.inv_l2[i2];
  if (cp.type == MOE_COV_SQUARE_EXPONENTIAL) {
    double t = u * v;
    if (i1 == i2) t += cp.inv_l2[i1];
    t *= di;
    if (i1 == dd) t -= v * cp.inv_l2[i1];
    if (i2 == dd) t += u * cp.inv_l2[i2];
    return t * rd.base;
  }
  if (rd.third == 0.0) return 0.0;  // r2 == 0 (gpp_covariance.cpp:452-454)
  double t = rd.third * u * v * di;
  if (i1 == dd) t -= rd.second * v * cp.inv_l2[i1];
  if (i2 == dd) t += rd.second * u * cp.inv_l2[i2];
  if (i1 == i2) t += rd.second * di * cp.inv_l2[i1];
  return t;
}

template <int DP>
__host__ __device__ __forceinline__ double grad_cov_entry(const CovParams& cp, const Radial& rd, const double (&diff)[DP], int a,
                                                 int b, int dd, const DerivList& d1, const DerivList& d2) {
  return grad_cov_entry_g(cp, rd, diff, a, b, dd, d1, d2);
}

// p1 - p2 taken where it is read, and the radial scalars of the pair: for the m x m algebra of a points state (kg_state.hip), where a
// thread owns one matrix entry and a register array of differences would be indexed dynamically.
struct PointDiff {
  const double* p1;
  const double* p2;
  __host__ __device__ __forceinline__ double operator[](int k) const { return p1[k] - p2[k]; }
};
__host__ __device__ __forceinline__ Radial pair_radial(const CovParams& cp, const PointDiff& df, int d) {
  double r2 = 0.0;
  for (int k = 0; k < d; ++k) {
    const double v = df[k];
    r2 = fma(v * v, cp.inv_l2[k], r2);
  }
  return radial_scalars(cp.type, cp.alpha, r2);
}

}  // namespace moe
