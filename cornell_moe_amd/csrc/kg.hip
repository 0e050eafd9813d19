// cornell_moe_amd/csrc/kg.hip -- q-KG Monte-Carlo evaluation (value + gradient) on gfx950.
//
// What the reference does per MC sample i (gpp_knowledge_gradient_optimization.cpp:170-196): draw z_i, set the fantasy
// observations y_i = mu(Xu) + L z_i, RE-SOLVE K_after^-1 (y - mean) with two O((N+m)^2) triangular sweeps
// (gpp_math.cpp:531-551), then maximise -mu_after,i(x) from the best discretised start with a back-tracking
// line-search gradient descent (.cpp:420-472, gpp_optimization.hpp:708-828).
//
// What this file does instead -- same mathematics, no N^2 work per sample:
//   K_after^-1 (y_i - mean) = [ K^-1(y - mean) - W beta_i ; beta_i ],   W = K^-1 K*(X,Xu),  beta_i = L^-T z_i,
// (block elimination of the (N+m) system; the reference's own gradient tail relies on the same identity, .cpp:199-209),
// so  mu_after,i(x) = mean + sum_j a_i[j] k(x, Xt_j)  over the N+m points Xt = X u Xu with a per-sample weight vector a_i
// that costs N*m flops to form.  One WAVEFRONT owns one MC sample: lanes stride over the N+m points (coordinates staged
// once per workgroup in LDS, dimension-major so a ds_read_b64 is conflict free), the per-sample weights live in
// registers, and each posterior-mean (or mean + gradient) evaluation ends in a 64-lane butterfly reduction.  Control flow
// of the line search is wave-uniform, so there is no intra-wave divergence; different samples diverge across waves only.
//
// The gradient tail (.cpp:199-225) is evaluated as: T = K(X, x*_i) for all samples (the N x M covariance build, HBM
// write-bound, kernels_cov.hip), S = [W | K^-1 dK*/dXq]^T T (tile GEMM), then one lane per sample finishes the m x m
// algebra and the block-reduced sums are added in a fixed order (bitwise reproducible for a given shard layout).
#include "kg.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>

#include "device_cov.hpp"
#include "fastmath.hpp"

namespace moe {

namespace {

constexpr int kWaves = 8;       // wavefronts (= MC samples) per workgroup
constexpr int kMaxUnionMc = 16;  // q + p limit of the device kernels

struct KgMcParams {
  CovParams cp;
  int n;       // training points (g == 0 -> N == n)
  int u;       // union points (q + p) == m
  int f;       // num_fidelity
  int A;       // discretised-set size (u + P)
  int ldx;     // row length of the LDS coordinate table (NPL * 64)
  double mean;
  const double* XsAll;  // [dp][ldx] scaled coordinates x_k / l_k of the n training points then the u union points, zero padded
  const double* KinvY;  // [n]
  const double* W;      // [n x u], ld = n
  // per-evaluation blob
  const double* Lsm;      // [u x u] col-major lower Cholesky factor of Var(Xu) + noise
  const double* mu_disc;  // [A]      mu_n at the discretised points
  const double* C_disc;   // [A][u]   L^-1 cov_n(Xu, x_j)
  const double* disc;     // [A][d-f] discretised points (unscaled)
  const double* bounds;   // [2 (d-f)]
  const double* normals;  // [ceil(M/2)][u]
  int first_sample, num_local;
  int max_num_steps, max_num_restarts;
  double gamma, pre_mult, max_relative_change, tolerance;
  double* best_point;  // [num_local][dp] (unscaled, fidelity coords = 1, pads = 0)
  double* best_value;  // [num_local]
  unsigned long long* counters;  // [0] value passes, [1] value+gradient passes
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ double uniform(double v) {
  // all lanes hold the same bits after a butterfly; tell the compiler so (value moves to SGPRs, branches become scalar)
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

// base = cov[0,0] / alpha, first = (gradient coefficient) / alpha (see device_cov.hpp) -- the two scalars the inner loop
// needs; alpha is folded into the per-sample weights.  want_first is wave-uniform.
template <int COV>
__device__ __forceinline__ void radial2(double r2, bool want_first, double& base, double& first) {
  if (COV == MOE_COV_SQUARE_EXPONENTIAL) {
    base = exp_nonpos(-0.5 * r2);
    first = base;
  } else {
    const double a = 2.236067977499789696409173668731276235 * sqrt_nonneg(r2);
    const double e = exp_nonpos(-a);
    base = e * fma(a, fma(a, 1.0 / 3.0, 1.0), 1.0);  // e^-a (1 + a + a^2/3)   [5 r2 / 3 == a^2 / 3]
    first = want_first ? (5.0 / 3.0) * (e * (a + 1.0)) : 0.0;
  }
}

// One pass over the N+m points for the wave's sample: returns f = -mu_after(x) and (if want_grad) grad f.
// xq = scaled query coordinates (wave-uniform).  xs = coordinate table [DP][ldx] (LDS, or global when it does not fit),
// aw = this wave's weight vector [ldx] in LDS (zero beyond the N+m real points, so the padded columns contribute 0).
template <int DP, int COV>
__device__ __forceinline__ double eval_point(const double* __restrict__ xs, const double* __restrict__ aw, int ldx,
                                             const double (&xq)[DP], const KgMcParams& P, bool want_grad,
                                             double (&grad)[DP], int lane) {
  double accf = 0.0;
  double accg[DP];
#pragma unroll
  for (int k = 0; k < DP; ++k) accg[k] = 0.0;
#pragma unroll 2
  for (int j = lane; j < ldx; j += 64) {
    double diff[DP];
    double r2 = 0.0;
#pragma unroll
    for (int k = 0; k < DP; ++k) {
      diff[k] = xs[k * ldx + j] - xq[k];
      r2 = fma(diff[k], diff[k], r2);
    }
    const double aj = aw[j];  // alpha * a_i[j]
    double base, first;
    radial2<COV>(r2, want_grad, base, first);
    accf = fma(aj, base, accf);
    if (want_grad) {
      const double coef = aj * first;
#pragma unroll
      for (int k = 0; k < DP; ++k) accg[k] = fma(coef, diff[k], accg[k]);
    }
  }
  const double mu = P.mean + uniform(wave_sum(accf));
  if (want_grad) {
#pragma unroll
    for (int k = 0; k < DP; ++k) {
      // d mu / d x_k = sum a first (X_k - x_k) / l_k^2 = inv_l[k] * sum a first (Xs_k - xq_k);  f = -mu
      grad[k] = -(uniform(wave_sum(accg[k])) * P.cp.inv_l[k]);
    }
  }
  return -mu;
}

// TensorProductDomain::LimitUpdate (gpp_domain.cpp:64-105) on one coordinate.
__device__ __forceinline__ double limit_update_1d(double lo, double hi, double max_relative_change, double x, double desired) {
  double dist = fmin(x - lo, hi - x);
  if (fabs(desired) > max_relative_change * dist) desired = copysign(max_relative_change * dist, desired);
  const double next = x + desired;
  if (next < lo || next > hi) {
    if (next < lo) {
      dist = lo - x;
      desired = (x + desired * 0.5 < lo) ? dist * 0.5 : desired * 0.5;
    } else {
      dist = hi - x;
      desired = (x + desired * 0.5 > hi) ? dist * 0.5 : desired * 0.5;
    }
  }
  return desired;
}

// VectorNorm (gpp_linear_algebra.cpp:53-72)
template <int DP>
__device__ __forceinline__ double vector_norm(const double (&v)[DP], int size) {
  if (size == 1) return fabs(v[0]);
  double scale = 0.0, scaled = 1.0;
#pragma unroll
  for (int i = 0; i < DP; ++i) {
    if (i < size && v[i] != 0.0) {
      const double av = fabs(v[i]);
      if (scale < av) {
        const double t = scale / av;
        scaled = 1.0 + scaled * (t * t);
        scale = av;
      } else {
        const double t = av / scale;
        scaled += t * t;
      }
    }
  }
  return scale * sqrt(scaled);
}

template <int DP, int COV, bool XLDS>
__global__ __launch_bounds__(kWaves * 64) void kg_mc_kernel(KgMcParams P) {
  constexpr int MU = kMaxUnionMc;
  extern __shared__ __attribute__((aligned(16))) double smem[];  // [DP][ldx] coordinates (if XLDS) + [kWaves][ldx] weights
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int ldx = P.ldx;
  const int np = P.n + P.u;
  double* aw = smem + (XLDS ? DP * ldx : 0) + wave * ldx;
  const double* xs = P.XsAll;
  if (XLDS) {
    // ---- stage scaled coordinates of X u Xu (dimension-major) once per workgroup ----
    for (int t = threadIdx.x; t < DP * ldx; t += blockDim.x) smem[t] = P.XsAll[t];
    xs = smem;
    __syncthreads();
  }
  const int sl = blockIdx.x * kWaves + wave;  // local sample index
  if (sl >= P.num_local) return;
  const int s = P.first_sample + sl;  // global sample index
  const int u = P.u;
  const int size = P.cp.dim - P.f;  // problem size of the inner optimisation

  // ---- z_i (antithetic, .cpp:171-180), beta = L^-T z ----
  double z[MU], beta[MU];
  const double sign = (s & 1) ? -1.0 : 1.0;
#pragma unroll
  for (int c = 0; c < MU; ++c) z[c] = (c < u) ? sign * P.normals[(long)(s >> 1) * u + c] : 0.0;
#pragma unroll
  for (int c = MU - 1; c >= 0; --c) {
    double t = z[c];
#pragma unroll
    for (int i = MU - 1; i > c; --i)
      if (i < u) t -= P.Lsm[i + c * u] * beta[i];
    beta[c] = (c < u) ? t / P.Lsm[c + c * u] : 0.0;
  }
  // ---- per-sample weights a_i (see file header) into this wave's LDS row ----
  for (int j = lane; j < ldx; j += 64) {
    double v = 0.0;
    if (j < P.n) {
      v = P.KinvY[j];
#pragma unroll
      for (int c = 0; c < MU; ++c)
        if (c < u) v = fma(-P.W[(long)j + (long)c * P.n], beta[c], v);
    } else if (j < np) {
#pragma unroll
      for (int c = 0; c < MU; ++c)
        if (j - P.n == c) v = beta[c];
    }
    aw[j] = P.cp.alpha * v;
  }
  // (each lane only ever reads back the aw[] entries it wrote itself: j = lane mod 64 -- no cross-lane hazard)

  // ---- discretised-set scan (.cpp:436-449): f_j = -(mu_n(x_j) + c_j . z); keep the FIRST best ----
  double best_f = -INFINITY;
  int best_j = 0;
  for (int j0 = 0; j0 < P.A; j0 += 64) {
    const int j = j0 + lane;
    double fj = -INFINITY;
    if (j < P.A) {
      double v = P.mu_disc[j];
#pragma unroll
      for (int c = 0; c < MU; ++c)
        if (c < u) v = fma(P.C_disc[(long)j * u + c], z[c], v);
      fj = -v;
    }
    double wmax = fj;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wmax = fmax(wmax, __shfl_xor(wmax, off, 64));
    const unsigned long long ballot = __ballot(fj == wmax);
    const int first_lane = __ffsll((long long)ballot) - 1;
    if (wmax > best_f) {  // strict: an earlier chunk wins ties (priority-queue semantics of .cpp:440-447)
      best_f = wmax;
      best_j = j0 + first_lane;
    }
  }
  best_j = __builtin_amdgcn_readfirstlane(best_j);

  double x[DP], xq[DP];
#pragma unroll
  for (int k = 0; k < DP; ++k) {
    x[k] = (k < size) ? P.disc[(long)best_j * size + k] : ((k < P.cp.dim) ? 1.0 : 0.0);
    xq[k] = x[k] * P.cp.inv_l[k];
  }

  unsigned long long n_val = 0, n_grad = 0;  // passes over the N+m points (the A-point scan is O(A m), not counted)
  const double step_tolerance = P.tolerance / (double)P.max_num_steps;
  double fcur = 0.0;

  // GradientDescentOptimizerLineSearch::Optimize (gpp_optimization.hpp:1242-1283) around
  // GradientDescentOptimizationLineSearch (:708-828), written as a wave-uniform state machine with ONE evaluation site,
  // so every posterior-mean value is produced by the same instruction sequence (re-evaluating a point reproduces its
  // value bit for bit, which lets us reuse f(x) where the reference recomputes it).
  enum { PH_GRAD = 0, PH_TRIAL = 1, PH_CLAMPED = 2 };
  if (P.max_num_restarts > 0) {
    int phase = PH_GRAD, restart = 0, istep = 0, search = 0;
    double alpha_n = 0.0, norm = 0.0, f0 = 0.0;
    double grad[DP], step[DP], xstart[DP], tq[DP], gtmp[DP];
#pragma unroll
    for (int k = 0; k < DP; ++k) {
      xstart[k] = x[k];
      grad[k] = 0.0;
      step[k] = 0.0;
    }
    while (true) {
#pragma unroll
      for (int k = 0; k < DP; ++k) {
        double tk = x[k];
        if (k < size) {
          if (phase == PH_TRIAL) tk = x[k] + alpha_n * grad[k];
          if (phase == PH_CLAMPED) tk = x[k] + step[k];
        }
        tq[k] = tk * P.cp.inv_l[k];
      }
      const bool wg = (phase == PH_GRAD);
      const double fval = eval_point<DP, COV>(xs, aw, ldx, tq, P, wg, gtmp, lane);
      bool accept_test = false, end_gd = false;
      double obj2 = 0.0;
      if (phase == PH_GRAD) {
        n_grad++;
        f0 = fval;
        fcur = fval;
        norm = 0.0;
#pragma unroll
        for (int k = 0; k < DP; ++k) {
          grad[k] = gtmp[k];
          if (k < size) norm = fma(grad[k], grad[k], norm);
        }
        alpha_n = P.pre_mult * pow((double)(istep + 1), -P.gamma);
        search = 0;
        phase = PH_TRIAL;
        continue;
      } else if (phase == PH_TRIAL) {
        n_val++;
        const bool armijo = (fval - f0 > 0.5 * alpha_n * norm);
        if (!armijo) {
          alpha_n *= 0.5;
          search += 1;
          if (search < 30) continue;
        }
        bool changed = false, nonzero = false;
#pragma unroll
        for (int k = 0; k < DP; ++k) {
          step[k] = 0.0;
          if (k < size) {
            const double want = alpha_n * grad[k];
            step[k] = limit_update_1d(P.bounds[2 * k], P.bounds[2 * k + 1], P.max_relative_change, x[k], want);
            changed = changed || (step[k] != want);
            nonzero = nonzero || (step[k] != 0.0);
          }
        }
        if (search == 30 || !nonzero) {
          end_gd = true;  // .hpp:781-785: x restored (a zero step re-evaluates f(x) == f0 and is rejected)
        } else if (changed) {
          phase = PH_CLAMPED;
          continue;
        } else {
          obj2 = fval;  // clamp left the step untouched: f(x + step) is the last trial value
          accept_test = true;
        }
      } else {  // PH_CLAMPED
        n_val++;
        obj2 = fval;
        accept_test = true;
      }
      if (accept_test) {
        if (obj2 <= f0) {
          end_gd = true;
        } else {
#pragma unroll
          for (int k = 0; k < DP; ++k) {
            if (k < size) x[k] += step[k];
            xq[k] = x[k] * P.cp.inv_l[k];
          }
          fcur = obj2;
          istep += 1;
          if (vector_norm<DP>(step, size) < step_tolerance || istep >= P.max_num_steps) {
            end_gd = true;
          } else {
            phase = PH_GRAD;
            continue;
          }
        }
      }
      if (end_gd) {
        restart += 1;
        double delta[DP];
#pragma unroll
        for (int k = 0; k < DP; ++k) delta[k] = xstart[k] - x[k];
        if (restart < P.max_num_restarts && vector_norm<DP>(delta, size) > P.tolerance) {
#pragma unroll
          for (int k = 0; k < DP; ++k) xstart[k] = x[k];
          istep = 0;
          phase = PH_GRAD;
          continue;
        }
        break;
      }
    }
  } else {
    // reference returns without touching its outputs (.cpp:425-427): value 0, point filled with 1.0 (.cpp:163)
#pragma unroll
    for (int k = 0; k < DP; ++k) x[k] = (k < P.cp.dim) ? 1.0 : 0.0;
    fcur = 0.0;
  }

  if (lane == 0) {
    P.best_value[sl] = fcur;
    atomicAdd(&P.counters[0], n_val);
    atomicAdd(&P.counters[1], n_grad);
  }
  if (lane < DP) {
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < DP; ++k)
      if (lane == k) v = x[k];
    P.best_point[(long)sl * DP + lane] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Gradient tail, one lane per sample (gpp_knowledge_gradient_optimization.cpp:199-225, gpp_math.cpp:1601-1651 with g = 0).
// ---------------------------------------------------------------------------------------------------------------------
struct KgTailParams {
  CovParams cp;
  int u, q, num_local, first_sample;
  const double* S;       // [(u + q*d) x num_local], ld = u + q*d : [W | G]^T K(X, x*_i)
  const double* best_point;  // [num_local][dp]
  const double* Xu;      // [u][dp] unscaled, padded
  const double* Lsm;     // [u x u]
  const double* Mk;      // [q][d][u x u] col-major: L^-1 (dL/dXq_k,dd)
  const double* normals; // [ceil(M/2)][u]
  double* partial;       // [gridDim.x][q*d]
};

template <int DP, int MU>
__global__ __launch_bounds__(256) void kg_tail_kernel(KgTailParams P) {
  __shared__ double red[4];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = i < P.num_local;
  const int u = P.u, d = P.cp.dim;
  const int ldS = u + P.q * d;
  double z[MU], beta[MU], c[MU];
  double xstar[DP];
  if (active) {
    const int s = P.first_sample + i;
    const double sign = (s & 1) ? -1.0 : 1.0;
#pragma unroll
    for (int r = 0; r < MU; ++r) z[r] = (r < u) ? sign * P.normals[(long)(s >> 1) * u + r] : 0.0;
#pragma unroll
    for (int k = 0; k < DP; ++k) xstar[k] = P.best_point[(long)i * DP + k];
    // beta = L^-T z
#pragma unroll
    for (int r = MU - 1; r >= 0; --r) {
      double t = z[r];
#pragma unroll
      for (int j = MU - 1; j > r; --j)
        if (j < u) t -= P.Lsm[j + r * u] * beta[j];
      beta[r] = (r < u) ? t / P.Lsm[r + r * u] : 0.0;
    }
    // c = L^-1 cov_n(Xu, x*)
#pragma unroll
    for (int r = 0; r < MU; ++r) {
      double v = 0.0;
      if (r < u) {
        double r2 = 0.0;
#pragma unroll
        for (int k = 0; k < DP; ++k) {
          const double df = P.Xu[r * DP + k] - xstar[k];
          r2 = fma(df * df, P.cp.inv_l2[k], r2);
        }
        const Radial rd = radial_scalars(P.cp.type, P.cp.alpha, r2);
        v = rd.base - P.S[(long)r + (long)i * ldS];
#pragma unroll
        for (int j = 0; j < MU; ++j)
          if (j < r) v -= P.Lsm[r + j * u] * c[j];
        v /= P.Lsm[r + r * u];
      }
      c[r] = v;
    }
  }
  for (int k = 0; k < P.q; ++k) {
    double first = 0.0;
    double df[DP];
    if (active) {
      double r2 = 0.0;
#pragma unroll
      for (int kk = 0; kk < DP; ++kk) {
        df[kk] = P.Xu[k * DP + kk] - xstar[kk];
        r2 = fma(df[kk] * df[kk], P.cp.inv_l2[kk], r2);
      }
      first = radial_scalars(P.cp.type, P.cp.alpha, r2).first;
    }
    for (int dd = 0; dd < d; ++dd) {
      double contrib = 0.0;
      if (active) {
        // d cov_n(Xu_k, x*) / d Xu_k,dd = first * (x*_dd - Xu_k,dd) / l^2 - G_{k,dd} . K(X, x*)
        double dfd = 0.0;
#pragma unroll
        for (int kk = 0; kk < DP; ++kk)
          if (kk == dd) dfd = df[kk];
        const double dcov = first * (-dfd * P.cp.inv_l2[dd]) - P.S[(long)(u + k * d + dd) + (long)i * ldS];
        // z^T d c = beta_k dcov - z^T (L^-1 dL) c
        double bk = 0.0;
#pragma unroll
        for (int r = 0; r < MU; ++r)
          if (r == k) bk = beta[r];
        double zMc = 0.0;
        const double* M = P.Mk + ((long)k * d + dd) * u * u;
#pragma unroll
        for (int r = 0; r < MU; ++r) {
          if (r < u) {
            double t = 0.0;
#pragma unroll
            for (int j = 0; j < MU; ++j)
              if (j <= r && j < u) t = fma(M[r + j * u], c[j], t);
            zMc = fma(z[r], t, zMc);
          }
        }
        contrib = -(bk * dcov - zMc);  // aggregate -= gic . z   (.cpp:214-221)
      }
      // block reduction in a fixed order
      double w = wave_sum(contrib);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
      __syncthreads();
      if (threadIdx.x == 0) P.partial[(long)blockIdx.x * (P.q * d) + k * d + dd] = (red[0] + red[1]) + (red[2] + red[3]);
      __syncthreads();
    }
  }
}

// Final fixed-order sums: out[0] = sum_i (best_posterior + best_value_i); out[1 + c] = sum_b partial[b][c] (+ winner term)
__global__ __launch_bounds__(256) void kg_finish_kernel(const double* __restrict__ best_value, int num_local,
                                                       double best_posterior, const double* __restrict__ partial,
                                                       int num_blocks, int ncomp, double* __restrict__ out) {
  __shared__ double red[256];
  for (int comp = -1; comp < ncomp; ++comp) {
    double acc = 0.0;
    if (comp < 0) {
      for (int i = threadIdx.x; i < num_local; i += 256) acc += best_posterior + best_value[i];
    } else {
      for (int b = threadIdx.x; b < num_blocks; b += 256) acc += partial[(long)b * ncomp + comp];
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x == 0) out[comp + 1] = red[0];
    __syncthreads();
  }
}

template <int DP, int COV, bool XLDS>
void launch_mc_inst(const KgMcParams& P, int blocks, size_t shm, hipStream_t s) {
  auto kern = kg_mc_kernel<DP, COV, XLDS>;
  MOE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(kWaves * 64), shm, s, P);
}

template <int DP>
void launch_mc_dp(const KgMcParams& P, bool xlds, int blocks, size_t shm, hipStream_t s) {
  if (P.cp.type == MOE_COV_SQUARE_EXPONENTIAL) {
    if (xlds)
      launch_mc_inst<DP, MOE_COV_SQUARE_EXPONENTIAL, true>(P, blocks, shm, s);
    else
      launch_mc_inst<DP, MOE_COV_SQUARE_EXPONENTIAL, false>(P, blocks, shm, s);
  } else {
    if (xlds)
      launch_mc_inst<DP, MOE_COV_MATERN_NU_2P5, true>(P, blocks, shm, s);
    else
      launch_mc_inst<DP, MOE_COV_MATERN_NU_2P5, false>(P, blocks, shm, s);
  }
}

void launch_mc(const KgMcParams& P, bool xlds, int blocks, size_t shm, hipStream_t s) {
  switch (P.cp.dp) {
    case 4: launch_mc_dp<4>(P, xlds, blocks, shm, s); break;
    case 8: launch_mc_dp<8>(P, xlds, blocks, shm, s); break;
    case 12: launch_mc_dp<12>(P, xlds, blocks, shm, s); break;
    case 16: launch_mc_dp<16>(P, xlds, blocks, shm, s); break;
    default: throw Error(MOE_ERR_RUNTIME, "unsupported padded dimension");
  }
  MOE_HIP_CHECK(hipGetLastError());
}

template <int DP>
void launch_tail_dp(const KgTailParams& P, int blocks, hipStream_t s) {
  if (P.u <= 4)
    hipLaunchKernelGGL((kg_tail_kernel<DP, 4>), dim3(blocks), dim3(256), 0, s, P);
  else if (P.u <= 8)
    hipLaunchKernelGGL((kg_tail_kernel<DP, 8>), dim3(blocks), dim3(256), 0, s, P);
  else
    hipLaunchKernelGGL((kg_tail_kernel<DP, 16>), dim3(blocks), dim3(256), 0, s, P);
}

void launch_tail(const KgTailParams& P, int blocks, hipStream_t s) {
  switch (P.cp.dp) {
    case 4: launch_tail_dp<4>(P, blocks, s); break;
    case 8: launch_tail_dp<8>(P, blocks, s); break;
    case 12: launch_tail_dp<12>(P, blocks, s); break;
    case 16: launch_tail_dp<16>(P, blocks, s); break;
    default: throw Error(MOE_ERR_RUNTIME, "unsupported padded dimension");
  }
  MOE_HIP_CHECK(hipGetLastError());
}

// XsAll[k][j] = x_k / l_k for the n training points (j < n) then the u union points, zero beyond (dimension-major).
__global__ void build_xs_all_kernel(const double* __restrict__ X, int n, const double* __restrict__ Xu, int u, int dp, int ldx,
                                    CovParams cp, double* __restrict__ XsAll) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= dp * ldx) return;
  const int k = t / ldx, j = t - k * ldx;
  double v = 0.0;
  if (j < n)
    v = X[(long)j * dp + k] * cp.inv_l[k];
  else if (j < n + u)
    v = Xu[(long)(j - n) * dp + k] * cp.inv_l[k];
  XsAll[t] = v;
}

struct EventTimer {
  hipEvent_t a, b;
  EventTimer() {
    MOE_HIP_CHECK(hipEventCreate(&a));
    MOE_HIP_CHECK(hipEventCreate(&b));
  }
  ~EventTimer() {
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
  }
  void start(hipStream_t s) { MOE_HIP_CHECK(hipEventRecord(a, s)); }
  void stop(hipStream_t s) { MOE_HIP_CHECK(hipEventRecord(b, s)); }
  double ms() {
    MOE_HIP_CHECK(hipEventSynchronize(b));
    float t = 0.f;
    MOE_HIP_CHECK(hipEventElapsedTime(&t, a, b));
    return t;
  }
};

}  // namespace

void kg_evaluate_batch(GpDev& gp, int num_fidelity, const moe_gd_params_t& gd, const double* bounds, const double* discrete,
                       int P, const double* Xq_all, int num_evals, const double* Xp, int q, int p, int num_mc,
                       double best_so_far, const double* normals, int first_sample, int num_local, bool want_grad,
                       double* kg_sum, double* grad_sum, double* best_points, moe_kg_stats_t* stats) {
  gp.use_device();
  hipStream_t s = gp.stream;
  const int d = gp.d, dp = gp.dp, f = num_fidelity, u = q + p, n = gp.n;
  if (gp.g != 0)
    throw Error(MOE_ERR_RUNTIME, "d-KG (GP with derivative observations) is not implemented on the device path yet");
  if (q <= 0) throw Error(MOE_ERR_BOUNDS, "num_to_sample must be positive", q, 1, 1e9);
  if (p < 0) throw Error(MOE_ERR_BOUNDS, "num_being_sampled must be non-negative", p, 0, 1e9);
  if (u > 16) throw Error(MOE_ERR_BOUNDS, "q + p > 16 is not supported by the device kernels", u, 1, 16);
  if (f < 0 || f >= d) throw Error(MOE_ERR_BOUNDS, "num_fidelity out of range", f, 0, d - 1);
  if (num_mc <= 0) throw Error(MOE_ERR_BOUNDS, "num_mc must be positive", num_mc, 1, 1e12);
  if (first_sample < 0 || (first_sample & 1) || num_local <= 0 || first_sample + num_local > num_mc)
    throw Error(MOE_ERR_INVALID_VALUE, "MC shard must be an even-aligned slice of [0, num_mc)", first_sample, 0, 0);
  if (gd.max_num_steps <= 0) throw Error(MOE_ERR_BOUNDS, "max_num_steps must be positive", gd.max_num_steps, 1, 1e9);
  const int size = d - f;
  const int A = u + P;
  const int np = n + u;
  const int ldx = round_up(np, 64);
  const size_t shm_x = sizeof(double) * (size_t)dp * ldx, shm_a = sizeof(double) * (size_t)kWaves * ldx;
  const bool xlds = (shm_x + shm_a) <= 160 * 1024;
  const size_t shm = (xlds ? shm_x : 0) + shm_a;
  if (shm > 160 * 1024) throw Error(MOE_ERR_RUNTIME, "training set too large for the round-1 MC kernel (weights exceed LDS)");

  DevBuf<double> dXsAll, dBlob, dNormals, dBestPoint, dBestValue, dT, dS, dPartial, dOut;
  DevBuf<unsigned long long> dCounters;
  dXsAll.reserve((size_t)dp * ldx);
  const long num_norm = (long)((num_mc + 1) / 2) * u;
  dNormals.upload(normals, num_norm, s);
  dBestPoint.reserve((size_t)num_local * dp);
  dBestValue.reserve(num_local);
  dCounters.reserve(2);
  const int cW = u + q * d;
  if (want_grad) {
    dT.reserve((size_t)n * num_local);
    dS.reserve((size_t)cW * num_local);
  }
  const int tail_blocks = (num_local + 255) / 256;
  dPartial.reserve((size_t)tail_blocks * q * d);
  dOut.reserve(1 + q * d);
  EventTimer t_mc, t_cov, t_tail, t_all;
  double ms_mc = 0, ms_cov = 0, ms_tail = 0, ms_state = 0;
  if (stats) std::memset(stats, 0, sizeof(*stats));
  const auto wall0 = std::chrono::steady_clock::now();

  std::vector<double> U((size_t)u * d), disc_set((size_t)A * size), extra((size_t)A * d);
  for (int e = 0; e < num_evals; ++e) {
    const auto ws0 = std::chrono::steady_clock::now();
    const double* Xq = Xq_all + (size_t)e * q * d;
    // union of points, discretised set = [Xu without fidelity dims ; discrete points]  (.cpp:246-261)
    std::copy(Xq, Xq + (size_t)q * d, U.begin());
    if (p > 0) std::copy(Xp, Xp + (size_t)p * d, U.begin() + (size_t)q * d);
    for (int i = 0; i < u; ++i) std::copy(&U[(size_t)i * d], &U[(size_t)i * d] + size, &disc_set[(size_t)i * size]);
    std::copy(discrete, discrete + (size_t)P * size, disc_set.begin() + (size_t)u * size);
    for (int j = 0; j < A; ++j) {
      for (int k = 0; k < size; ++k) extra[(size_t)j * d + k] = disc_set[(size_t)j * size + k];
      for (int k = size; k < d; ++k) extra[(size_t)j * d + k] = 1.0;
    }
    StateDev sd;
    StateHost sh;
    DerivList none;
    none.g = 0;
    for (int i = 0; i < kMaxDerivs; ++i) none.idx[i] = 0;
    compute_state(gp, U.data(), u, none, want_grad ? q : 0, extra.data(), A, true, &sd, &sh);
    // ---- PreCompute (.cpp:292-317): mu(Xu), chol(Var + noise) ----
    const int m = u;
    std::vector<double> mu(m), chol((size_t)m * m);
    host_mean(sh, mu.data());
    host_variance(sh, chol.data());
    for (int i = 0; i < u; ++i) chol[i + (size_t)i * m] += gp.noise[0];
    const int lm = host_cholesky(m, chol.data());
    if (lm != 0)
      throw Error(MOE_ERR_SINGULAR,
                  "GP-Variance matrix singular. Check for duplicate points_to_sample/being_sampled or "
                  "points_to_sample/being_sampled duplicating points_sampled with 0 noise.",
                  m, lm);
    for (int c = 0; c < m; ++c)
      for (int r = 0; r < c; ++r) chol[r + (size_t)c * m] = 0.0;
    int winner = -1;
    double best_posterior = best_so_far;
    for (int j = 0; j < u; ++j)
      if (mu[j] < best_posterior) {
        winner = j;
        best_posterior = mu[j];
      }
    // discretised set: mu_n(x_j), c_j = L^-1 cov_n(Xu, x_j)
    std::vector<double> mu_disc(A), C_disc((size_t)A * m);
    for (int j = 0; j < A; ++j) {
      mu_disc[j] = sh.mean + sh.ek[sh.lay.col_extra(j)];
      double cv[16];
      for (int r = 0; r < m; ++r) {
        double kv;
        host_cov(gp.cp, &U[(size_t)r * d], none, &extra[(size_t)j * d], none, &kv);
        cv[r] = kv - sh.G(sh.lay.col_kstar(r, 0), sh.lay.col_extra(j));
      }
      host_tri_solve(chol.data(), 'N', m, cv);
      for (int r = 0; r < m; ++r) C_disc[(size_t)j * m + r] = cv[r];
    }
    // gradient pieces: grad mu, grad chol (Smith), Mk = L^-1 dL
    std::vector<double> grad_mu, Mk;
    if (want_grad) {
      grad_mu.resize((size_t)q * d);
      host_grad_mean(sh, grad_mu.data());
      Mk.assign((size_t)q * d * m * m, 0.0);
      std::vector<double> gc((size_t)d * m * m), col(m);
      for (int k = 0; k < q; ++k) {
        host_grad_cholesky_per_point(sh, k, chol.data(), gc.data());
        for (int dd = 0; dd < d; ++dd) {
          double* M = &Mk[((size_t)k * d + dd) * m * m];
          for (int j = 0; j < m; ++j) {  // column j of dL: entries (l, j), l >= j, stored at gc[dd + j*d + l*d*m]
            for (int l = 0; l < m; ++l) col[l] = (l >= j) ? gc[dd + (size_t)j * d + (size_t)l * d * m] : 0.0;
            host_tri_solve(chol.data(), 'N', m, col.data());
            for (int l = 0; l < m; ++l) M[l + (size_t)j * m] = col[l];
          }
        }
      }
    }
    // ---- blob upload ----
    std::vector<double> blob;
    auto push = [&](const double* ptr, size_t cnt) {
      const size_t off = blob.size();
      blob.insert(blob.end(), ptr, ptr + cnt);
      while (blob.size() % 2) blob.push_back(0.0);
      return off;
    };
    const size_t o_L = push(chol.data(), (size_t)m * m);
    const size_t o_mud = push(mu_disc.data(), A);
    const size_t o_C = push(C_disc.data(), (size_t)A * m);
    const size_t o_disc = push(disc_set.data(), (size_t)A * size);
    std::vector<double> XuP((size_t)u * dp, 0.0);
    for (int i = 0; i < u; ++i)
      for (int k = 0; k < d; ++k) XuP[(size_t)i * dp + k] = U[(size_t)i * d + k];
    const size_t o_XuP = push(XuP.data(), XuP.size());
    const size_t o_bounds = push(bounds, (size_t)2 * size);
    const size_t o_Mk = want_grad ? push(Mk.data(), Mk.size()) : 0;
    dBlob.upload(blob.data(), blob.size(), s);
    MOE_HIP_CHECK(hipMemsetAsync(dCounters.p, 0, 2 * sizeof(unsigned long long), s));
    hipLaunchKernelGGL(build_xs_all_kernel, dim3((dp * ldx + 255) / 256), dim3(256), 0, s, gp.dX.p, n, dBlob.p + o_XuP, u, dp,
                       ldx, gp.cp, dXsAll.p);
    ms_state += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ws0).count();

    // ---- MC kernel ----
    KgMcParams mp;
    mp.cp = gp.cp;
    mp.n = n;
    mp.u = u;
    mp.f = f;
    mp.A = A;
    mp.ldx = ldx;
    mp.mean = gp.mean;
    mp.XsAll = dXsAll.p;
    mp.KinvY = gp.dKinvY.p;
    mp.W = gp.dWE.p;
    mp.Lsm = dBlob.p + o_L;
    mp.mu_disc = dBlob.p + o_mud;
    mp.C_disc = dBlob.p + o_C;
    mp.disc = dBlob.p + o_disc;
    mp.bounds = dBlob.p + o_bounds;
    mp.normals = dNormals.p;
    mp.first_sample = first_sample;
    mp.num_local = num_local;
    mp.max_num_steps = gd.max_num_steps;
    mp.max_num_restarts = gd.max_num_restarts;
    mp.gamma = gd.gamma;
    mp.pre_mult = gd.pre_mult;
    mp.max_relative_change = gd.max_relative_change;
    mp.tolerance = gd.tolerance;
    mp.best_point = dBestPoint.p;
    mp.best_value = dBestValue.p;
    mp.counters = dCounters.p;
    t_mc.start(s);
    launch_mc(mp, xlds, (num_local + kWaves - 1) / kWaves, shm, s);
    t_mc.stop(s);

    // ---- gradient tail ----
    int ncomp = 0;
    if (want_grad) {
      ncomp = q * d;
      t_cov.start(s);
      launch_cov_build(gp.cp, gp.dX.p, n, none, dBestPoint.p, num_local, none, nullptr, dT.p, n, 0, s);
      t_cov.stop(s);
      t_tail.start(s);
      launch_gemm_tn(cW, num_local, n, gp.dWE.p, n, dT.p, n, dS.p, cW, s);
      KgTailParams tp;
      tp.cp = gp.cp;
      tp.u = u;
      tp.q = q;
      tp.num_local = num_local;
      tp.first_sample = first_sample;
      tp.S = dS.p;
      tp.best_point = dBestPoint.p;
      tp.Xu = dBlob.p + o_XuP;
      tp.Lsm = dBlob.p + o_L;
      tp.Mk = dBlob.p + o_Mk;
      tp.normals = dNormals.p;
      tp.partial = dPartial.p;
      launch_tail(tp, tail_blocks, s);
      t_tail.stop(s);
    }
    hipLaunchKernelGGL(kg_finish_kernel, dim3(1), dim3(256), 0, s, dBestValue.p, num_local, best_posterior, dPartial.p,
                       tail_blocks, ncomp, dOut.p);
    MOE_HIP_CHECK(hipGetLastError());
    std::vector<double> out(1 + ncomp);
    dOut.download(out.data(), out.size(), s);
    unsigned long long counters[2] = {0, 0};
    dCounters.download(counters, 2, s);
    if (best_points && num_evals == 1) {
      std::vector<double> bp((size_t)num_local * dp);
      dBestPoint.download(bp.data(), bp.size(), s);
      MOE_HIP_CHECK(hipStreamSynchronize(s));
      for (int i = 0; i < num_local; ++i)
        for (int k = 0; k < d; ++k) best_points[(size_t)i * d + k] = bp[(size_t)i * dp + k];
    }
    MOE_HIP_CHECK(hipStreamSynchronize(s));
    kg_sum[e] = out[0];
    if (want_grad) {
      for (int c = 0; c < ncomp; ++c) grad_sum[(size_t)e * ncomp + c] = out[1 + c];
      // winner term: + M * grad_mu[winner]  (.cpp:157-161); added once, by the shard that owns sample 0
      if (winner >= 0 && winner < q && first_sample == 0)
        for (int k = 0; k < d; ++k) grad_sum[(size_t)e * ncomp + winner * d + k] += (double)num_mc * grad_mu[(size_t)winner * d + k];
    }
    ms_mc += t_mc.ms();
    if (want_grad) {
      ms_cov += t_cov.ms();
      ms_tail += t_tail.ms();
    }
    if (stats) {
      stats->posterior_mean_evals += (long long)counters[0];
      stats->posterior_grad_evals += (long long)counters[1];
    }
  }
  const double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
  gp.last_ms[0] = ms_mc / num_evals;
  gp.last_ms[1] = ms_cov / num_evals;
  gp.last_ms[2] = ms_tail / num_evals;
  gp.last_ms[3] = ms_state / num_evals;
  gp.last_ms[4] = wall / num_evals;
  if (stats) {
    stats->ms_state = ms_state;
    stats->ms_mc = ms_mc;
    stats->ms_tail = ms_cov + ms_tail;
  }
}

}  // namespace moe
