// cornell_moe_amd/csrc/query_grad.hip -- round 6: the two gradient posterior-query endpoints on the device.
//
// GaussianProcess::ComputeGradVarianceOfPoints[PerPoint] (gpp_math.cpp:1267-1373) and ComputeGradCholeskyVarianceOfPoints[PerPoint]
// (:1389-1474) behind moe_gp_grad_variance / moe_gp_grad_cholesky_variance.  Until r5 the N-sized contractions ran on the GPU (the
// Gram matrix of [K* | dK*/dXs] against K^-1) and the m x m x d algebra -- the variance gradient's assembly and Smith's forward-mode
// derivative of the Cholesky factor -- on the host, over a downloaded Gram matrix.  Here ONE workgroup per differentiated point does
// that algebra on the Gram matrix where it lies (gp.dGram) and writes the block of the output in the reference's layout
// grad[dd + row d + col d m] (d fastest; for the Cholesky variant the meaning is transposed, grad_chol[dd + c d + r d m] =
// d L[r][c] / d Xs_{dd,p}, entries below the block zeroed: :1403-1411): one device->host copy per call, no host arithmetic.
// Every entry goes through the same operations in the same order as csrc/host_math.hip's (which stays: the m x m algebra of the
// variance endpoints below 16 rows and the factor of a variance of fewer than 224 (gp.hpp: device_variance_min_m), held by the golden fixtures, and tools/).
#include <hip/hip_runtime.h>

#include "device_cov.hpp"
#include "gp.hpp"
#include "host_math.hpp"

namespace moe {

namespace {

struct QueryGradParams {
  CovParams cp;
  DerivList dt;
  int d, dp, u, gt, m, nd, c;
  int want_chol;
  const double* gram;  // [c x c] col-major (symmetric): columns [K* | dK*/dXs] contracted against K^-1
  const double* U;     // [u][dp] the state's points
  double* chol;        // [nd][m x m] scratch: Var and then its factor (Cholesky variant)
  double* out;         // [nd][d m m]
  int* info;           // [0]: failing pivot (1-based) of the factorisation, 0 = none
};

__device__ __forceinline__ double gram_at(const QueryGradParams& P, int i, int j) { return P.gram[(size_t)i + (size_t)j * P.c]; }

__global__ __launch_bounds__(256) void query_grad_kernel(QueryGradParams P) {
  const int p = blockIdx.x;
  const int d = P.d, gt = P.gt, g1 = 1 + P.gt, m = P.m, u = P.u;
  const int tid = threadIdx.x, nthr = blockDim.x;
  double* gv = P.out + (size_t)p * d * m * m;
  const size_t total = (size_t)d * m * m;
  __shared__ int s_fail;
  if (tid == 0) s_fail = 0;
  // ---- Var = Kss - K*^T K^-1 K* and its factor (gpp_math.cpp:924-970, gpp_linear_algebra.cpp:109-148; pivot rule 1e-16) ----
  double* ch = P.chol + (size_t)p * m * m;
  if (P.want_chol) {
    for (int e = tid; e < m * m; e += nthr) {
      const int row = e % m, col = e / m;
      const int i = row / g1, a = row % g1, j = col / g1, b = col % g1;
      const PointDiff df{P.U + (size_t)i * P.dp, P.U + (size_t)j * P.dp};
      const Radial rd = pair_radial(P.cp, df, d);
      ch[e] = cov_entry_g(P.cp, rd, df, a, b, P.dt, P.dt) - gram_at(P, row, col);
    }
    __syncthreads();
    for (int k = 0; k < m; ++k) {
      double* col = ch + (size_t)k * m;
      const double pivot = col[k];
      if (!(pivot > 1.0e-16)) {
        if (tid == 0) s_fail = k + 1;
        break;  // (uniform: every thread read the same value)
      }
      const double akk = sqrt(pivot);
      __syncthreads();
      for (int j = k + 1 + tid; j < m; j += nthr) col[j] /= akk;
      if (tid == 0) col[k] = akk;
      __syncthreads();
      const int rem = m - k - 1;
      for (int e = tid; e < rem * rem; e += nthr) {
        const int j = k + 1 + e / rem, i = k + 1 + e % rem;
        if (i >= j) ch[(size_t)j * m + i] = ch[(size_t)j * m + i] - col[i] * col[j];
      }
      __syncthreads();
    }
    __syncthreads();
    if (s_fail != 0) {
      if (tid == 0 && p == 0) P.info[0] = s_fail;
      return;
    }
  }
  // ---- grad Var wrt point p (gpp_math.cpp:1267-1357; host_math.hip: host_grad_variance_per_point) ----
  for (size_t e = tid; e < total; e += nthr) gv[e] = 0.0;
  __syncthreads();
  // column block p:  -(d K*_p)^T K^-1 K*_j
  for (int e = tid; e < g1 * m * d; e += nthr) {
    const int dd = e % d, row = (e / d) % m, a = e / (d * m);
    const int col = p * g1 + a;
    gv[dd + (size_t)row * d + (size_t)col * d * m] = -gram_at(P, m + (p * g1 + a) * d + dd, row);
  }
  __syncthreads();
  // (p, p) block: both factors depend on Xs_p
  for (int e = tid; e < g1 * g1 * d; e += nthr) {
    const int dd = e % d, a = (e / d) % g1, b = e / (d * g1);
    if (b < a) continue;
    const size_t row = (size_t)p * g1 + a, col = (size_t)p * g1 + b;
    const size_t i1 = dd + row * d + col * d * m, i2 = dd + col * d + row * d * m;
    const double v = gv[i1] + gv[i2];
    gv[i1] = v;
    gv[i2] = v;
  }
  __syncthreads();
  // + d Kss / d Xs_p
  for (int e = tid; e < u * g1 * g1 * d; e += nthr) {
    const int dd = e % d, b = (e / d) % g1, a = (e / (d * g1)) % g1, j = e / (d * g1 * g1);
    const PointDiff df{P.U + (size_t)p * P.dp, P.U + (size_t)j * P.dp};
    const Radial rd = pair_radial(P.cp, df, d);
    const size_t row = (size_t)j * g1 + a, col = (size_t)p * g1 + b;
    // host: tmp[dd + x d + y d (1+gt)] = d cov(Xs_p, Xs_j)[x, y] / d Xs_p,dd
    double add = grad_cov_entry_g(P.cp, rd, df, b, a, dd, P.dt, P.dt);
    if (j == p) add += grad_cov_entry_g(P.cp, rd, df, a, b, dd, P.dt, P.dt);
    gv[dd + row * d + col * d * m] += add;
  }
  __syncthreads();
  // mirror block column p into block row p
  for (int e = tid; e < g1 * u * g1 * d; e += nthr) {
    const int dd = e % d, b = (e / d) % g1, j = (e / (d * g1)) % u, i = e / (d * g1 * u);
    if (j == p) continue;
    const size_t row = (size_t)p * g1 + i, col = (size_t)j * g1 + b;
    gv[dd + d * row + (size_t)d * m * col] = gv[dd + d * col + (size_t)d * m * row];
  }
  __syncthreads();
  if (!P.want_chol) return;
  // ---- Smith's forward-mode derivative of the factor (gpp_math.cpp:1389-1452; host_grad_cholesky_per_point) ----
  const double kMinimumStdDev = 2.220446049250313e-16;  // gpp_math.hpp:291
  for (size_t e = tid; e < total; e += nthr) {
    const int i = (int)(e / ((size_t)m * d));       // column block index
    const int within = (int)(e % ((size_t)m * d));  // dd + row d
    if (within >= (i + 1) * d) gv[e] = 0.0;
  }
  __syncthreads();
#define MOE_CH(i, j) ch[(size_t)(j)*m + (i)]
#define MOE_GC(dd, i, j) gv[(size_t)(j)*m * d + (size_t)(i)*d + (dd)]
  for (int k = 0; k < m; ++k) {
    const double Lkk = MOE_CH(k, k);
    if (!(Lkk > kMinimumStdDev)) continue;  // (uniform)
    for (int dd = tid; dd < d; dd += nthr) MOE_GC(dd, k, k) = 0.5 * MOE_GC(dd, k, k) / Lkk;
    __syncthreads();
    const int rem = m - k - 1;
    for (int e = tid; e < rem * d; e += nthr) {
      const int dd = e % d, j = k + 1 + e / d;
      MOE_GC(dd, k, j) = (MOE_GC(dd, k, j) - MOE_CH(j, k) * MOE_GC(dd, k, k)) / Lkk;
    }
    __syncthreads();
    for (long e = tid; e < (long)rem * rem * d; e += nthr) {
      const int dd = (int)(e % d), j = k + 1 + (int)((e / d) % rem), i = k + 1 + (int)(e / ((long)d * rem));
      if (i >= j) MOE_GC(dd, j, i) = MOE_GC(dd, j, i) - MOE_GC(dd, k, i) * MOE_CH(j, k) - MOE_CH(i, k) * MOE_GC(dd, k, j);
    }
    __syncthreads();
  }
#undef MOE_CH
#undef MOE_GC
}

}  // namespace

// moe_gp_grad_variance / moe_gp_grad_cholesky_variance (api.hip): out[num_derivs][d m m]
void grad_variance_on_device(GpDev& gp, const double* pts, int num_pts, int num_derivs, bool cholesky, double* out) {
  if (num_derivs <= 0) return;
  const StateEnqueued se = enqueue_state_batch(gp, pts, num_pts, gp.derivs, num_derivs, nullptr, 0, false, 1);
  hipStream_t s = gp.stream;
  const StateLayout& lay = se.lay;
  const int m = lay.m, d = gp.d;
  const size_t blk = (size_t)d * m * m;
  gp.dVarWork.reserve((size_t)num_derivs * (blk + (size_t)m * m));
  gp.dInfo.reserve(4);
  MOE_HIP_CHECK(hipMemsetAsync(gp.dInfo.p, 0, sizeof(int) * 4, s));
  QueryGradParams P;
  P.cp = gp.cp;
  P.dt = gp.derivs;
  P.d = d;
  P.dp = gp.dp;
  P.u = num_pts;
  P.gt = gp.derivs.g;
  P.m = m;
  P.nd = num_derivs;
  P.c = lay.c();
  P.want_chol = cholesky ? 1 : 0;
  P.gram = gp.dGram.p;
  P.U = gp.dUnion;
  P.out = gp.dVarWork.p;
  P.chol = gp.dVarWork.p + (size_t)num_derivs * blk;
  P.info = gp.dInfo.p;
  MOE_LAUNCH(query_grad_kernel, dim3(num_derivs), dim3(256), 0, s, P);
  MOE_HIP_CHECK(hipGetLastError());
  gp.dVarWork.download(out, (size_t)num_derivs * blk, s);
  int info[4] = {0, 0, 0, 0};
  gp.dInfo.download(info, 4, s);
  MOE_HIP_CHECK(hipStreamSynchronize(s));
  if (info[0] != 0)
    throw Error(MOE_ERR_SINGULAR,
                "GP-Variance matrix singular. Check for duplicate points_to_sample or points_to_sample "
                "duplicating points_sampled with 0 noise.",
                m, info[0]);
}

}  // namespace moe
