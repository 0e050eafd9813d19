// cornell_moe_amd/csrc/ei.hip -- q,p-EI by Monte Carlo (value + gradient) on gfx950.
//
// ExpectedImprovementEvaluator::ComputeExpectedImprovement / ComputeGradExpectedImprovement (gpp_math.cpp:1991-2126):
//   V = Var(Xu) + 1e-6 I,  L = chol(V),  per sample: y = mu + L z,  I = max(0, max_j (best_so_far - y_j)),  w = argmax;
//   EI = sum I / M;   grad EI[k,:] = (1/M) sum_{I>0} ( -[w == k] grad mu_k  -  sum_j dL[w][j]/dXs_k z_j ).
// One lane per MC sample (the per-sample work is O(u^2), u <= 16); sums are block-reduced in a fixed order and finished
// by a single workgroup, so results are bitwise reproducible.
#include <cmath>
#include <cstring>

#include "kg.hpp"

namespace moe {

namespace {

constexpr int kMaxUnionEi = 16;

struct EiParams {
  int u, q, d, num_mc;
  double best_so_far;
  const double* mu;       // [u]
  const double* L;        // [u x u] col-major lower
  const double* grad_mu;  // [q][d]
  const double* gchol;    // [q][u][u][d]: gchol[k*d*u*u + dd + c*d + r*d*u] = dL[r][c]/dXs_{k,dd}, r >= c
  const double* normals;  // [num_mc][u]  (shared by all evaluations: common random numbers)
  double* partial;        // [E][gridDim.x][1 + q*d]
  int want_grad;
  long blob_stride;       // doubles between consecutive evaluations' (mu, L, grad_mu, gchol) records
};

__device__ __forceinline__ double wave_sum_ei(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__global__ __launch_bounds__(256) void ei_mc_kernel(EiParams P) {
  constexpr int MU = kMaxUnionEi;
  __shared__ double red[4];
  const long eoff = (long)blockIdx.y * P.blob_stride;  // this evaluation's record
  P.mu += eoff;
  P.L += eoff;
  P.grad_mu += eoff;
  P.gchol += eoff;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = i < P.num_mc;
  const int u = P.u, d = P.d;
  const int ncomp = 1 + (P.want_grad ? P.q * d : 0);
  double z[MU];
  double imp = 0.0;
  int winner = u + 1;
  if (active) {
#pragma unroll
    for (int j = 0; j < MU; ++j) z[j] = (j < u) ? P.normals[(long)i * u + j] : 0.0;
#pragma unroll
    for (int r = 0; r < MU; ++r) {
      if (r < u) {
        double y = P.mu[r];
#pragma unroll
        for (int c = 0; c < MU; ++c)
          if (c <= r) y = fma(P.L[r + c * u], z[c], y);
        const double t = P.best_so_far - y;
        if (t > imp) {
          imp = t;
          winner = r;
        }
      }
    }
  }
  for (int comp = 0; comp < ncomp; ++comp) {
    double contrib = 0.0;
    if (active && imp > 0.0) {
      if (comp == 0) {
        contrib = imp;
      } else {
        const int k = (comp - 1) / d, dd = (comp - 1) % d;
        double v = 0.0;
        if (winner == k) v = -P.grad_mu[k * d + dd];
        const double* g = P.gchol + (long)k * d * u * u + dd + (long)winner * d * u;
#pragma unroll
        for (int j = 0; j < MU; ++j)
          if (j <= winner && j < u) v = fma(-g[j * d], z[j], v);
        contrib = v;
      }
    }
    const double w = wave_sum_ei(contrib);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0)
      P.partial[((long)blockIdx.y * gridDim.x + blockIdx.x) * ncomp + comp] = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const double* __restrict__ partial, int num_blocks, int ncomp,
                                                          double* __restrict__ out) {
  __shared__ double red[256];
  partial += (long)blockIdx.x * num_blocks * ncomp;  // one workgroup per evaluation
  out += (long)blockIdx.x * ncomp;
  for (int comp = 0; comp < ncomp; ++comp) {
    double acc = 0.0;
    for (int b = threadIdx.x; b < num_blocks; b += 256) acc += partial[(long)b * ncomp + comp];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x == 0) out[comp] = red[0];
    __syncthreads();
  }
}

}  // namespace

void ei_evaluate_batch(GpDev& gp, const double* Xq_all, int num_evals, const double* Xp, int q, int p, int num_mc,
                       double best_so_far, const double* normals, double* ei, double* grad_ei) {
  gp.use_device();
  hipStream_t s = gp.stream;
  const int d = gp.d, u = q + p, E = num_evals;
  if (q <= 0) throw Error(MOE_ERR_BOUNDS, "num_to_sample must be positive", q, 1, 1e9);
  if (p < 0) throw Error(MOE_ERR_BOUNDS, "num_being_sampled must be non-negative", p, 0, 1e9);
  if (E <= 0) throw Error(MOE_ERR_BOUNDS, "num_evals must be positive", E, 1, 1e9);
  if (u > kMaxUnionEi) throw Error(MOE_ERR_BOUNDS, "q + p > 16 is not supported by the device kernels", u, 1, kMaxUnionEi);
  if (num_mc <= 0) throw Error(MOE_ERR_BOUNDS, "num_mc must be positive", num_mc, 1, 1e12);
  const bool want_grad = grad_ei != nullptr;
  std::vector<double> U_all((size_t)E * u * d);
  for (int e = 0; e < E; ++e) {
    std::copy(Xq_all + (size_t)e * q * d, Xq_all + (size_t)(e + 1) * q * d, &U_all[(size_t)e * u * d]);
    if (p > 0) std::copy(Xp, Xp + (size_t)p * d, &U_all[(size_t)e * u * d + (size_t)q * d]);
  }
  DerivList none;
  none.g = 0;
  for (int i = 0; i < kMaxDerivs; ++i) none.idx[i] = 0;
  // EI points carry no derivative observations even when the GP does (ExpectedImprovementState, gpp_math.cpp:2149-2150)
  std::vector<StateHost> hosts;
  compute_state_batch(gp, U_all.data(), u, none, want_grad ? q : 0, nullptr, 0, false, E, nullptr, &hosts);
  // per-evaluation record: mu [u] | L [u*u] | grad_mu [q*d] | gchol [q*d*u*u]
  const size_t o_mu = 0, o_L = u, o_gmu = o_L + (size_t)u * u, o_gc = o_gmu + (size_t)q * d;
  const size_t rec = o_gc + (want_grad ? (size_t)q * d * u * u : 0);
  const size_t n_norm = (size_t)num_mc * u;
  gp.hKgIn.reserve(rec * E + n_norm);
  double* blob = gp.hKgIn.p;
  for (int e = 0; e < E; ++e) {
    const StateHost& sh = hosts[e];
    double* r = blob + rec * e;
    double* chol = r + o_L;
    host_mean(sh, r + o_mu);
    host_variance(sh, chol);
    for (int i = 0; i < u; ++i) chol[i + (size_t)i * u] += 1.0e-6;  // gpp_math.cpp:2000-2002
    const int lm = host_cholesky(u, chol);
    if (lm != 0)
      throw Error(MOE_ERR_SINGULAR,
                  "GP-Variance matrix singular. Check for duplicate points_to_sample/being_sampled or "
                  "points_to_sample/being_sampled duplicating points_sampled with 0 noise.",
                  u, lm);
    if (want_grad) {
      host_grad_mean(sh, r + o_gmu);
      for (int k = 0; k < q; ++k) host_grad_cholesky_per_point(sh, k, chol, r + o_gc + (size_t)k * d * u * u);
    }
  }
  std::memcpy(blob + rec * E, normals, sizeof(double) * n_norm);
  const int ncomp = 1 + (want_grad ? q * d : 0);
  const int blocks = (num_mc + 255) / 256;
  // persistent workspaces (hipMalloc / hipFree per call cost more than the whole evaluation)
  DevBuf<double>&dBlob = gp.kBlob, &dPartial = gp.kTB, &dOut = gp.kOut;
  dBlob.upload(blob, rec * E + n_norm, s);
  dPartial.reserve((size_t)E * blocks * ncomp);
  dOut.reserve((size_t)E * ncomp);
  EiParams P;
  P.u = u;
  P.q = q;
  P.d = d;
  P.num_mc = num_mc;
  P.best_so_far = best_so_far;
  P.mu = dBlob.p + o_mu;
  P.L = dBlob.p + o_L;
  P.grad_mu = dBlob.p + o_gmu;
  P.gchol = dBlob.p + o_gc;
  P.normals = dBlob.p + rec * E;
  P.partial = dPartial.p;
  P.want_grad = want_grad ? 1 : 0;
  P.blob_stride = (long)rec;
  hipLaunchKernelGGL(ei_mc_kernel, dim3(blocks, E), dim3(256), 0, s, P);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(E), dim3(256), 0, s, dPartial.p, blocks, ncomp, dOut.p);
  MOE_HIP_CHECK(hipGetLastError());
  gp.hKgOut.reserve((size_t)E * ncomp);
  double* out = gp.hKgOut.p;
  dOut.download(out, (size_t)E * ncomp, s);
  MOE_HIP_CHECK(hipStreamSynchronize(s));
  for (int e = 0; e < E; ++e) {
    if (ei) ei[e] = out[(size_t)e * ncomp] / (double)num_mc;
    if (want_grad)
      for (int c = 0; c < q * d; ++c) grad_ei[(size_t)e * q * d + c] = out[(size_t)e * ncomp + 1 + c] / (double)num_mc;
  }
}

void ei_evaluate(GpDev& gp, const double* Xq, const double* Xp, int q, int p, int num_mc, double best_so_far,
                 const double* normals, double* ei, double* grad_ei) {
  ei_evaluate_batch(gp, Xq, 1, Xp, q, p, num_mc, best_so_far, normals, ei, grad_ei);
}

// OnePotentialSampleExpectedImprovementEvaluator (gpp_math.cpp:2195-2259) for `num_evals` single points: the posterior at
// one point is a scalar Gaussian, so EI = (best - mu) Phi(c) + sigma phi(c), c = (best - mu) / sigma.  The N-sized work
// (K*, K^-1 K*, grad K*) for ALL points runs in one batched device pass; what is left per point is scalar.
void ei_analytic_batch(GpDev& gp, const double* pts, int num_evals, double best_so_far, double* ei, double* grad_ei) {
  gp.use_device();
  const int d = gp.d, E = num_evals;
  if (E <= 0) throw Error(MOE_ERR_BOUNDS, "num_evals must be positive", E, 1, 1e9);
  const bool want_grad = grad_ei != nullptr;
  DerivList none;
  none.g = 0;
  for (int i = 0; i < kMaxDerivs; ++i) none.idx[i] = 0;
  std::vector<StateHost> hosts;
  compute_state_batch(gp, pts, 1, none, want_grad ? 1 : 0, nullptr, 0, false, E, nullptr, &hosts);
  constexpr double kMinVarEI = 2.2250738585072014e-308;                                    // gpp_math.hpp:1316
  constexpr double kMinVarGradEI = 150.0 * 2.220446049250313e-16 * 2.220446049250313e-16;  // gpp_math.hpp:1323
  auto pdf = [](double z) { return std::exp(-0.5 * z * z) / 2.5066282746310002; };
  auto cdf = [](double z) { return 0.5 * std::erfc(-z * 0.70710678118654752440); };
  std::vector<double> gmu(d), gchol(d);
  for (int e = 0; e < E; ++e) {
    const StateHost& sh = hosts[e];
    double mu, var;
    host_mean(sh, &mu);
    host_variance(sh, &var);
    const double t = best_so_far - mu;
    if (ei) {
      const double sigma = std::sqrt(std::fmax(kMinVarEI, var));
      ei[e] = std::fmax(0.0, t * cdf(t / sigma) + sigma * pdf(t / sigma));
    }
    if (want_grad) {
      const double v = std::fmax(kMinVarGradEI, var);
      double sigma = std::sqrt(v);
      host_grad_mean(sh, gmu.data());
      host_grad_cholesky_per_point(sh, 0, &sigma, gchol.data());
      const double c = t / sigma, pdf_c = pdf(c), cdf_c = cdf(c);
      for (int i = 0; i < d; ++i) {
        const double d_c = (-sigma * gmu[i] - gchol[i] * t) / v;
        const double d_a = -gmu[i] * cdf_c + t * pdf_c * d_c;
        const double d_b = gchol[i] * pdf_c + sigma * (-c) * pdf_c * d_c;
        grad_ei[(size_t)e * d + i] = d_a + d_b;
      }
    }
  }
}

}  // namespace moe
