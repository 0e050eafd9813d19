// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (never linked into, or called by, the product path).
//
// A C-ABI harness around the *unmodified* reference C++ core (wujian16/Cornell-MOE,
// moe/optimal_learning/cpp/*.cpp, compiled in place from /root/reference by oracle/Makefile into
// oracle/_ref/libmoe_ref.so).  It drives the reference classes exactly the way the reference's own
// boost::python boundary does (gpp_python_gaussian_process.cpp:42-236, gpp_python_expected_improvement.cpp:44-109,
// gpp_python_knowledge_gradient.cpp:44-154) but with explicit normal tables injected through
// NormalRNGSimulator (gpp_random.hpp:314-340) so runs are reproducible draw-for-draw.
//
// Uses: (1) pinning oracle/moe_oracle.c (the committed restatement), (2) generating tests/golden fixtures,
// (3) the timed CPU baseline ("kind": "reference") in bench.py.
//
// `#define private public` is used for the one accessor the reference lacks (GaussianProcess::K_chol_); it does not
// change any layout or code path of the reference.

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

#include <omp.h>

#define private public
#include "gpp_math.hpp"
#undef private
#include "gpp_model_selection.hpp"
#include "gpp_common.hpp"
#include "gpp_covariance.hpp"
#include "gpp_domain.hpp"
#include "gpp_exception.hpp"
#include "gpp_geometry.hpp"
#include "gpp_expected_improvement_mcmc_optimization.hpp"
#include "gpp_knowledge_gradient_mcmc_optimization.hpp"
#include "gpp_knowledge_gradient_optimization.hpp"
#include "gpp_linear_algebra.hpp"
#include "gpp_optimization.hpp"
#include "gpp_optimizer_parameters.hpp"
#include "gpp_random.hpp"

using namespace optimal_learning;  // NOLINT

namespace {

thread_local std::string g_last_error;

struct RefGP {
  CovarianceInterface* cov;
  GaussianProcess* gp;
};

int dummy_int = 0;
inline const int* nn(const int* p) { return p ? p : &dummy_int; }

template <typename F>
int guarded(F&& f) {
  try {
    f();
    return 0;
  } catch (const SingularMatrixException& e) {
    g_last_error = e.what();
    return 4;
  } catch (const BoundsException<double>& e) {
    g_last_error = e.what();
    return 2;
  } catch (const BoundsException<int>& e) {
    g_last_error = e.what();
    return 2;
  } catch (const InvalidValueException<double>& e) {
    g_last_error = e.what();
    return 3;
  } catch (const InvalidValueException<int>& e) {
    g_last_error = e.what();
    return 3;
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return 1;
  }
}

CovarianceInterface* make_cov(int cov_type, int dim, double alpha, const double* lengths) {
  if (cov_type == 0) return new SquareExponential(dim, alpha, lengths);
  return new MaternNu2p5(dim, alpha, lengths);
}

}  // namespace

extern "C" {

const char* ref_last_error() { return g_last_error.c_str(); }

// ---- covariance blocks (gpp_covariance.cpp:121-234, 339-459) ----
int ref_covariance(int cov_type, int dim, double alpha, const double* lengths, const double* p1, const int* d1, int g1,
                   const double* p2, const int* d2, int g2, double* cov, double* grad_cov) {
  return guarded([&] {
    CovarianceInterface* c = make_cov(cov_type, dim, alpha, lengths);
    if (cov) c->Covariance(p1, nn(d1), g1, p2, nn(d2), g2, cov);
    if (grad_cov) c->GradCovariance(p1, nn(d1), g1, p2, nn(d2), g2, grad_cov);
    delete c;
  });
}

// ---- linear algebra known-answer entry points (gpp_linear_algebra.cpp:109-208) ----
int ref_cholesky(int n, double* a) { return ComputeCholeskyFactorL(n, a); }
void ref_chol_solve(const double* L, int n, double* b) { CholeskyFactorLMatrixVectorSolve(L, n, b); }
void ref_tri_solve(const double* L, char trans, int n, double* b) { TriangularMatrixVectorSolve(L, trans, n, n, b); }

// ---- GP (gpp_math.cpp:553-573) ----
void* ref_gp_create(int cov_type, double alpha, const double* lengths, const double* X, const double* y,
                    const double* noise, const int* derivs, int g, int d, int n) {
  RefGP* h = nullptr;
  int rc = guarded([&] {
    h = new RefGP{nullptr, nullptr};
    h->cov = make_cov(cov_type, d, alpha, lengths);
    h->gp = new GaussianProcess(*h->cov, X, y, noise, nn(derivs), g, d, n);
  });
  if (rc != 0) {
    if (h) { delete h->cov; delete h; }
    return nullptr;
  }
  return h;
}

void ref_gp_destroy(void* hv) {
  RefGP* h = static_cast<RefGP*>(hv);
  if (!h) return;
  delete h->gp;
  delete h->cov;
  delete h;
}

int ref_gp_num_sampled(void* hv) { return static_cast<RefGP*>(hv)->gp->num_sampled(); }

int ref_gp_add_points(void* hv, const double* pts, const double* vals, int k) {
  return guarded([&] { static_cast<RefGP*>(hv)->gp->AddPointsToGP(pts, vals, k); });
}

// K_chol (lower triangle meaningful, N*N col-major), K_inv_y (N), mean.
void ref_gp_dump(void* hv, double* K_chol, double* K_inv_y, double* mean) {
  GaussianProcess* gp = static_cast<RefGP*>(hv)->gp;
  const int N = gp->num_sampled() * (1 + gp->num_derivatives());
  if (K_chol) std::copy(gp->K_chol_.begin(), gp->K_chol_.begin() + static_cast<size_t>(N) * N, K_chol);
  if (K_inv_y) std::copy(gp->get_K_inv_y().begin(), gp->get_K_inv_y().end(), K_inv_y);
  if (mean) *mean = gp->get_mean();
}

// K(X, pts) with derivative blocks; out[N x k(1+g2)] col-major (gpp_math.cpp:469-479).
void ref_gp_mix_cov(void* hv, const double* pts, int k, const int* d2, int g2, double* out) {
  static_cast<RefGP*>(hv)->gp->BuildMixCovarianceMatrix(pts, k, nn(d2), g2, out);
}

// The following mirror the Python-visible queries one for one (gpp_python_gaussian_process.cpp:64-236), but return the raw
// column-major arrays (no symmetrisation / transposition; that is boundary formatting and is tested separately).
int ref_gp_mean(void* hv, const double* pts, int k, double* out) {
  return guarded([&] {
    GaussianProcess* gp = static_cast<RefGP*>(hv)->gp;
    GaussianProcess::StateType st(*gp, pts, k, &dummy_int, 0, 0);
    gp->ComputeMeanOfPoints(st, out);
  });
}

int ref_gp_additional_mean(void* hv, const double* pts, int k, const int* d2, int g2, double* out) {
  return guarded([&] { static_cast<RefGP*>(hv)->gp->ComputeMeanOfAdditionalPoints(pts, k, nn(d2), g2, out); });
}

// out[d * k * (1+g)]
int ref_gp_grad_mean(void* hv, const double* pts, int k, double* out) {
  return guarded([&] {
    GaussianProcess* gp = static_cast<RefGP*>(hv)->gp;
    GaussianProcess::StateType st(*gp, pts, k, nn(gp->derivatives().data()), gp->num_derivatives(), k);
    gp->ComputeGradMeanOfPoints(st, out);
  });
}

// out[(k(1+g))^2] col-major, as ComputeVarianceOfPoints fills it.
int ref_gp_var(void* hv, const double* pts, int k, double* out) {
  return guarded([&] {
    GaussianProcess* gp = static_cast<RefGP*>(hv)->gp;
    GaussianProcess::StateType st(*gp, pts, k, nn(gp->derivatives().data()), gp->num_derivatives(), 0);
    gp->ComputeVarianceOfPoints(&st, nn(gp->derivatives().data()), gp->num_derivatives(), out);
  });
}

// out = chol(Var) in place (upper triangle holds leftovers, as in the reference).
int ref_gp_chol_var(void* hv, const double* pts, int k, double* out) {
  return guarded([&] {
    GaussianProcess* gp = static_cast<RefGP*>(hv)->gp;
    GaussianProcess::StateType st(*gp, pts, k, nn(gp->derivatives().data()), gp->num_derivatives(), 0);
    gp->ComputeVarianceOfPoints(&st, nn(gp->derivatives().data()), gp->num_derivatives(), out);
    const int m = k * (1 + gp->num_derivatives());
    int lm = ComputeCholeskyFactorL(m, out);
    if (lm != 0) {
      OL_THROW_EXCEPTION(SingularMatrixException, "GP-Variance matrix singular.", out, m, lm);
    }
  });
}

// out[d * m^2 * nd]
int ref_gp_grad_var(void* hv, const double* pts, int k, int nd, double* out) {
  return guarded([&] {
    GaussianProcess* gp = static_cast<RefGP*>(hv)->gp;
    GaussianProcess::StateType st(*gp, pts, k, nn(gp->derivatives().data()), gp->num_derivatives(), nd);
    gp->ComputeGradVarianceOfPoints(&st, out);
  });
}

int ref_gp_grad_chol_var(void* hv, const double* pts, int k, int nd, double* out) {
  return guarded([&] {
    GaussianProcess* gp = static_cast<RefGP*>(hv)->gp;
    const int m = k * (1 + gp->num_derivatives());
    std::vector<double> chol(static_cast<size_t>(m) * m);
    GaussianProcess::StateType st(*gp, pts, k, nn(gp->derivatives().data()), gp->num_derivatives(), nd);
    gp->ComputeVarianceOfPoints(&st, nn(gp->derivatives().data()), gp->num_derivatives(), chol.data());
    int lm = ComputeCholeskyFactorL(m, chol.data());
    if (lm != 0) {
      OL_THROW_EXCEPTION(SingularMatrixException, "GP-Variance matrix singular.", chol.data(), m, lm);
    }
    gp->ComputeGradCholeskyVarianceOfPoints(&st, chol.data(), out);
  });
}

// ---- posterior mean objective (gpp_knowledge_gradient_optimization.cpp:322-351) ----
int ref_posterior_mean(void* hv, int num_fidelity, const double* pt, double* value, double* grad) {
  return guarded([&] {
    GaussianProcess* gp = static_cast<RefGP*>(hv)->gp;
    PosteriorMeanEvaluator ev(*gp);
    PosteriorMeanEvaluator::StateType st(ev, num_fidelity, pt, true);
    if (value) *value = ev.ComputePosteriorMean(&st);
    if (grad) ev.ComputeGradPosteriorMean(&st, grad);
  });
}

// ---- q,p-EI by Monte Carlo (gpp_math.cpp:1991-2126); normals table [M][q+p] ----
int ref_ei(void* hv, const double* Xq, const double* Xp, int q, int p, int M, double best_so_far, const double* normals,
           double* ei, double* grad, double* seconds) {
  return guarded([&] {
    GaussianProcess* gp = static_cast<RefGP*>(hv)->gp;
    std::vector<double> table(normals, normals + static_cast<size_t>(M) * (q + p));
    NormalRNGSimulator rng(table);
    ExpectedImprovementEvaluator ev(*gp, M, best_so_far);
    double dummy = 0.0;
    auto t0 = std::chrono::steady_clock::now();
    ExpectedImprovementEvaluator::StateType st(ev, Xq, p > 0 ? Xp : &dummy, q, p, grad != nullptr, &rng);
    if (ei) *ei = ev.ComputeExpectedImprovement(&st);
    if (grad) ev.ComputeGradExpectedImprovement(&st, grad);
    auto t1 = std::chrono::steady_clock::now();
    if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
  });
}

// ---- analytic 1,0-EI (OnePotentialSampleExpectedImprovementEvaluator, gpp_math.cpp:2195-2259) ----
int ref_ei_analytic(void* hv, const double* pt, double best_so_far, double* ei, double* grad) {
  return guarded([&] {
    GaussianProcess* gp = static_cast<RefGP*>(hv)->gp;
    OnePotentialSampleExpectedImprovementEvaluator ev(*gp, best_so_far);
    OnePotentialSampleExpectedImprovementEvaluator::StateType st(ev, pt, grad != nullptr);
    if (ei) *ei = ev.ComputeExpectedImprovement(&st);
    if (grad) ev.ComputeGradExpectedImprovement(&st, grad);
  });
}

// ---- 1,0-EI multistart gradient descent from a given start set (ComputeOptimalPointsToSampleViaMultistartGradientDescent,
// gpp_math.hpp:1683-1742).  q = 1, p = 0 takes the analytic evaluator, so no random source is involved and the result is a
// deterministic function of the inputs.  num_starts must be >= 20 (the reference pops its top-20 queue unconditionally). ----
// domain_type: 0 = TensorProductDomain, 1 = SimplexIntersectTensorProductDomain (the dispatch of gpp_python_expected_improvement.cpp:262-271)
int ref_ei_multistart_analytic_dom(void* hv, const double* gd, const double* bounds, const double* starts, int num_starts,
                                   double best_so_far, int domain_type, int* found, double* best_point) {
  return guarded([&] {
    GaussianProcess* gp = static_cast<RefGP*>(hv)->gp;
    const int d = gp->dim();
    std::vector<ClosedInterval> iv(d);
    for (int i = 0; i < d; ++i) iv[i] = ClosedInterval(bounds[2 * i], bounds[2 * i + 1]);
    GradientDescentParameters gdp(static_cast<int>(gd[0]), static_cast<int>(gd[1]), static_cast<int>(gd[2]),
                                  static_cast<int>(gd[3]), gd[4], gd[5], gd[6], gd[7]);
    ThreadSchedule sched(1, omp_sched_static);
    NormalRNG rng(1);
    double dummy = 0.0;
    bool found_flag = false;
    if (domain_type == 1) {
      SimplexIntersectTensorProductDomain dom(iv.data(), d);
      ComputeOptimalPointsToSampleViaMultistartGradientDescent(*gp, gdp, dom, sched, starts, &dummy, num_starts, 1, 0,
                                                               best_so_far, 1, &rng, &found_flag, best_point);
    } else {
      TensorProductDomain dom(iv.data(), d);
      ComputeOptimalPointsToSampleViaMultistartGradientDescent(*gp, gdp, dom, sched, starts, &dummy, num_starts, 1, 0,
                                                               best_so_far, 1, &rng, &found_flag, best_point);
    }
    *found = found_flag ? 1 : 0;
  });
}

int ref_ei_multistart_analytic(void* hv, const double* gd, const double* bounds, const double* starts, int num_starts,
                               double best_so_far, int* found, double* best_point) {
  return ref_ei_multistart_analytic_dom(hv, gd, bounds, starts, num_starts, best_so_far, 0, found, best_point);
}

// ---- q-KG / d-KG (gpp_knowledge_gradient_optimization.cpp:69-227) ----
// gd = {num_multistarts, max_num_steps, max_num_restarts, num_steps_averaged, gamma, pre_mult, max_relative_change, tolerance}
// bounds = [min0,max0,...] over dim - num_fidelity coordinates; discrete[P][dim-num_fidelity];
// normals: table of ceil(M/2)*m values, m = (q+p)(1+g) (only even samples draw; odd ones are antithetic, .cpp:171-180).
// Optional outputs (may be NULL): grad[q*d]; best_point[M*d]; to_sample_mean[m]; chol_var[m*m]; grad_chol[d*m*m*q];
// chol_inverse_cov[m*M]; seconds[2] = {state construction, evaluation}.
}  // extern "C"

namespace {
// ref_kg's body for either inner domain (DomainTypes::kTensorProduct / kSimplex, gpp_python_knowledge_gradient.cpp:288-296: the
// inner optimisations of a KG evaluation run over the SAME domain type as the outer one)
template <typename DomainT>
void kg_body(GaussianProcess* gp, int num_fidelity, const double* gd, const double* bounds, const double* discrete, int P,
             const double* Xq, const double* Xp, int q, int p, int M, double best_so_far, const double* normals,
             long num_normals, int want_grad, double* kg, double* grad, double* best_point, double* to_sample_mean,
             double* chol_var, double* grad_chol, double* chol_inverse_cov, double* seconds) {
  const int d = gp->dim();
  std::vector<ClosedInterval> iv(d - num_fidelity);
  for (int i = 0; i < d - num_fidelity; ++i) iv[i] = ClosedInterval(bounds[2 * i], bounds[2 * i + 1]);
  DomainT dom(iv.data(), d - num_fidelity);
  GradientDescentParameters gdp(static_cast<int>(gd[0]), static_cast<int>(gd[1]), static_cast<int>(gd[2]),
                                static_cast<int>(gd[3]), gd[4], gd[5], gd[6], gd[7]);
  std::vector<double> table(normals, normals + num_normals);
  NormalRNGSimulator rng(table);
  double dummy = 0.0;
  auto t0 = std::chrono::steady_clock::now();
  KnowledgeGradientEvaluator<DomainT> ev(*gp, num_fidelity, discrete, P, M, dom, gdp, best_so_far);
  typename KnowledgeGradientEvaluator<DomainT>::StateType st(ev, Xq, p > 0 ? Xp : &dummy, q, p, P, nn(gp->derivatives().data()),
                                                            gp->num_derivatives(), want_grad != 0, &rng);
  auto t1 = std::chrono::steady_clock::now();
  double val;
  if (want_grad) {
    std::vector<double> gtmp(static_cast<size_t>(q) * d);
    val = ev.ComputeGradKnowledgeGradient(&st, gtmp.data());
    if (grad) std::copy(gtmp.begin(), gtmp.end(), grad);
  } else {
    val = ev.ComputeKnowledgeGradient(&st);
  }
  auto t2 = std::chrono::steady_clock::now();
  if (kg) *kg = val;
  if (best_point) std::copy(st.best_point.begin(), st.best_point.end(), best_point);
  if (to_sample_mean) std::copy(st.to_sample_mean_.begin(), st.to_sample_mean_.end(), to_sample_mean);
  if (chol_var) std::copy(st.cholesky_to_sample_var.begin(), st.cholesky_to_sample_var.end(), chol_var);
  if (grad_chol && want_grad) std::copy(st.grad_chol_decomp.begin(), st.grad_chol_decomp.end(), grad_chol);
  if (chol_inverse_cov && want_grad) std::copy(st.chol_inverse_cov.begin(), st.chol_inverse_cov.end(), chol_inverse_cov);
  if (seconds) {
    seconds[0] = std::chrono::duration<double>(t1 - t0).count();
    seconds[1] = std::chrono::duration<double>(t2 - t1).count();
  }
}
}  // namespace

extern "C" {
int ref_kg_dom(void* hv, int num_fidelity, const double* gd, const double* bounds, const double* discrete, int P,
               const double* Xq, const double* Xp, int q, int p, int M, double best_so_far, const double* normals,
               long num_normals, int want_grad, int domain_type, double* kg, double* grad, double* best_point,
               double* to_sample_mean, double* chol_var, double* grad_chol, double* chol_inverse_cov, double* seconds) {
  return guarded([&] {
    GaussianProcess* gp = static_cast<RefGP*>(hv)->gp;
    if (domain_type == 1)
      kg_body<SimplexIntersectTensorProductDomain>(gp, num_fidelity, gd, bounds, discrete, P, Xq, Xp, q, p, M, best_so_far, normals,
                                                   num_normals, want_grad, kg, grad, best_point, to_sample_mean, chol_var,
                                                   grad_chol, chol_inverse_cov, seconds);
    else
      kg_body<TensorProductDomain>(gp, num_fidelity, gd, bounds, discrete, P, Xq, Xp, q, p, M, best_so_far, normals, num_normals,
                                   want_grad, kg, grad, best_point, to_sample_mean, chol_var, grad_chol, chol_inverse_cov, seconds);
  });
}

int ref_kg(void* hv, int num_fidelity, const double* gd, const double* bounds, const double* discrete, int P,
           const double* Xq, const double* Xp, int q, int p, int M, double best_so_far, const double* normals,
           long num_normals, int want_grad, double* kg, double* grad, double* best_point, double* to_sample_mean,
           double* chol_var, double* grad_chol, double* chol_inverse_cov, double* seconds) {
  return ref_kg_dom(hv, num_fidelity, gd, bounds, discrete, P, Xq, Xp, q, p, M, best_so_far, normals, num_normals, want_grad, 0, kg,
                    grad, best_point, to_sample_mean, chol_var, grad_chol, chol_inverse_cov, seconds);
}

// ---- MCMC-averaged evaluators: GaussianProcessMCMC (gpp_knowledge_gradient_mcmc_optimization.cpp:24-49),
// KnowledgeGradientMCMCEvaluator (:51-180), ExpectedImprovementMCMCEvaluator (gpp_expected_improvement_mcmc_optimization.cpp) ----
void* ref_gpmcmc_create(const double* hypers, const double* noises, int num_mcmc, const double* X, const double* y,
                        const int* derivs, int g, int d, int n) {
  GaussianProcessMCMC* h = nullptr;
  const int rc = guarded([&] { h = new GaussianProcessMCMC(hypers, noises, num_mcmc, X, y, nn(derivs), g, d, n); });
  return rc == 0 ? h : nullptr;
}

void ref_gpmcmc_destroy(void* hv) { delete static_cast<GaussianProcessMCMC*>(hv); }

// discrete_all[num_mcmc][P][d - num_fidelity]; best_so_far[num_mcmc]; normals as for ref_kg (every GP replays the same
// stream: each per-GP evaluator rewinds the shared RNG, gpp_knowledge_gradient_optimization.cpp:78, 139).
int ref_kg_mcmc(void* hv, int num_fidelity, const double* gd, const double* bounds, const double* discrete_all, int P,
                const double* Xq, const double* Xp, int q, int p, int M, const double* best_so_far, const double* normals,
                long num_normals, int want_grad, double* kg, double* grad) {
  return guarded([&] {
    GaussianProcessMCMC* gpm = static_cast<GaussianProcessMCMC*>(hv);
    const int d = gpm->dim();
    std::vector<ClosedInterval> iv(d - num_fidelity);
    for (int i = 0; i < d - num_fidelity; ++i) iv[i] = ClosedInterval(bounds[2 * i], bounds[2 * i + 1]);
    TensorProductDomain dom(iv.data(), d - num_fidelity);
    GradientDescentParameters gdp(static_cast<int>(gd[0]), static_cast<int>(gd[1]), static_cast<int>(gd[2]),
                                  static_cast<int>(gd[3]), gd[4], gd[5], gd[6], gd[7]);
    std::vector<double> table(normals, normals + num_normals);
    NormalRNGSimulator rng(table);
    double dummy = 0.0;
    std::vector<KnowledgeGradientState<TensorProductDomain>::EvaluatorType> evaluators;
    KnowledgeGradientMCMCEvaluator<TensorProductDomain> ev(*gpm, num_fidelity, discrete_all, P, M, dom, gdp, best_so_far,
                                                           &evaluators);
    std::vector<KnowledgeGradientEvaluator<TensorProductDomain>::StateType> states;
    KnowledgeGradientMCMCEvaluator<TensorProductDomain>::StateType st(ev, Xq, p > 0 ? Xp : &dummy, q, p, P,
                                                                      nn(gpm->derivatives().data()), gpm->num_derivatives(),
                                                                      want_grad != 0, &rng, &states);
    if (kg) *kg = ev.ComputeKnowledgeGradient(&st);
    if (want_grad && grad) {
      std::fill(grad, grad + static_cast<size_t>(q) * d, 0.0);  // the evaluator accumulates into its output (.cpp:163-166)
      ev.ComputeGradKnowledgeGradient(&st, grad);
    }
  });
}

int ref_ei_mcmc(void* hv, const double* Xq, const double* Xp, int q, int p, int M, const double* best_so_far,
                const double* normals, double* ei, double* grad) {
  return guarded([&] {
    GaussianProcessMCMC* gpm = static_cast<GaussianProcessMCMC*>(hv);
    const int d = gpm->dim();
    std::vector<double> table(normals, normals + static_cast<size_t>(M) * (q + p));
    NormalRNGSimulator rng(table);
    double dummy = 0.0;
    std::vector<ExpectedImprovementState::EvaluatorType> evaluators;
    ExpectedImprovementMCMCEvaluator ev(*gpm, M, best_so_far, &evaluators);
    std::vector<ExpectedImprovementEvaluator::StateType> states;
    ExpectedImprovementMCMCEvaluator::StateType st(ev, Xq, p > 0 ? Xp : &dummy, q, p, nn(gpm->derivatives().data()),
                                                   gpm->num_derivatives(), grad != nullptr, &rng, &states);
    if (ei) *ei = ev.ComputeExpectedImprovement(&st);
    if (grad) {
      std::fill(grad, grad + static_cast<size_t>(q) * d, 0.0);
      ev.ComputeGradExpectedImprovement(&st, grad);
    }
  });
}

// 1,0-EI multistart gradient descent on the MCMC-averaged analytic EI from a given start set
// (ComputeEIMCMCOptimalPointsToSampleViaMultistartGradientDescent, gpp_expected_improvement_mcmc_optimization.hpp:850-925):
// deterministic (no random source involved).  num_starts must be >= 20.
int ref_ei_mcmc_multistart_analytic(void* hv, const double* gd, const double* bounds, const double* starts, int num_starts,
                                    const double* best_so_far, int* found, double* best_point) {
  return guarded([&] {
    GaussianProcessMCMC* gpm = static_cast<GaussianProcessMCMC*>(hv);
    const int d = gpm->dim();
    std::vector<ClosedInterval> iv(d);
    for (int i = 0; i < d; ++i) iv[i] = ClosedInterval(bounds[2 * i], bounds[2 * i + 1]);
    TensorProductDomain dom(iv.data(), d);
    GradientDescentParameters gdp(static_cast<int>(gd[0]), static_cast<int>(gd[1]), static_cast<int>(gd[2]),
                                  static_cast<int>(gd[3]), gd[4], gd[5], gd[6], gd[7]);
    ThreadSchedule sched(1, omp_sched_static);
    NormalRNG rng(1);
    double dummy = 0.0;
    bool found_flag = false;
    ComputeEIMCMCOptimalPointsToSampleViaMultistartGradientDescent(*gpm, gdp, dom, sched, starts, &dummy, num_starts, 1, 0,
                                                                   best_so_far, 1, &rng, &found_flag, best_point);
    *found = found_flag ? 1 : 0;
  });
}

// ---- log marginal likelihood (LogMarginalLikelihoodEvaluator::ComputeLogLikelihood, gpp_model_selection.cpp:540-612), driven
// like ComputeLogLikelihoodWrapper (gpp_python_model_selection.cpp:43-69): Matern-5/2 unless cov_type == 0 ----
int ref_log_likelihood(int cov_type, double alpha, const double* lengths, const double* X, const double* y, const double* noise,
                       const int* derivs, int g, int d, int n, double* value) {
  return guarded([&] {
    CovarianceInterface* cov = make_cov(cov_type, d, alpha, lengths);
    LogMarginalLikelihoodEvaluator ev(X, y, nn(derivs), g, d, n);
    std::vector<double> nv(noise, noise + 1 + g);
    LogMarginalLikelihoodState st(ev, *cov, nv);
    *value = ev.ComputeLogLikelihood(st);
    delete cov;
  });
}

// LogMarginalLikelihoodEvaluator::ComputeGradLogLikelihood (gpp_model_selection.cpp:629-677) driven like
// ComputeHyperparameterGradLogLikelihoodWrapper (gpp_python_model_selection.cpp:88-135): grad[1 + d + 1 + g].
int ref_log_likelihood_grad(int cov_type, double alpha, const double* lengths, const double* X, const double* y,
                            const double* noise, const int* derivs, int g, int d, int n, double* grad) {
  return guarded([&] {
    CovarianceInterface* cov = make_cov(cov_type, d, alpha, lengths);
    LogMarginalLikelihoodEvaluator ev(X, y, nn(derivs), g, d, n);
    std::vector<double> nv(noise, noise + 1 + g);
    LogMarginalLikelihoodState st(ev, *cov, nv);
    ev.ComputeGradLogLikelihood(&st, grad);
    delete cov;
  });
}

// MultistartGradientDescentHyperparameterOptimization (gpp_model_selection.hpp:1063-1103) from CALLER-SUPPLIED initial guesses
// (linear space): the function's own body -- linear-space domain, SetupLogLikelihoodState, InitializeBestKnownPoint,
// MultistartOptimizer over GradientDescentOptimizer -- minus the Latin-hypercube draw, whose uniform stream is not pinned across
// Boost versions.  hyper0 = (alpha, lengths) of the covariance object the states are built from (the reference's wrapper passes the
// caller's current hyper-parameters, gpp_python_model_selection.cpp:176-260); domain_log10[nh][2]; best[nh].
int ref_ll_multistart(int cov_type, double alpha, const double* lengths, const double* X, const double* y, const double* noise,
                      const int* derivs, int g, int d, int n, const double* gd, const double* domain_log10,
                      const double* initial_guesses, int num_starts, int num_threads, int* found, double* best, double* best_value) {
  return guarded([&] {
    CovarianceInterface* cov = make_cov(cov_type, d, alpha, lengths);
    LogMarginalLikelihoodEvaluator ev(X, y, nn(derivs), g, d, n);
    std::vector<double> nv(noise, noise + 1 + g);
    GradientDescentParameters gdp(static_cast<int>(gd[0]), static_cast<int>(gd[1]), static_cast<int>(gd[2]), static_cast<int>(gd[3]),
                                  gd[4], gd[5], gd[6], gd[7]);
    const int nh = cov->GetNumberOfHyperparameters() + 1 + g;
    std::vector<ClosedInterval> dom(nh);
    for (int i = 0; i < nh; ++i) dom[i] = ClosedInterval(std::pow(10.0, domain_log10[2 * i]), std::pow(10.0, domain_log10[2 * i + 1]));
    TensorProductDomain domain_linearspace(dom.data(), nh);
    ThreadSchedule sched(num_threads > 0 ? num_threads : 1, omp_sched_static);
    std::vector<LogMarginalLikelihoodState> states;
    SetupLogLikelihoodState(ev, *cov, nv, sched.max_num_threads, &states);
    OptimizationIOContainer io(states[0].GetProblemSize());
    InitializeBestKnownPoint(ev, initial_guesses, nh, num_starts, states.data(), &io);
    GradientDescentOptimizer<LogMarginalLikelihoodEvaluator, TensorProductDomain> gd_opt;
    MultistartOptimizer<GradientDescentOptimizer<LogMarginalLikelihoodEvaluator, TensorProductDomain> > ms;
    ms.MultistartOptimize(gd_opt, ev, gdp, domain_linearspace, sched, initial_guesses, num_starts, states.data(), nullptr, &io);
    *found = io.found_flag ? 1 : 0;
    std::copy(io.best_point.begin(), io.best_point.end(), best);
    if (best_value) *best_value = io.best_objective_value_so_far;
    delete cov;
  });
}

// All-core CPU baseline the way the reference parallelises: independent evaluations under OpenMP, one State + RNG per
// thread (gpp_optimization.hpp:1472-1546). Xq_all[R][q][d]; kg_out[R]; grad_out[R][q*d]. Returns wall seconds.
int ref_kg_grad_batch(void* hv, int num_fidelity, const double* gd, const double* bounds, const double* discrete, int P,
                      const double* Xq_all, int R, int q, int M, double best_so_far, const double* normals,
                      long num_normals, int num_threads, double* kg_out, double* grad_out, double* wall_seconds) {
  int rc_all = 0;
  GaussianProcess* gp = static_cast<RefGP*>(hv)->gp;
  const int d = gp->dim();
  auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel for num_threads(num_threads) schedule(dynamic, 1)
  for (int r = 0; r < R; ++r) {
    double kgv = 0.0;
    int rc = ref_kg(hv, num_fidelity, gd, bounds, discrete, P, Xq_all + static_cast<size_t>(r) * q * d, nullptr, q, 0, M,
                    best_so_far, normals, num_normals, 1, &kgv, grad_out + static_cast<size_t>(r) * q * d, nullptr,
                    nullptr, nullptr, nullptr, nullptr, nullptr);
    kg_out[r] = kgv;
    if (rc != 0) {
#pragma omp critical
      rc_all = rc;
    }
  }
  auto t1 = std::chrono::steady_clock::now();
  if (wall_seconds) *wall_seconds = std::chrono::duration<double>(t1 - t0).count();
  return rc_all;
}

// ---- the KG outer optimiser: ComputeKGOptimalPointsToSampleViaMultistartGradientDescent
// (gpp_knowledge_gradient_optimization.hpp:860-935) and its MCMC twin (gpp_knowledge_gradient_mcmc_optimization.hpp:665-760).
// Both take a concrete NormalRNG* (not the interface), so explicit tables cannot be injected; instead the stream of
// NormalRNG(seed) is exported (ref_normal_draws) and replayed by the device driver as its explicit table: every KG evaluation
// rewinds the generator (gpp_knowledge_gradient_optimization.cpp:78, 139; ResetToMostRecentSeed re-seeds the engine AND resets
// the distribution, gpp_random.cpp:101-115, 156-158), so each evaluation consumes the first ceil(M/2)*m draws of that stream.
// One thread (thread_schedule.max_num_threads = 1): deterministic. ----
int ref_normal_draws(unsigned int seed, long count, double* out) {
  return guarded([&] {
    NormalRNG rng(seed);
    for (long i = 0; i < count; ++i) out[i] = rng();
  });
}

// KG value through a seeded NormalRNG (the route the multistart drivers take), to check the exported stream against the
// NormalRNGSimulator route of ref_kg.
int ref_kg_seeded(void* hv, int num_fidelity, const double* gd, const double* bounds, const double* discrete, int P,
                  const double* Xq, const double* Xp, int q, int p, int M, double best_so_far, unsigned int seed, double* kg) {
  return guarded([&] {
    GaussianProcess* gp = static_cast<RefGP*>(hv)->gp;
    const int d = gp->dim();
    std::vector<ClosedInterval> iv(d - num_fidelity);
    for (int i = 0; i < d - num_fidelity; ++i) iv[i] = ClosedInterval(bounds[2 * i], bounds[2 * i + 1]);
    TensorProductDomain dom(iv.data(), d - num_fidelity);
    GradientDescentParameters gdp(static_cast<int>(gd[0]), static_cast<int>(gd[1]), static_cast<int>(gd[2]),
                                  static_cast<int>(gd[3]), gd[4], gd[5], gd[6], gd[7]);
    NormalRNG rng(seed);
    double dummy = 0.0;
    KnowledgeGradientEvaluator<TensorProductDomain> ev(*gp, num_fidelity, discrete, P, M, dom, gdp, best_so_far);
    KnowledgeGradientEvaluator<TensorProductDomain>::StateType st(
        ev, Xq, p > 0 ? Xp : &dummy, q, p, P, nn(gp->derivatives().data()), gp->num_derivatives(), false, &rng);
    *kg = ev.ComputeKnowledgeGradient(&st);
    *kg = ev.ComputeKnowledgeGradient(&st);  // twice: the second call must replay the same draws
  });
}

// gd_outer / gd_inner as in ref_kg; bounds[2*d] (outer domain, all d coordinates), inner_bounds[2*(d-num_fidelity)];
// starts[num_starts][q][d] (num_starts >= 20: the reference pops its 20-deep queue unconditionally); best_point[q][d].
}  // extern "C"

namespace {
template <typename DomainT>
void kg_multistart_body(GaussianProcess* gp, int num_fidelity, const double* gd_outer, const double* gd_inner, const double* bounds,
                        const double* inner_bounds, const double* discrete, int P, const double* starts, int num_starts,
                        const double* Xp, int q, int p, int M, double best_so_far, unsigned int seed, int* found,
                        double* best_point) {
  const int d = gp->dim();
  std::vector<ClosedInterval> iv(d), ivi(d - num_fidelity);
  for (int i = 0; i < d; ++i) iv[i] = ClosedInterval(bounds[2 * i], bounds[2 * i + 1]);
  for (int i = 0; i < d - num_fidelity; ++i) ivi[i] = ClosedInterval(inner_bounds[2 * i], inner_bounds[2 * i + 1]);
  DomainT dom(iv.data(), d), inner_dom(ivi.data(), d - num_fidelity);
  GradientDescentParameters gdo(static_cast<int>(gd_outer[0]), static_cast<int>(gd_outer[1]), static_cast<int>(gd_outer[2]),
                                static_cast<int>(gd_outer[3]), gd_outer[4], gd_outer[5], gd_outer[6], gd_outer[7]);
  GradientDescentParameters gdi(static_cast<int>(gd_inner[0]), static_cast<int>(gd_inner[1]), static_cast<int>(gd_inner[2]),
                                static_cast<int>(gd_inner[3]), gd_inner[4], gd_inner[5], gd_inner[6], gd_inner[7]);
  ThreadSchedule sched(1, omp_sched_static);
  NormalRNG rng(seed);
  double dummy = 0.0;
  bool found_flag = false;
  ComputeKGOptimalPointsToSampleViaMultistartGradientDescent(*gp, num_fidelity, gdo, gdi, dom, inner_dom, sched, starts,
                                                             p > 0 ? Xp : &dummy, discrete, num_starts, q, p, P, best_so_far, M,
                                                             &rng, &found_flag, best_point);
  *found = found_flag ? 1 : 0;
}
}  // namespace

extern "C" {
int ref_kg_multistart_dom(void* hv, int num_fidelity, const double* gd_outer, const double* gd_inner, const double* bounds,
                          const double* inner_bounds, const double* discrete, int P, const double* starts, int num_starts,
                          const double* Xp, int q, int p, int M, double best_so_far, unsigned int seed, int domain_type, int* found,
                          double* best_point) {
  return guarded([&] {
    GaussianProcess* gp = static_cast<RefGP*>(hv)->gp;
    if (domain_type == 1)
      kg_multistart_body<SimplexIntersectTensorProductDomain>(gp, num_fidelity, gd_outer, gd_inner, bounds, inner_bounds, discrete, P,
                                                              starts, num_starts, Xp, q, p, M, best_so_far, seed, found, best_point);
    else
      kg_multistart_body<TensorProductDomain>(gp, num_fidelity, gd_outer, gd_inner, bounds, inner_bounds, discrete, P, starts,
                                              num_starts, Xp, q, p, M, best_so_far, seed, found, best_point);
  });
}

int ref_kg_multistart(void* hv, int num_fidelity, const double* gd_outer, const double* gd_inner, const double* bounds,
                      const double* inner_bounds, const double* discrete, int P, const double* starts, int num_starts,
                      const double* Xp, int q, int p, int M, double best_so_far, unsigned int seed, int* found,
                      double* best_point) {
  return ref_kg_multistart_dom(hv, num_fidelity, gd_outer, gd_inner, bounds, inner_bounds, discrete, P, starts, num_starts, Xp, q, p,
                               M, best_so_far, seed, 0, found, best_point);
}

// MCMC twin: discrete_all[num_mcmc][P][d - num_fidelity], best_so_far[num_mcmc].
int ref_kg_mcmc_multistart(void* hv, int num_fidelity, const double* gd_outer, const double* gd_inner, const double* bounds,
                           const double* inner_bounds, const double* discrete_all, int P, const double* starts, int num_starts,
                           const double* Xp, int q, int p, int M, const double* best_so_far, unsigned int seed, int* found,
                           double* best_point) {
  return guarded([&] {
    GaussianProcessMCMC* gpm = static_cast<GaussianProcessMCMC*>(hv);
    const int d = gpm->dim();
    std::vector<ClosedInterval> iv(d), ivi(d - num_fidelity);
    for (int i = 0; i < d; ++i) iv[i] = ClosedInterval(bounds[2 * i], bounds[2 * i + 1]);
    for (int i = 0; i < d - num_fidelity; ++i) ivi[i] = ClosedInterval(inner_bounds[2 * i], inner_bounds[2 * i + 1]);
    TensorProductDomain dom(iv.data(), d), inner_dom(ivi.data(), d - num_fidelity);
    GradientDescentParameters gdo(static_cast<int>(gd_outer[0]), static_cast<int>(gd_outer[1]), static_cast<int>(gd_outer[2]),
                                  static_cast<int>(gd_outer[3]), gd_outer[4], gd_outer[5], gd_outer[6], gd_outer[7]);
    GradientDescentParameters gdi(static_cast<int>(gd_inner[0]), static_cast<int>(gd_inner[1]), static_cast<int>(gd_inner[2]),
                                  static_cast<int>(gd_inner[3]), gd_inner[4], gd_inner[5], gd_inner[6], gd_inner[7]);
    ThreadSchedule sched(1, omp_sched_static);
    NormalRNG rng(seed);
    double dummy = 0.0;
    bool found_flag = false;
    ComputeKGMCMCOptimalPointsToSampleViaMultistartGradientDescent(*gpm, num_fidelity, gdo, gdi, dom, inner_dom, sched, starts,
                                                                   p > 0 ? Xp : &dummy, discrete_all, num_starts, q, p, P,
                                                                   best_so_far, M, &rng, &found_flag, best_point);
    *found = found_flag ? 1 : 0;
  });
}

// The same with `num_threads` OpenMP threads, the way the reference's own Python drivers call it (examples/bayesian_optimization.py:
// max_num_threads = 20): one NormalRNG per thread, all seeded alike (every evaluation rewinds its generator, so the stream a restart
// sees does not depend on the thread that runs it).  *wall_s = the optimiser's wall time.  bench.py --config suggest.
int ref_kg_mcmc_multistart_mt(void* hv, int num_fidelity, const double* gd_outer, const double* gd_inner, const double* bounds,
                              const double* inner_bounds, const double* discrete_all, int P, const double* starts, int num_starts,
                              const double* Xp, int q, int p, int M, const double* best_so_far, unsigned int seed, int num_threads,
                              int* found, double* best_point, double* wall_s) {
  return guarded([&] {
    GaussianProcessMCMC* gpm = static_cast<GaussianProcessMCMC*>(hv);
    const int d = gpm->dim();
    std::vector<ClosedInterval> iv(d), ivi(d - num_fidelity);
    for (int i = 0; i < d; ++i) iv[i] = ClosedInterval(bounds[2 * i], bounds[2 * i + 1]);
    for (int i = 0; i < d - num_fidelity; ++i) ivi[i] = ClosedInterval(inner_bounds[2 * i], inner_bounds[2 * i + 1]);
    TensorProductDomain dom(iv.data(), d), inner_dom(ivi.data(), d - num_fidelity);
    GradientDescentParameters gdo(static_cast<int>(gd_outer[0]), static_cast<int>(gd_outer[1]), static_cast<int>(gd_outer[2]),
                                  static_cast<int>(gd_outer[3]), gd_outer[4], gd_outer[5], gd_outer[6], gd_outer[7]);
    GradientDescentParameters gdi(static_cast<int>(gd_inner[0]), static_cast<int>(gd_inner[1]), static_cast<int>(gd_inner[2]),
                                  static_cast<int>(gd_inner[3]), gd_inner[4], gd_inner[5], gd_inner[6], gd_inner[7]);
    const int T = num_threads > 0 ? num_threads : 1;
    ThreadSchedule sched(T, omp_sched_dynamic);  // (gpp_python_knowledge_gradient_mcmc.cpp: omp_sched_dynamic, chunk as the default)
    std::vector<NormalRNG> rngs(T, NormalRNG(seed));
    double dummy = 0.0;
    bool found_flag = false;
    const double t0 = omp_get_wtime();
    ComputeKGMCMCOptimalPointsToSampleViaMultistartGradientDescent(*gpm, num_fidelity, gdo, gdi, dom, inner_dom, sched, starts,
                                                                   p > 0 ? Xp : &dummy, discrete_all, num_starts, q, p, P,
                                                                   best_so_far, M, rngs.data(), &found_flag, best_point);
    *wall_s = omp_get_wtime() - t0;
    *found = found_flag ? 1 : 0;
  });
}

int ref_num_procs() { return omp_get_num_procs(); }

}  // extern "C"
