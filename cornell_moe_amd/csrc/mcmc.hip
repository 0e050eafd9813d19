// cornell_moe_amd/csrc/mcmc.hip -- MCMC-averaged acquisition evaluators: the Bayesian treatment of the GP
// hyper-parameters the reference's examples actually run -- num_mcmc GPs over the same data, one per hyper-parameter sample
// (GaussianProcessMCMC, gpp_knowledge_gradient_mcmc_optimization.cpp:24-49), acquisition = the average of the per-GP
// acquisitions, KG additionally divided by the fidelity cost:
//   * KnowledgeGradientMCMCEvaluator::Compute[Grad]KnowledgeGradient, ComputeCost, ComputeGradCost (.cpp:84-180);
//   * ExpectedImprovementMCMCEvaluator / OnePotentialSampleExpectedImprovementMCMCEvaluator
//     (gpp_expected_improvement_mcmc_optimization.cpp:48-88, 136-176);
//   * ComputeKGMCMCOptimalPointsToSampleViaMultistartGradientDescent / EvaluateKGMCMCAtPointList
//     (gpp_knowledge_gradient_mcmc_optimization.hpp:665-862) and the EI twins (gpp_expected_improvement_mcmc_optimization.hpp).
// Every per-GP evaluator replays the same normal stream (each rewinds the shared RNG before it draws,
// gpp_knowledge_gradient_optimization.cpp:78, 139), so one table serves all GPs.  The GP index is a third independent shard
// axis: kg_mcmc_sums returns plain sums over the GPs it was given so ranks holding disjoint GP subsets can all-reduce them
// before kg_mcmc_finalize (cornell_moe_amd/dist.py).
#include <algorithm>
#include <cmath>

#include "gp.hpp"
#include "kg.hpp"

#include <cstdlib>

namespace moe {

namespace {

void check_ensemble(const std::vector<GpDev*>& gps) {
  if (gps.empty()) throw Error(MOE_ERR_BOUNDS, "num_mcmc must be positive", 0, 1, 1e9);
  for (const GpDev* g : gps) {
    if (g == nullptr) throw Error(MOE_ERR_RUNTIME, "NULL GP handle in the MCMC ensemble");
    if (g->d != gps[0]->d || g->g != gps[0]->g)
      throw Error(MOE_ERR_INVALID_VALUE, "MCMC ensemble members must share dim and the observed-derivative list", g->d, gps[0]->d, 0);
  }
}

}  // namespace

void kg_mcmc_sums(const std::vector<GpDev*>& gps, int num_fidelity, const moe_gd_params_t& inner, const double* bounds,
                  const double* discrete_all, int P, const double* Xq_all, int num_evals, const double* Xp, int q, int p,
                  int num_mc, const double* best_so_far, const double* normals, bool want_grad, double* kg_sum, double* grad_sum) {
  check_ensemble(gps);
  const int d = gps[0]->d, qd = q * d, E = num_evals;
  std::fill(kg_sum, kg_sum + E, 0.0);
  if (want_grad) std::fill(grad_sum, grad_sum + (size_t)E * qd, 0.0);
  std::vector<double> ks(E), gs(want_grad ? (size_t)E * qd : 0);
  const size_t disc_stride = (size_t)P * (d - num_fidelity);
  // launch every member (own stream, own workspaces), then collect: member i's kernels run while member i+1's state set-up
  // and host algebra are being prepared
  // (every member owns its workspaces: the batch each of them may carry is the budget divided by the ensemble size)
  const char* bg = std::getenv("MOE_KG_BATCH_GB");
  const double budget = ((bg && *bg) ? std::atof(bg) : 48.0) / (double)gps.size();
  const int max_e = kg_max_batch(*gps[0], P, q, p, num_mc, want_grad, budget);
  for (int e0 = 0; e0 < E; e0 += max_e) {
    const int ne = std::min(max_e, E - e0);
    std::vector<KgPending> pending;
    pending.reserve(gps.size());
    for (size_t i = 0; i < gps.size(); ++i)
      pending.push_back(kg_launch(*gps[i], num_fidelity, inner, bounds, discrete_all + i * disc_stride, P,
                                  Xq_all + (size_t)e0 * qd, ne, Xp, q, p, num_mc, best_so_far[i], normals, 0, num_mc, want_grad,
                                  false, budget));
    for (size_t i = 0; i < gps.size(); ++i) {
      pending[i].collect(ks.data(), want_grad ? gs.data() : nullptr, nullptr, nullptr);
      for (int e = 0; e < ne; ++e) kg_sum[e0 + e] += ks[e] / (double)num_mc;
      if (want_grad)
        for (size_t j = 0; j < (size_t)ne * qd; ++j) grad_sum[(size_t)e0 * qd + j] += gs[j] / (double)num_mc;
    }
  }
}

void kg_mcmc_finalize(double* kg, double* grad, const double* Xq_all, int num_evals, int q, int d, int num_fidelity, int num_mcmc) {
  const int qd = q * d;
  for (int e = 0; e < num_evals; ++e) {
    const double* x = Xq_all + (size_t)e * qd;
    double cost = 1.0;
    int index = -1;
    if (num_fidelity > 0) {  // ComputeCost (.cpp:84-102): the largest product of the fidelity coordinates over the q points
      cost = 0.0;
      for (int i = 0; i < q; ++i) {
        double pc = 1.0;
        for (int j = d - num_fidelity; j < d; ++j) pc *= x[i * d + j];
        if (cost < pc) {
          cost = pc;
          index = i;
        }
      }
    }
    const double mean_kg = kg[e] / (double)num_mcmc;
    kg[e] = mean_kg / cost;
    if (grad != nullptr) {
      double* g = grad + (size_t)e * qd;
      for (int k = 0; k < qd; ++k) {
        double gc = 0.0;  // ComputeGradCost (.cpp:107-127)
        if (num_fidelity > 0 && index >= 0 && k / d == index && k % d >= d - num_fidelity) gc = cost / x[k];
        g[k] = (g[k] / (double)num_mcmc * cost - mean_kg * gc) / (cost * cost);
      }
    }
  }
}

void ei_mcmc_batch(const std::vector<GpDev*>& gps, const double* Xq_all, int num_evals, const double* Xp, int q, int p, int num_mc,
                   const double* best_so_far, const double* normals, bool analytic, double* ei, double* grad_ei) {
  check_ensemble(gps);
  const int d = gps[0]->d, qd = q * d, E = num_evals;
  if (analytic && !(q == 1 && p == 0)) throw Error(MOE_ERR_RUNTIME, "analytic EI needs num_to_sample == 1 and num_being_sampled == 0");
  if (ei) std::fill(ei, ei + E, 0.0);
  if (grad_ei) std::fill(grad_ei, grad_ei + (size_t)E * qd, 0.0);
  std::vector<double> es(ei ? E : 0), gs(grad_ei ? (size_t)E * qd : 0);
  for (size_t i = 0; i < gps.size(); ++i) {
    if (analytic)
      ei_analytic_batch(*gps[i], Xq_all, E, best_so_far[i], ei ? es.data() : nullptr, grad_ei ? gs.data() : nullptr);
    else
      ei_evaluate_batch(*gps[i], Xq_all, E, Xp, q, p, num_mc, best_so_far[i], normals, ei ? es.data() : nullptr,
                        grad_ei ? gs.data() : nullptr);
    if (ei)
      for (int e = 0; e < E; ++e) ei[e] += es[e];
    if (grad_ei)
      for (size_t j = 0; j < (size_t)E * qd; ++j) grad_ei[j] += gs[j];
  }
  const double inv = 1.0 / (double)gps.size();
  if (ei)
    for (int e = 0; e < E; ++e) ei[e] *= inv;
  if (grad_ei)
    for (size_t j = 0; j < (size_t)E * qd; ++j) grad_ei[j] *= inv;
}

void kg_mcmc_multistart(const std::vector<GpDev*>& gps, int num_fidelity, const moe_gd_params_t& outer,
                        const moe_gd_params_t& inner, const double* bounds, const double* discrete_all, int P,
                        const double* starts, int num_starts, const double* Xp, int q, int p, int num_mc,
                        const double* best_so_far, const double* normals, int do_gradient_ascent, double* best_points,
                        double* best_kg, int* found) {
  check_ensemble(gps);
  const int d = gps[0]->d, qd = q * d, M = (int)gps.size();
  BatchObjective f;
  f.values = [&](const double* x_all, int n, double* values) {
    kg_mcmc_sums(gps, num_fidelity, inner, bounds, discrete_all, P, x_all, n, Xp, q, p, num_mc, best_so_far, normals, false,
                 values, nullptr);
    kg_mcmc_finalize(values, nullptr, x_all, n, q, d, num_fidelity, M);
  };
  f.grads = [&](const double* x_all, int n, double* grads) {
    std::vector<double> vals(n);
    kg_mcmc_sums(gps, num_fidelity, inner, bounds, discrete_all, P, x_all, n, Xp, q, p, num_mc, best_so_far, normals, true,
                 vals.data(), grads);
    kg_mcmc_finalize(vals.data(), grads, x_all, n, q, d, num_fidelity, M);
  };
  multistart(f, outer, bounds, d, qd, starts, num_starts, do_gradient_ascent, -INFINITY, best_points, best_kg, found);
}

void ei_mcmc_multistart(const std::vector<GpDev*>& gps, const moe_gd_params_t& outer, const double* bounds, const double* starts,
                        int num_starts, const double* Xp, int q, int p, int num_mc, const double* best_so_far,
                        const double* normals, int do_gradient_ascent, double* best_points, double* best_ei, int* found) {
  check_ensemble(gps);
  const int d = gps[0]->d, qd = q * d;
  const bool analytic = (q == 1 && p == 0);  // gpp_expected_improvement_mcmc_optimization.hpp:871
  if (!analytic && normals == nullptr) throw Error(MOE_ERR_RUNTIME, "q,p-EI by Monte Carlo needs a normal table");
  BatchObjective f;
  f.values = [&](const double* x_all, int n, double* values) {
    ei_mcmc_batch(gps, x_all, n, Xp, q, p, num_mc, best_so_far, normals, analytic, values, nullptr);
  };
  f.grads = [&](const double* x_all, int n, double* grads) {
    ei_mcmc_batch(gps, x_all, n, Xp, q, p, num_mc, best_so_far, normals, analytic, nullptr, grads);
  };
  // the MCMC drivers seed their IO container with 0.0, not the -1.0 of the single-GP ones
  // (gpp_expected_improvement_mcmc_optimization.hpp:914, 968; .cpp:268, 295): an all-zero EI surface reports found = 0
  multistart(f, outer, bounds, d, qd, starts, num_starts, do_gradient_ascent, 0.0, best_points, best_ei, found);
}

}  // namespace moe
