// cornell_moe_amd/csrc/kg.hpp -- Monte-Carlo acquisition evaluators (q-EI, q-KG) on the device GP.
#pragma once
#include <functional>
#include <memory>
#include <vector>

#include "gp.hpp"

namespace moe {

// ExpectedImprovementEvaluator::Compute[Grad]ExpectedImprovement (gpp_math.cpp:1991-2126).  normals[num_mc][q+p].
void ei_evaluate(GpDev& gp, const double* Xq, const double* Xp, int q, int p, int num_mc, double best_so_far,
                 const double* normals, double* ei, double* grad_ei);

// The same for `num_evals` independent points_to_sample sets (EvaluateEIAtPointList, gpp_math.hpp:1900-1950, and the
// multistart axis): ei[num_evals], grad_ei[num_evals][q*dim] (either may be NULL).
void ei_evaluate_batch(GpDev& gp, const double* Xq_all, int num_evals, const double* Xp, int q, int p, int num_mc,
                       double best_so_far, const double* normals, double* ei, double* grad_ei);
// The same in two halves (r6): everything that is enqueued, and the collection of the results once the stream has run.
struct EiPending {
  std::function<void(double* ei, double* grad_ei)> collect;
};
EiPending ei_launch(GpDev& gp, const double* Xq_all, int num_evals, const double* Xp, int q, int p, int num_mc, double best_so_far,
                    const double* normals, bool want_value, bool want_grad);
bool ei_device_algebra();

// Analytic 1,0-EI (OnePotentialSampleExpectedImprovementEvaluator, gpp_math.cpp:2195-2259) at `num_evals` single points
// pts[num_evals][dim]: ei[num_evals], grad_ei[num_evals][dim] (either may be NULL).
void ei_analytic_batch(GpDev& gp, const double* pts, int num_evals, double best_so_far, double* ei, double* grad_ei);

// KnowledgeGradientEvaluator::Compute[Grad]KnowledgeGradient (gpp_knowledge_gradient_optimization.cpp:69-227) for
// `num_evals` independent points_to_sample sets; see include/moe_hip.h (moe_kg / moe_kg_batch) for argument meaning.
// best_points (may be NULL) is only filled for num_evals == 1.
// disc_head [q][dim] (or NULL): the points_to_sample the evaluating state was BUILT at.  KnowledgeGradientState's constructor
// fills its discretised set with [union points ; discrete points] (.cpp:259-261) and SetCurrentPoint (.cpp:232-243) does not
// refresh it, so the reference's multistart / point-list drivers -- which build their states at the first start and move them
// (gpp_knowledge_gradient_optimization.hpp:886-889) -- score and start every inner optimisation from the FIRST start's points.
// NULL = a fresh state per evaluation (the single-evaluation entry points, gpp_python_knowledge_gradient.cpp:74-154).
void kg_evaluate_batch(GpDev& gp, int num_fidelity, const moe_gd_params_t& gd, const double* bounds, const double* discrete,
                       int P, const double* Xq_all, int num_evals, const double* Xp, int q, int p, int num_mc,
                       double best_so_far, const double* normals, int first_sample, int num_local, bool want_grad,
                       double* kg_sum, double* grad_sum, double* best_points, moe_kg_stats_t* stats,
                       const double* disc_head = nullptr);

// kg_evaluate_batch split at its only wait: kg_launch enqueues the whole evaluation on gp.stream -- one host->device copy, the
// state set-up with its m x m algebra (kg_state.hip), the MC kernel, the gradient tail -- and collect() waits for the stream and
// hands out kg_sum / grad_sum (same meaning as above).  gd.domain_type selects the inner optimisations' domain (tensor product
// or its intersection with the unit simplex).
struct KgPending {
  std::function<void(double* kg_sum, double* grad_sum, double* best_points, moe_kg_stats_t* stats)> collect;
};
// Evaluations one kg_launch may carry within `budget_gb` of device workspace on this GP (see kg.hip).
int kg_max_batch(const GpDev& gp, int P, int q, int p, int num_local, bool want_grad, double budget_gb);

KgPending kg_launch(GpDev& gp, int num_fidelity, const moe_gd_params_t& gd, const double* bounds, const double* discrete, int P,
                    const double* Xq_all, int num_evals, const double* Xp, int q, int p, int num_mc, double best_so_far,
                    const double* normals, int first_sample, int num_local, bool want_grad, bool want_best_points,
                    double weight_table_gb = -1.0,  // cap of the per-sample weight table; < 0: MOE_KG_V_MAX_GB (default 4)
                    const double* disc_head = nullptr);

// ---- callers of the hot path (multistart.hip) ----
// Whether the outer-optimisation drivers reproduce the reference's EXECUTION (state objects built at start_points[0] whose
// discretised set is never refreshed; the KG-MCMC state's partial SetCurrentPoint and accumulating gradient) or its intent
// (fresh state per evaluation, all q points move, plain gradient).  Default: on -- results identical to the reference's
// (env MOE_REFERENCE_QUIRKS=0 or moe_set_reference_quirks(0) turn it off; include/moe_hip.h).
bool reference_quirks();
void set_reference_quirks(int on);  // < 0: back to the environment's setting
// ensemble-wide launches of the MCMC-averaged KG evaluators (mcmc.hip, launch.hpp; include/moe_hip.h)
bool ensemble_launches();
void set_ensemble_launches(int on);  // < 0: back to the environment's setting (MOE_ENS_LAUNCH)
void ensemble_launch_stats(long long* out4);
// how many ensemble members share the launches being recorded on this thread (1: none) -- kg_launch sizes its MC grid for its share
int ensemble_members_hint();
void set_ensemble_members_hint(int members);
// The exchange step of an outer optimisation that runs on several ranks (r5): an all-gather of `count` doubles per rank, every rank
// receiving recv[world][count] in rank order.  One process per GPU passes torch.distributed's collective through the C ABI
// (moe_comm_t: RCCL over xGMI, or gloo); one process driving several devices gets an in-memory exchange between its host threads
// (moe_kg_multistart_multi).  The optimisers call it once per batched evaluation -- the merge the reference does under
// `omp critical` (gpp_optimization.hpp:1537-1545) -- and every rank then takes the same decisions on the same bits.
struct Comm {
  int rank = 0, world = 1;
  std::function<void(const double* send, double* recv, int count)> allgather;
};
// `width` doubles per item for `n` items, item i evaluated by rank i % world (the reference's omp schedule(static, 1)): eval_local
// fills out_local[n_local][width] for the items idx[0 .. n_local); the result out[n][width] is complete on every rank.  A rank
// whose evaluation throws still takes part in the exchange, and then EVERY rank throws that error (no rank is left waiting).
void sharded_items(const Comm& comm, int n, int width,
                   const std::function<void(const std::vector<int>& idx, double* out_local)>& eval_local, double* out);

// Timeline of the last outer optimisation of this process (r5; moe_multistart_trace): one row per batched evaluation the optimiser
// issued -- kind (0 = values, 1 = gradients), items, wall milliseconds (device work + exchange).  Recorded on rank 0 / worker 0 only.
void multistart_trace_begin(bool enabled);
void multistart_trace_add(int kind, int items, double ms);
int multistart_trace_get(double* out, int cap);

// A maximisation objective evaluated at a BATCH of points [n][qd]: values [n], grads [n][qd].
struct BatchObjective {
  std::function<void(const double* x_all, int n, double* values)> values;
  std::function<void(const double* x_all, int n, double* grads)> grads;
};
// RepeatedDomain<DomainType>::LimitUpdate of the outer optimisers (gpp_domain.hpp:509-520): bounds[2 d] apply to each of the qd / d
// points; outer.domain_type selects TensorProductDomain::LimitUpdate (gpp_domain.cpp:64-105) or
// SimplexIntersectTensorProductDomain::LimitUpdate (:234-290; the constructor's clipping to the unit hypercube and its emptiness
// test included: MOE_ERR_BOUNDS).  step is limited in place.
struct DomainLimiter {
  DomainLimiter(const moe_gd_params_t& outer, const double* bounds, int d);
  ~DomainLimiter();
  void apply(double max_relative_change, const double* x, double* step, int qd) const;

 private:
  struct Impl;
  int d;
  const double* bounds;
  std::unique_ptr<Impl> impl;
};
// GradientDescentOptimizer::Optimize (gpp_optimization.hpp:619-705, 1144-1185) for S starts at once, x [S][qd] in place; every step
// evaluates the gradients of all running starts in one f.grads call.  bounds[2 d] apply to each of the qd / d points.
void gradient_ascent_batch(const BatchObjective& f, const moe_gd_params_t& outer, const double* bounds, int d, int qd, double* x, int S);
// MultistartOptimizer over a batched objective: see multistart.hip.  bounds[2*d] apply to each of the qd/d points.
void multistart(const BatchObjective& f, const moe_gd_params_t& outer, const double* bounds, int d, int qd, const double* starts,
                int num_starts, int do_gradient_ascent, double floor_value, double* best_points, double* best_value,
                int* found);
// ComputeLatinHypercubePointsInDomain (gpp_random.cpp:173-194): out[num_points][dim], mt19937(seed).
void latin_hypercube(unsigned int seed, const double* bounds, int dim, int num_points, double* out);
// ComputeKGOptimalPointsToSampleViaMultistartGradientDescent / ...ViaLatinHypercubeSearch
// (gpp_knowledge_gradient_optimization.hpp:860-1141) from caller-supplied starts [num_starts][q][d].
// comm != NULL (r5): the restarts of every batched evaluation are dealt to the ranks (rank r evaluates items r, r + world, ...) and
// exchanged; every rank returns the same point, bit for bit the single-rank one (an evaluation's bits do not depend on its batch).
void kg_multistart(GpDev& gp, int num_fidelity, const moe_gd_params_t& outer, const moe_gd_params_t& inner, const double* bounds,
                   const double* discrete, int P, const double* starts, int num_starts, const double* Xp, int q, int p,
                   int num_mc, double best_so_far, const double* normals, int do_gradient_ascent, double* best_points,
                   double* best_kg, int* found, const Comm* comm = nullptr);
// ComputeOptimalPointsToSampleViaMultistartGradientDescent / EvaluateEIAtPointList (gpp_math.hpp:1683-1800,
// gpp_math.cpp:2305-2356) from caller-supplied starts [num_starts][q][d]; q = 1, p = 0 takes the analytic evaluator.
void ei_multistart(GpDev& gp, const moe_gd_params_t& outer, const double* bounds, const double* starts, int num_starts,
                   const double* Xp, int q, int p, int num_mc, double best_so_far, const double* normals,
                   int do_gradient_ascent, double* best_points, double* best_ei, int* found);
// ComputeOptimalPosteriorMean from one initial guess (gpp_knowledge_gradient_optimization.cpp:420-472).
void posterior_mean_optimize(GpDev& gp, int num_fidelity, const moe_gd_params_t& gd, const double* bounds, const double* x0,
                             double* best_point, double* best_value);

// ---- MCMC-averaged evaluators (mcmc.hip): `gps` are num_mcmc device GPs over the same data, one per
// hyper-parameter sample (GaussianProcessMCMC, gpp_knowledge_gradient_mcmc_optimization.cpp:24-49) ----
// Sums over the given GPs of the per-GP KG / grad KG (each already divided by num_mc): kg_sum[E], grad_sum[E][q*d].
void kg_mcmc_sums(const std::vector<GpDev*>& gps, int num_fidelity, const moe_gd_params_t& inner, const double* bounds,
                  const double* discrete_all, int P, const double* Xq_all, int num_evals, const double* Xp, int q, int p,
                  int num_mc, const double* best_so_far, const double* normals, bool want_grad, double* kg_sum, double* grad_sum,
                  const double* disc_head = nullptr);
// The reference's top-20 selection, tie for tie: the order in which its std::priority_queue of (-value, index) pairs pops the
// kept starts (gpp_knowledge_gradient_optimization.hpp:895-921, gpp_math.hpp:1717-1738) -- lowest kept value first, equal values
// by descending index.  All starts are kept when there are fewer than 20 (the reference pops an under-filled queue).
std::vector<int> top_k_order(const double* vals, int num_starts);
// KnowledgeGradientMCMCEvaluator::Compute[Grad]KnowledgeGradient's last step (.cpp:84-180): divide by num_mcmc and by the
// fidelity cost (and add the cost-gradient term).  In place; grad may be NULL.
void kg_mcmc_finalize(double* kg, double* grad, const double* Xq_all, int num_evals, int q, int d, int num_fidelity, int num_mcmc);
void ei_mcmc_batch(const std::vector<GpDev*>& gps, const double* Xq_all, int num_evals, const double* Xp, int q, int p, int num_mc,
                   const double* best_so_far, const double* normals, bool analytic, double* ei, double* grad_ei);
// comm != NULL (r5): `gps` are THIS rank's members of an ensemble of total_num_mcmc -- member g lives on rank g % world, local
// index g / world (discrete_all / best_so_far hold the local members' rows) -- every batched evaluation exchanges the per-member
// values and every rank adds them up in global member order: the single-rank sums, bit for bit.
void kg_mcmc_multistart(const std::vector<GpDev*>& gps, int num_fidelity, const moe_gd_params_t& outer,
                        const moe_gd_params_t& inner, const double* bounds, const double* discrete_all, int P,
                        const double* starts, int num_starts, const double* Xp, int q, int p, int num_mc,
                        const double* best_so_far, const double* normals, int do_gradient_ascent, double* best_points,
                        double* best_kg, int* found, int total_num_mcmc = -1, const Comm* comm = nullptr);
// Per-member values of a batch (each already divided by num_mc): kg_mem[num_members][E], grad_mem[num_members][E][q*d] (NULL = values only).
void kg_mcmc_members(const std::vector<GpDev*>& gps, int num_fidelity, const moe_gd_params_t& inner, const double* bounds,
                     const double* discrete_all, int P, const double* Xq_all, int num_evals, const double* Xp, int q, int p,
                     int num_mc, const double* best_so_far, const double* normals, bool want_grad, double* kg_mem, double* grad_mem,
                     const double* disc_head = nullptr);
void ei_mcmc_multistart(const std::vector<GpDev*>& gps, const moe_gd_params_t& outer, const double* bounds, const double* starts,
                        int num_starts, const double* Xp, int q, int p, int num_mc, const double* best_so_far,
                        const double* normals, int do_gradient_ascent, double* best_points, double* best_ei, int* found);

}  // namespace moe
