// cornell_moe_amd/csrc/kg.hpp -- Monte-Carlo acquisition evaluators (q-EI, q-KG) on the device GP.
#pragma once
#include "gp.hpp"

namespace moe {

// ExpectedImprovementEvaluator::Compute[Grad]ExpectedImprovement (gpp_math.cpp:1991-2126).  normals[num_mc][q+p].
void ei_evaluate(GpDev& gp, const double* Xq, const double* Xp, int q, int p, int num_mc, double best_so_far,
                 const double* normals, double* ei, double* grad_ei);

// The same for `num_evals` independent points_to_sample sets (EvaluateEIAtPointList, gpp_math.hpp:1900-1950, and the
// multistart axis): ei[num_evals], grad_ei[num_evals][q*dim] (either may be NULL).
void ei_evaluate_batch(GpDev& gp, const double* Xq_all, int num_evals, const double* Xp, int q, int p, int num_mc,
                       double best_so_far, const double* normals, double* ei, double* grad_ei);

// Analytic 1,0-EI (OnePotentialSampleExpectedImprovementEvaluator, gpp_math.cpp:2195-2259) at `num_evals` single points
// pts[num_evals][dim]: ei[num_evals], grad_ei[num_evals][dim] (either may be NULL).
void ei_analytic_batch(GpDev& gp, const double* pts, int num_evals, double best_so_far, double* ei, double* grad_ei);

// KnowledgeGradientEvaluator::Compute[Grad]KnowledgeGradient (gpp_knowledge_gradient_optimization.cpp:69-227) for
// `num_evals` independent points_to_sample sets; see include/moe_hip.h (moe_kg / moe_kg_batch) for argument meaning.
// best_points (may be NULL) is only filled for num_evals == 1.
void kg_evaluate_batch(GpDev& gp, int num_fidelity, const moe_gd_params_t& gd, const double* bounds, const double* discrete,
                       int P, const double* Xq_all, int num_evals, const double* Xp, int q, int p, int num_mc,
                       double best_so_far, const double* normals, int first_sample, int num_local, bool want_grad,
                       double* kg_sum, double* grad_sum, double* best_points, moe_kg_stats_t* stats);

// ---- callers of the hot path (multistart.hip) ----
// ComputeLatinHypercubePointsInDomain (gpp_random.cpp:173-194): out[num_points][dim], mt19937(seed).
void latin_hypercube(unsigned int seed, const double* bounds, int dim, int num_points, double* out);
// ComputeKGOptimalPointsToSampleViaMultistartGradientDescent / ...ViaLatinHypercubeSearch
// (gpp_knowledge_gradient_optimization.hpp:860-1141) from caller-supplied starts [num_starts][q][d].
void kg_multistart(GpDev& gp, int num_fidelity, const moe_gd_params_t& outer, const moe_gd_params_t& inner, const double* bounds,
                   const double* discrete, int P, const double* starts, int num_starts, const double* Xp, int q, int p,
                   int num_mc, double best_so_far, const double* normals, int do_gradient_ascent, double* best_points,
                   double* best_kg, int* found);
// ComputeOptimalPointsToSampleViaMultistartGradientDescent / EvaluateEIAtPointList (gpp_math.hpp:1683-1800,
// gpp_math.cpp:2305-2356) from caller-supplied starts [num_starts][q][d]; q = 1, p = 0 takes the analytic evaluator.
void ei_multistart(GpDev& gp, const moe_gd_params_t& outer, const double* bounds, const double* starts, int num_starts,
                   const double* Xp, int q, int p, int num_mc, double best_so_far, const double* normals,
                   int do_gradient_ascent, double* best_points, double* best_ei, int* found);
// ComputeOptimalPosteriorMean from one initial guess (gpp_knowledge_gradient_optimization.cpp:420-472).
void posterior_mean_optimize(GpDev& gp, int num_fidelity, const moe_gd_params_t& gd, const double* bounds, const double* x0,
                             double* best_point, double* best_value);

}  // namespace moe
