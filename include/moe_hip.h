/* include/moe_hip.h -- public C ABI of libmoe_hip.so (MI355X / gfx950).
 *
 * A drop-in for the one hot path of wujian16/Cornell-MOE: GP posterior + Monte-Carlo acquisition (q-EI, q-KG, d-KG).
 * Each entry point states the reference interface it replaces (file:line under moe/optimal_learning/cpp/ of the
 * reference).  The reference crosses this boundary through a boost::python module (`moe.build.GPP`,
 * gpp_python.cpp:453-600); INTEGRATION.md shows the ctypes/pybind stub a maintainer would add to bind these symbols
 * instead.
 *
 * Conventions (identical to the reference's Python boundary, gpp_python_common.cpp:52-129):
 *   - all floating point is FP64; points are row-major [point][dim]; sizes are explicit ints;
 *   - all pointers are HOST pointers owned by the caller unless the name ends in `_dev`;
 *   - every function returns a status code (0 = ok) and, on failure, fills *err (may be NULL);
 *     codes map one-to-one onto the reference's exception classes (gpp_exception.hpp:144-509).
 *   - matrices returned to the caller are column-major exactly as the reference C++ fills them
 *     (the Python shim applies the same symmetrisation / transposition as gpp_python_gaussian_process.cpp:136-185).
 * There is NO CPU fallback: every compute entry point requires a visible gfx950 device and fails with
 * MOE_ERR_RUNTIME otherwise.
 *
 * SIZE LIMITS of the device kernels (the reference has none: gpp_knowledge_gradient_optimization.hpp:310-480 allocates by
 * size).  Beyond them the KG entry points return MOE_ERR_BOUNDS (payload: value, min, max) and compute nothing:
 *   - dim <= 32                       (coordinate tables are built for padded dimensions 4, 8, 12, 16, 24, 32);
 *   - num_derivatives <= 12           (derivative-weight slots of the d-KG Monte-Carlo kernels: 0..4, 8, 12);
 *   - (num_to_sample + num_being_sampled) * (1 + num_derivatives) <= 128   (m, the fantasy-observation count of one evaluation);
 *   - |x - mean(X)| / length <= 1e5 per coordinate for every tabulated point (32-bit exponent arithmetic of the table exp).
 * q,p-EI (moe_ei*): num_to_sample + num_being_sampled <= 64; up to 16 the whole evaluation stays on the device, beyond that its
 *   u x u algebra (variance, factor, Smith's derivative) runs on the host between two waits (r6; it was refused).
 * Every configuration BASELINE.json names is inside them (C5 with all 12 derivatives observed: m = 104).  The GP itself
 * (moe_gp_*, moe_ll_*) is limited by device memory only (N = 26 000 builds in 0.3 s; two N x N matrices stay resident).
 */
#ifndef MOE_HIP_H_
#define MOE_HIP_H_

#ifdef __cplusplus
extern "C" {
#endif

#define MOE_OK 0
#define MOE_ERR_RUNTIME 1        /* OptimalLearningException (gpp_exception.hpp:144) - also HIP runtime failures */
#define MOE_ERR_BOUNDS 2         /* BoundsException<T>(value, min, max)        (gpp_exception.hpp:260) */
#define MOE_ERR_INVALID_VALUE 3  /* InvalidValueException<T>(value, truth, tol) (gpp_exception.hpp:350) */
#define MOE_ERR_SINGULAR 4       /* SingularMatrixException(num_rows, leading_minor_index) (gpp_exception.hpp:440) */

#define MOE_COV_SQUARE_EXPONENTIAL 0 /* gpp_covariance.hpp:195 */
#define MOE_COV_MATERN_NU_2P5 1      /* gpp_covariance.hpp:313 (what the Python boundary always builds, gpp_python_gaussian_process.cpp:53) */

typedef struct moe_error {
  int code;
  char message[480];
  double payload[3]; /* Bounds: value,min,max; InvalidValue: value,truth,tolerance; Singular: num_rows,leading_minor_index,0 */
} moe_error_t;

/* GradientDescentParameters (gpp_optimizer_parameters.hpp:81-133) */
typedef struct moe_gd_params {
  int num_multistarts;
  int max_num_steps;
  int max_num_restarts;
  int num_steps_averaged;
  double gamma;
  double pre_mult;
  double max_relative_change;
  double tolerance;
  int domain_type; /* (r4) 0 = tensor-product domain, 1 = its intersection with the unit simplex {x_i >= 0, sum x_i <= 1}
                      (SimplexIntersectTensorProductDomain, gpp_domain.hpp:215-349; gpp_domain.cpp:234-290).  In the OUTER parameters
                      of a multistart driver: applied to each of the q points (RepeatedDomain).  In the INNER parameters of a KG
                      evaluation / driver and in moe_posterior_mean_optimize: the domain of every sample's posterior-mean
                      optimisation over the dim - num_fidelity free coordinates (the reference builds both of one type:
                      gpp_python_knowledge_gradient.cpp:279-296, 327-341; its single-evaluation entry points always pass a
                      tensor product: :97-144). */
} moe_gd_params_t;
#define MOE_DOMAIN_TENSOR_PRODUCT 0
#define MOE_DOMAIN_SIMPLEX 1

/* Counters the device fills during one KG evaluation (S and G must be counted on device). */
typedef struct moe_kg_stats {
  long long posterior_mean_evals; /* value-only passes over the N+m points made by the line search (all samples) */
  long long posterior_grad_evals; /* value+gradient passes */
  double ms_state;                /* wall ms: state set-up (K*, solves, Var, chol, grad chol) */
  double ms_mc;                   /* wall ms: MC inner-optimisation kernel */
  double ms_tail;                 /* wall ms: gradient tail (N x M covariance build + contraction) */
} moe_kg_stats_t;

typedef struct moe_gp moe_gp_t; /* opaque; replaces the heap GaussianProcess owned by the Python object
                                   (gpp_python_gaussian_process.cpp:55-61).  `const moe_gp_t*` means the GP it represents is
                                   not changed; every call works in the handle's own device workspaces, so calls on ONE handle
                                   are serialised by a per-handle mutex (calls on different handles run concurrently).
                                   NULL handles / NULL mandatory arguments return MOE_ERR_RUNTIME, never crash. */

/* ---- runtime ---- */
int moe_device_count(int* count);
/* "gfx950" etc. of device `device`; name_len >= 64. */
int moe_device_arch(int device, char* name, int name_len);
const char* moe_version(void);

/* ---- Gaussian process: GaussianProcess ctor (gpp_math.cpp:553-573, RecomputeDerivedVariables :481-511) ----
 * hyperparameters = [alpha, length_0 .. length_{d-1}] (gpp_python_common.cpp:100-105). K assembly, Cholesky,
 * triangular inverse and K^-1 (y - mean) run on device `device` and stay resident there.
 * Fails with MOE_ERR_SINGULAR (payload: N, leading minor index) when a pivot <= 1e-16 (gpp_linear_algebra.cpp:118). */
int moe_gp_create(const double* hyperparameters, int cov_type, const double* points_sampled,
                  const double* points_sampled_value, const double* noise_variance, const int* derivatives,
                  int num_derivatives, int dim, int num_sampled, int device, moe_gp_t** gp_out, moe_error_t* err);
int moe_gp_destroy(moe_gp_t* gp);
int moe_gp_dim(const moe_gp_t* gp);
int moe_gp_num_sampled(const moe_gp_t* gp);
int moe_gp_num_derivatives(const moe_gp_t* gp);
/* AddPointsToGP (gpp_math.cpp:1699-1718): appends points and re-derives everything (mean included). */
int moe_gp_add_points(moe_gp_t* gp, const double* new_points, const double* new_values, int num_new, moe_error_t* err);
/* Debug/parity accessors: K_chol [N*N col-major, lower triangle valid], K_inv_y [N], mean (any may be NULL). */
int moe_gp_get_factor(const moe_gp_t* gp, double* K_chol, double* K_inv_y, double* mean, moe_error_t* err);

/* ---- posterior queries; semantics of the Python-visible methods (gpp_python_gaussian_process.cpp:64-236) ----
 * m = num_pts * (1 + num_derivatives of the GP). */
/* compute_mean_of_points -> ComputeMeanOfPoints (gpp_math.cpp:662-678); out[num_pts] (function values only) */
int moe_gp_mean(const moe_gp_t* gp, const double* pts, int num_pts, double* out, moe_error_t* err);
/* compute_mean_of_additional_points -> ComputeMeanOfAdditionalPoints (gpp_math.cpp:688-710); out[num_pts] */
int moe_gp_additional_mean(const moe_gp_t* gp, const double* pts, int num_pts, double* out, moe_error_t* err);
/* compute_grad_mean_of_points -> ComputeGradMeanOfPoints (gpp_math.cpp:721-726); out[dim * m] */
int moe_gp_grad_mean(const moe_gp_t* gp, const double* pts, int num_pts, double* out, moe_error_t* err);
/* compute_variance_of_points -> ComputeVarianceOfPoints (gpp_math.cpp:924-970); out[m*m] col-major */
int moe_gp_variance(const moe_gp_t* gp, const double* pts, int num_pts, double* out, moe_error_t* err);
/* compute_cholesky_variance_of_points: chol of the above (lower; strict upper holds the variance leftovers exactly like
 * ComputeCholeskyFactorL leaves them); MOE_ERR_SINGULAR on failure */
int moe_gp_cholesky_variance(const moe_gp_t* gp, const double* pts, int num_pts, double* out, moe_error_t* err);
/* compute_grad_variance_of_points -> ComputeGradVarianceOfPoints (gpp_math.cpp:1359-1373); out[num_derivs][m][m][dim] */
int moe_gp_grad_variance(const moe_gp_t* gp, const double* pts, int num_pts, int num_derivs, double* out, moe_error_t* err);
/* compute_grad_cholesky_variance_of_points -> ComputeGradCholeskyVarianceOfPoints (gpp_math.cpp:1454-1474) */
int moe_gp_grad_cholesky_variance(const moe_gp_t* gp, const double* pts, int num_pts, int num_derivs, double* out,
                                  moe_error_t* err);

/* compute_posterior_mean / compute_grad_posterior_mean (gpp_python_knowledge_gradient.cpp:44-72 ->
 * PosteriorMeanEvaluator, gpp_knowledge_gradient_optimization.cpp:322-351). point[dim - num_fidelity];
 * value = -mu(point, fidelity coords = 1); grad[dim - num_fidelity] = -grad mu. Either output may be NULL. */
int moe_posterior_mean(const moe_gp_t* gp, int num_fidelity, const double* point, double* value, double* grad,
                       moe_error_t* err);

/* ---- normal draws ----
 * Fills out[count] with N(0,1) draws: the host-side stand-in for NormalRNG(seed) (gpp_random.hpp:204-303 = boost::mt19937 +
 * boost::normal_distribution).  The engine is bit-identical to std::mt19937; boost::normal_distribution's algorithm is
 * Boost-version dependent and the reference pins no draws, so parity with an arbitrary Boost is NOT claimed.
 * The stream IS, draw for draw, that of the reference as it builds in this repository (oracle/_ref, Boost shimmed onto the
 * C++ standard library: Marsaglia's polar method over generate_canonical<double, 53>) -- pinned by tests/test_oracle.py and
 * the committed fixture tests/golden/ref_kg_multistart.npz (keys stream_seeds / stream_draws) -- so a seeded RandomnessSourceContainer run reproduces that
 * build's results; cross-implementation parity runs still pass explicit tables. */
int moe_normal_draws(unsigned int seed, long long count, double* out);

/* ---- q,p-EI by Monte Carlo: compute_expected_improvement / compute_grad_expected_improvement
 * (gpp_python_expected_improvement.cpp:44-109 -> ExpectedImprovementEvaluator, gpp_math.cpp:1991-2126).
 * normals[num_mc][q+p] is the explicit N(0,1) table (row i feeds sample i, the role NormalRNGSimulator plays in the
 * reference's tests, gpp_random.hpp:314-340).  ei and/or grad_ei[q*dim] may be NULL. */
int moe_ei(const moe_gp_t* gp, const double* points_to_sample, const double* points_being_sampled, int num_to_sample,
           int num_being_sampled, int num_mc, double best_so_far, const double* normals, double* ei, double* grad_ei,
           moe_error_t* err);

/* Batched form: `num_evals` independent points_to_sample sets (points_to_sample_all[e][q][dim]) against the same
 * points_being_sampled and normal table -- evaluate_EI_at_point_list (gpp_python_expected_improvement.cpp:401-440 ->
 * EvaluateEIAtPointList, gpp_math.hpp:1900-1950).  ei[num_evals] and/or grad_ei[num_evals][q*dim] may be NULL. */
int moe_ei_batch(const moe_gp_t* gp, const double* points_to_sample_all, int num_evals, const double* points_being_sampled,
                 int num_to_sample, int num_being_sampled, int num_mc, double best_so_far, const double* normals,
                 double* ei, double* grad_ei, moe_error_t* err);

/* Analytic 1,0-EI and its gradient at `num_evals` single points (points[num_evals][dim]) --
 * OnePotentialSampleExpectedImprovementEvaluator::Compute[Grad]ExpectedImprovement (gpp_math.cpp:2195-2259), the evaluator
 * the reference's multistart / point-list drivers take when num_to_sample == 1 and num_being_sampled == 0
 * (gpp_math.hpp:1703, gpp_math.cpp:2317).  ei[num_evals] and/or grad_ei[num_evals][dim] may be NULL. */
int moe_ei_analytic_batch(const moe_gp_t* gp, const double* points, int num_evals, double best_so_far, double* ei,
                          double* grad_ei, moe_error_t* err);

/* q,p-EI optimisation from caller-supplied starts (start_points[num_starts][q][dim]) --
 * ComputeOptimalPointsToSampleViaMultistartGradientDescent (gpp_math.hpp:1683-1800: EI at every start, best 20 kept,
 * restarted gradient ascent on each, best end point returned) when do_gradient_ascent != 0, EvaluateEIAtPointList
 * (gpp_math.cpp:2305-2356: best start by value) otherwise; behind multistart_expected_improvement_optimization
 * (gpp_python_expected_improvement.cpp:221-276).  q == 1 && p == 0 uses the analytic evaluator (normals may be NULL);
 * otherwise normals[num_mc][q+p] is replayed by every evaluation.  Tensor-product domain: domain_bounds[2*dim] applies to
 * each of the q points.  *found = 1 when some end point beats -1.0, the reference's starting best (gpp_math.hpp:1728). */
int moe_ei_multistart(const moe_gp_t* gp, const moe_gd_params_t* outer_params, const double* domain_bounds,
                      const double* start_points, int num_starts, const double* points_being_sampled, int num_to_sample,
                      int num_being_sampled, int num_mc, double best_so_far, const double* normals, int do_gradient_ascent,
                      double* best_points, double* best_ei, int* found, moe_error_t* err);

/* ---- q-KG / d-KG by Monte Carlo: compute_knowledge_gradient / compute_grad_knowledge_gradient
 * (gpp_python_knowledge_gradient.cpp:74-154 -> KnowledgeGradientEvaluator<TensorProductDomain>,
 * gpp_knowledge_gradient_optimization.cpp:69-227, state :246-317, inner optimisation :420-472,
 * gpp_optimization.hpp:708-828/1242-1283, gpp_domain.cpp:64-105).
 *   domain_bounds[2*(dim-num_fidelity)] = [min0,max0,...]; discrete_pts[num_pts][dim-num_fidelity];
 *   normals[ceil(num_mc/2)][m], m = (q+p)(1+g): even sample 2j uses row j, odd sample 2j+1 uses -row j
 *   (antithetic pairs, .cpp:171-180);
 *   first_sample/num_local select the contiguous, even-aligned slice [first_sample, first_sample+num_local) of the MC
 *   samples this call evaluates (multi-GPU sharding); pass 0,num_mc for the whole evaluation.
 * Outputs: kg_sum = SUM over the local samples of (best_posterior + best_function_value) (divide by num_mc after the
 * cross-rank sum); grad_sum[q*dim] likewise un-normalised, the winner term M*grad_mu (.cpp:157-161) being added only
 * by the call with first_sample == 0; best_points[num_local][dim] (may be NULL); stats (may be NULL).
 * want_grad = 0 reproduces ComputeKnowledgeGradient (value only). */
int moe_kg(const moe_gp_t* gp, int num_fidelity, const moe_gd_params_t* inner_params, const double* domain_bounds,
           const double* discrete_pts, int num_pts, const double* points_to_sample, const double* points_being_sampled,
           int num_to_sample, int num_being_sampled, int num_mc, double best_so_far, const double* normals,
           int first_sample, int num_local, int want_grad, double* kg_sum, double* grad_sum, double* best_points,
           moe_kg_stats_t* stats, moe_error_t* err);

/* Batched form: `num_evals` independent evaluations (different points_to_sample[e][q][dim], same GP / discrete set /
 * normals) in one pass -- the multistart axis of ComputeKGOptimalPointsToSampleViaMultistartGradientDescent
 * (gpp_knowledge_gradient_optimization.hpp:860-935) and EvaluateKGAtPointList (:1090-1141).
 * kg_sum[num_evals], grad_sum[num_evals][q*dim]. */
int moe_kg_batch(const moe_gp_t* gp, int num_fidelity, const moe_gd_params_t* inner_params, const double* domain_bounds,
                 const double* discrete_pts, int num_pts, const double* points_to_sample_all, int num_evals,
                 const double* points_being_sampled, int num_to_sample, int num_being_sampled, int num_mc,
                 double best_so_far, const double* normals, int first_sample, int num_local, int want_grad,
                 double* kg_sum, double* grad_sum, moe_kg_stats_t* stats, moe_error_t* err);

/* ---- one node, several devices, from a plain C/C++ host (SURVEY 8b "multistart drivers taking num_devices"; the axis the
 * reference runs as OpenMP iterations, gpp_optimization.hpp:1472-1546).  gps[num_devices] are handles of the SAME GP built on
 * different devices (moe_gp_create with device = 0 .. num_devices-1; distinct handles on one device are accepted too); one
 * host thread per handle drives its device.
 *   shard_mode 0 (restarts): evaluation e runs whole on handle e % num_devices; results equal moe_kg_batch's bit for bit
 *     (which kernel an evaluation takes is decided from the training set, points_being_sampled and the domain box alone --
 *     never from the other evaluations of its batch -- as long as its points_to_sample lie inside domain_bounds).
 *   shard_mode 1 (MC samples): every handle evaluates every point set on its contiguous EVEN-ALIGNED slice of the samples and
 *     the per-handle sums are added on the host in handle order (a fixed-order reduction of num_evals x (1 + q d) doubles).
 * Outputs as moe_kg_batch (un-normalised sums over all num_mc samples); stats: pass counts summed, times = the slowest
 * handle's.  The multi-PROCESS equivalent (one rank per GPU, RCCL) is cornell_moe_amd/dist.py. */
int moe_kg_batch_multi(const moe_gp_t* const* gps, int num_devices, int shard_mode, int num_fidelity,
                       const moe_gd_params_t* inner_params, const double* domain_bounds, const double* discrete_pts, int num_pts,
                       const double* points_to_sample_all, int num_evals, const double* points_being_sampled,
                       int num_to_sample, int num_being_sampled, int num_mc, double best_so_far, const double* normals,
                       int want_grad, double* kg_sum, double* grad_sum, moe_kg_stats_t* stats, moe_error_t* err);

/* ---- callers of the hot path: the outer optimisation over points_to_sample ----
 * multistart_knowledge_gradient_optimization (gpp_python_knowledge_gradient.cpp:243-313) ->
 * ComputeKGOptimalPointsToSampleViaMultistartGradientDescent (gpp_knowledge_gradient_optimization.hpp:860-935) when
 * do_gradient_ascent != 0: KG at every start, best 20 kept, restarted gradient ascent with `outer_params`
 * (gpp_optimization.hpp:619-705, 1144-1185), best end point returned; with do_gradient_ascent == 0 the value search of
 * ...ViaLatinHypercubeSearch / EvaluateKGAtPointList (:1090-1141).  start_points[num_starts][q][dim] are supplied by
 * the caller (moe_latin_hypercube reproduces the reference's generator, gpp_random.cpp:173-194); domain_bounds[2*dim].
 * Every step evaluates all live restarts in ONE batched device pass.  *found = 1 iff a point with KG > -inf was found.
 * Reproduced from the reference, because they decide which point is returned (pinned to the reference's own end points,
 * tests/golden/ref_kg_multistart.npz): (1) the drivers build their KnowledgeGradientState at the FIRST start and move it with
 * SetCurrentPoint, which does not refresh the state's discretised set (gpp_knowledge_gradient_optimization.cpp:232-243,
 * 259-261): every evaluation of a run scores / starts its inner optimisation from start_points[0]'s q points (+ the
 * points being sampled + discrete_pts), not from the points being evaluated -- moe_kg / moe_kg_batch, like the
 * single-evaluation Python entry points, evaluate on a fresh state; (2) the best 20 starts are kept and walked in the order
 * the reference's std::priority_queue pops them (lowest kept value first, equal values by descending index), and the first
 * of equal end values wins. */
int moe_kg_multistart(const moe_gp_t* gp, int num_fidelity, const moe_gd_params_t* outer_params,
                      const moe_gd_params_t* inner_params, const double* domain_bounds, const double* discrete_pts, int num_pts,
                      const double* start_points, int num_starts, const double* points_being_sampled, int num_to_sample,
                      int num_being_sampled, int num_mc, double best_so_far, const double* normals, int do_gradient_ascent,
                      double* best_points, double* best_kg, int* found, moe_error_t* err);
/* The outer-optimisation drivers (moe_kg_multistart, moe_kg_mcmc_multistart) reproduce the reference's EXECUTION by default --
 * the frozen discretised set, and for KG-MCMC the partially-updated state and the accumulating gradient described at
 * moe_kg_mcmc_multistart -- because the bar of this library is "the reference's results on the reference's inputs".  Those
 * behaviours are defects of the reference's drivers, and for q > 1 the KG-MCMC one optimises only the first point.  Switch:
 *   moe_set_reference_quirks(0)  (or MOE_REFERENCE_QUIRKS=0 in the environment)  -> the drivers as the reference intends them:
 *     every evaluation on a fresh state, all q points move and are returned, the plain gradient at every step;
 *   moe_set_reference_quirks(1)  -> bug-compatible (the default);  moe_set_reference_quirks(-1) -> back to the environment.
 * Process-wide; the single-evaluation entry points (moe_kg, moe_kg_batch, moe_kg_mcmc_batch) are unaffected: they always build
 * a fresh state per call, as the reference's Python boundary does. */
int moe_set_reference_quirks(int on);
int moe_get_reference_quirks(void);
/* Ensemble-wide launches (r6): the MCMC-averaged KG entry points (moe_kg_mcmc_batch, moe_kg_mcmc_multistart and their _comm
 * forms: KnowledgeGradientMCMCEvaluator, gpp_knowledge_gradient_mcmc_optimization.cpp:129-180, evaluates the ensemble members one
 * after another; and moe_ei_mcmc_batch / moe_ei_mcmc_multistart for Monte-Carlo EI) record every member's chain of kernels and issue each kernel ONCE for all members of the ensemble; same bits as
 * member-by-member launches.  moe_set_ensemble_launches(0) -> member by member (MOE_ENS_LAUNCH=0 in the environment does the same),
 * (1) -> on (the default), (-1) -> back to the environment.  moe_ensemble_launch_stats: out[0] = evaluations of an ensemble that
 * went down merged, out[1] = that fell back to member-by-member launches, out[2] = kernel launches issued by merged evaluations,
 * out[3] = member launches they stand for.  Process-wide. */
int moe_set_ensemble_launches(int on);
int moe_ensemble_launch_stats(long long* out4);
/* posterior_mean_optimization (gpp_python_knowledge_gradient.cpp:315-342) -> ComputeOptimalPosteriorMean from ONE initial
 * guess: line-search ascent on -mu with fidelity coordinates pinned to 1.  best_point[dim - num_fidelity]. */
int moe_posterior_mean_optimize(const moe_gp_t* gp, int num_fidelity, const moe_gd_params_t* params, const double* domain_bounds,
                                const double* initial_guess, double* best_point, double* best_value, moe_error_t* err);
/* ComputeLatinHypercubePointsInDomain (gpp_random.cpp:173-194) with mt19937(seed): out[num_points][dim]. */
int moe_latin_hypercube(unsigned int seed, const double* domain_bounds, int dim, int num_points, double* out);

/* ---- MCMC-averaged evaluators: the acquisition averaged over `num_mcmc` GPs built on the same data, one
 * per hyper-parameter sample -- GaussianProcessMCMC (gpp_knowledge_gradient_mcmc_optimization.cpp:24-49; Python ctor
 * gpp_python_knowledge_gradient_mcmc.cpp:45-75).  `gps` is an array of num_mcmc handles from moe_gp_create (the caller owns
 * them; they must share dim and the observed-derivative list).  best_so_far[num_mcmc] is per GP; every GP replays the same
 * normal table. ----
 * compute_knowledge_gradient_mcmc / compute_grad_knowledge_gradient_mcmc / evaluate_KG_mcmc_at_point_list
 * (gpp_python_knowledge_gradient_mcmc.cpp:77-190, 400-470 -> KnowledgeGradientMCMCEvaluator, .cpp:51-180) for `num_evals`
 * point sets points_to_sample_all[num_evals][q][dim]; discrete_pts_all[num_mcmc][num_pts][dim - num_fidelity].
 * finalize != 0: kg[e] = mean_i KG_i / cost, grad likewise with the cost-gradient term (cost = largest product of the
 * fidelity coordinates over the q points, 1 when num_fidelity == 0; .cpp:84-127), normalised by total_num_mcmc.
 * finalize == 0: plain sums over the GPs given -- the GP-index shard of a multi-GPU evaluation, to be all-reduced and then
 * passed through moe_kg_mcmc_finalize.  grad_kg may be NULL (value only). */
int moe_kg_mcmc_batch(const moe_gp_t* const* gps, int num_mcmc, int num_fidelity, const moe_gd_params_t* inner_params,
                      const double* domain_bounds, const double* discrete_pts_all, int num_pts,
                      const double* points_to_sample_all, int num_evals, const double* points_being_sampled, int num_to_sample,
                      int num_being_sampled, int num_mc, const double* best_so_far, const double* normals, int finalize,
                      int total_num_mcmc, double* kg, double* grad_kg, moe_error_t* err);
int moe_kg_mcmc_finalize(double* kg, double* grad_kg, const double* points_to_sample_all, int num_evals, int num_to_sample,
                         int dim, int num_fidelity, int total_num_mcmc);
/* compute_expected_improvement_mcmc / compute_grad_expected_improvement_mcmc / evaluate_EI_mcmc_at_point_list
 * (gpp_python_expected_improvement_mcmc.cpp:42-108 -> ExpectedImprovementMCMCEvaluator,
 * gpp_expected_improvement_mcmc_optimization.cpp:48-88); analytic != 0 takes the 1,0-EI evaluator (:136-176; needs
 * num_to_sample == 1, num_being_sampled == 0; normals may be NULL).  ei and/or grad_ei may be NULL. */
int moe_ei_mcmc_batch(const moe_gp_t* const* gps, int num_mcmc, const double* points_to_sample_all, int num_evals,
                      const double* points_being_sampled, int num_to_sample, int num_being_sampled, int num_mc,
                      const double* best_so_far, const double* normals, int analytic, double* ei, double* grad_ei,
                      moe_error_t* err);
/* multistart_knowledge_gradient_mcmc_optimization / multistart_expected_improvement_mcmc_optimization from caller-supplied
 * starts (gpp_knowledge_gradient_mcmc_optimization.hpp:665-862, gpp_expected_improvement_mcmc_optimization.hpp:840-990):
 * same driver as moe_kg_multistart / moe_ei_multistart on the MCMC-averaged objective.  The EI driver reports found = 0
 * unless some end point has EI > 0 (the reference seeds it with 0.0, not -1.0).
 * moe_kg_mcmc_multistart follows the reference's EXECUTION, not its intent (end points pinned to the reference's,
 * tests/golden/ref_kg_multistart.npz): KnowledgeGradientMCMCState::SetCurrentPoint copies only the first of the q points into
 * the array GetCurrentPoint and the fidelity cost read (gpp_knowledge_gradient_mcmc_optimization.cpp:186-195, .hpp:439-441), so
 * the optimiser steps from and returns [moved first point ; points 2..q of start_points[0]] while the objective is evaluated at
 * the points really reached; ComputeGradKnowledgeGradient accumulates into its output (.cpp:163-166), which the optimiser
 * allocates once per restart: step i sees ((G_{i-1} + sum of the per-GP gradients) / num_mcmc * cost - KG * gradcost) / cost^2;
 * the per-GP states keep start_points[0]'s discretised set (see moe_kg_multistart).  With q = 1 and no fidelity dimension
 * only the last two are visible. */
int moe_kg_mcmc_multistart(const moe_gp_t* const* gps, int num_mcmc, int num_fidelity, const moe_gd_params_t* outer_params,
                           const moe_gd_params_t* inner_params, const double* domain_bounds, const double* discrete_pts_all,
                           int num_pts, const double* start_points, int num_starts, const double* points_being_sampled,
                           int num_to_sample, int num_being_sampled, int num_mc, const double* best_so_far,
                           const double* normals, int do_gradient_ascent, double* best_points, double* best_kg, int* found,
                           moe_error_t* err);
/* ---- r5: a whole suggestion on several GPUs (SURVEY 8e + 8f rank 1 / 2).  The reference parallelises its outer optimisers over
 * the restarts -- omp-parallel GradientDescentOptimizer runs merged under `omp critical`
 * (gpp_optimization.hpp:1472-1546, gpp_knowledge_gradient_optimization.hpp:860-935) -- and its MCMC-averaged objective is a sum
 * over independent GPs (gpp_knowledge_gradient_mcmc_optimization.hpp:666-1023).  Here every batched evaluation of the optimiser is
 * dealt to the ranks and followed by ONE all-gather; every rank then takes the same decisions on the same bits and returns the
 * same point:
 *   - moe_kg_multistart_comm: the restarts are dealt (evaluation i of a batch on rank i % world): the result is moe_kg_multistart's
 *     BIT FOR BIT, for any world size (an evaluation's bits do not depend on the batch it shares a call with);
 *   - moe_kg_mcmc_multistart_comm: the ensemble members are dealt -- member g is built and evaluated on rank g % world, which passes
 *     its members (ascending g) with their rows of discrete_pts / best_so_far; the per-member values are exchanged and every rank
 *     adds them up in global member order: moe_kg_mcmc_multistart's result bit for bit.
 * The exchange is the caller's: one process per GPU hands in its collective (cornell_moe_amd/dist.py: torch.distributed all_gather,
 * backend nccl = RCCL over xGMI, or gloo); a rank whose evaluation fails still takes part in the exchange and then EVERY rank
 * returns that error -- no rank is left waiting in a collective.  Payloads are small (a GD step of 20 restarts at q d = 32:
 * 5 KB per rank). */
typedef int (*moe_allgather_fn)(void* ctx, const double* send, double* recv, int count); /* recv[world][count], rank order; 0 = ok */
typedef struct moe_comm {
  int rank;  /* this process */
  int world; /* number of processes; 1 = no exchange (allgather may be NULL) */
  moe_allgather_fn allgather;
  void* ctx; /* passed back to allgather */
} moe_comm_t;
/* Timeline of the LAST outer optimisation this process ran (moe_kg_multistart*, moe_kg_mcmc_multistart*; rank 0 / worker 0): one row
 * per batched evaluation the optimiser issued, out[3 i ..] = kind (0 values, 1 gradients), items in the batch, wall milliseconds
 * (device work + exchange).  Returns the number of rows (out may be NULL / cap 0 to ask). */
int moe_multistart_trace(double* out, int cap);

/* Device-memory pool (r5).  The library keeps the device buffers, pinned staging buffers and streams its objects release (a GP of
 * N = 8000 holds 1 GB; a fresh hipMalloc / hipStreamCreate per hyper-parameter sample costs a 14 ms build 3 ms) and hands them to the
 * next object they fit.  Environment: MOE_POOL=0 switches it off; MOE_POOL_MAX_GB (default 48) bounds the device bytes held.
 * moe_pool_held_bytes: device bytes the pool holds right now (not those in use by live objects).  moe_pool_trim: returns all of them
 * (and the pinned buffers) to the runtime -- call it before handing the device to another library that needs the memory.  No
 * counterpart in the reference (its GaussianProcess lives in host memory). */
long long moe_pool_held_bytes(void);
int moe_pool_trim(void);
/* (diagnostic) the deal-and-exchange step alone, on synthetic items -- out[n][width], item i = seed + i + j / 1000; fail_item >= 0
 * makes its owner fail with MOE_ERR_SINGULAR, which every rank must then report.  No device work: the CPU tests run it over gloo. */
int moe_debug_sharded_items(const moe_comm_t* comm, int n, int width, double seed, int fail_item, double* out, moe_error_t* err);
int moe_kg_multistart_comm(const moe_gp_t* gp, const moe_comm_t* comm, int num_fidelity, const moe_gd_params_t* outer_params,
                           const moe_gd_params_t* inner_params, const double* domain_bounds, const double* discrete_pts,
                           int num_pts, const double* start_points, int num_starts, const double* points_being_sampled,
                           int num_to_sample, int num_being_sampled, int num_mc, double best_so_far, const double* normals,
                           int do_gradient_ascent, double* best_points, double* best_kg, int* found, moe_error_t* err);
int moe_kg_mcmc_multistart_comm(const moe_gp_t* const* local_gps, int num_local, int total_num_mcmc, const moe_comm_t* comm,
                                int num_fidelity, const moe_gd_params_t* outer_params, const moe_gd_params_t* inner_params,
                                const double* domain_bounds, const double* discrete_pts_local, int num_pts,
                                const double* start_points, int num_starts, const double* points_being_sampled,
                                int num_to_sample, int num_being_sampled, int num_mc, const double* best_so_far_local,
                                const double* normals, int do_gradient_ascent, double* best_points, double* best_kg, int* found,
                                moe_error_t* err);
/* Native exchange (r6): moe_comm_t carried by RCCL itself -- ncclAllGather on a stream and staging buffers the library owns, one pinned
 * copy in, one out, one stream wait per exchange; the optimiser loop never re-enters the host language.  This is the merge the reference
 * does under `omp critical` (cpp/gpp_optimization.hpp:1537-1545) once its restarts are dealt to processes, one per GPU.  librccl is
 * resolved at run time (dlopen; MOE_RCCL_LIB overrides the search): without it these entry points return MOE_ERR_RUNTIME.
 *   moe_rccl_unique_id: on rank 0; the 128 bytes travel to the other ranks by whatever the host has (dist.py: the gloo control group).
 *   moe_rccl_create: collective over the `world` ranks (ncclCommInitRank), one rank per device.
 *   moe_rccl_comm: fills a moe_comm_t (valid while the handle lives) for moe_kg_multistart_comm / moe_kg_mcmc_multistart_comm.
 *   moe_rccl_allreduce_sum: the MC-sharded evaluation's one collective (1 + q d doubles), host buffer in and out.
 *   moe_rccl_stats: exchanges so far, bytes received, seconds spent in them. */
#define MOE_RCCL_ID_BYTES 128
typedef struct moe_rccl moe_rccl_t;
int moe_rccl_unique_id(char* id, moe_error_t* err);
int moe_rccl_create(const char* id, int rank, int world, int device, moe_rccl_t** out, moe_error_t* err);
int moe_rccl_comm(moe_rccl_t* r, moe_comm_t* out);
int moe_rccl_allreduce_sum(moe_rccl_t* r, double* inout, int count, moe_error_t* err);
int moe_rccl_stats(const moe_rccl_t* r, long long* calls, long long* bytes, double* seconds);
void moe_rccl_destroy(moe_rccl_t* r);
/* The same for ONE process that drives several devices (a C / C++ host without torch; the twins of moe_kg_batch_multi): one host
 * thread per worker, the exchange in shared memory.  moe_kg_multistart_multi: gps[num_devices] hold the SAME GP on different
 * devices.  moe_kg_mcmc_multistart_multi: gps[num_mcmc] is the whole ensemble (the caller builds member g on device
 * g % num_workers); worker k takes members k, k + num_workers, ...  Results as above: bit for bit the single-device ones. */
int moe_kg_multistart_multi(const moe_gp_t* const* gps, int num_devices, int num_fidelity, const moe_gd_params_t* outer_params,
                            const moe_gd_params_t* inner_params, const double* domain_bounds, const double* discrete_pts,
                            int num_pts, const double* start_points, int num_starts, const double* points_being_sampled,
                            int num_to_sample, int num_being_sampled, int num_mc, double best_so_far, const double* normals,
                            int do_gradient_ascent, double* best_points, double* best_kg, int* found, moe_error_t* err);
int moe_kg_mcmc_multistart_multi(const moe_gp_t* const* gps, int num_mcmc, int num_workers, int num_fidelity,
                                 const moe_gd_params_t* outer_params, const moe_gd_params_t* inner_params,
                                 const double* domain_bounds, const double* discrete_pts_all, int num_pts,
                                 const double* start_points, int num_starts, const double* points_being_sampled,
                                 int num_to_sample, int num_being_sampled, int num_mc, const double* best_so_far,
                                 const double* normals, int do_gradient_ascent, double* best_points, double* best_kg, int* found,
                                 moe_error_t* err);
int moe_ei_mcmc_multistart(const moe_gp_t* const* gps, int num_mcmc, const moe_gd_params_t* outer_params,
                           const double* domain_bounds, const double* start_points, int num_starts,
                           const double* points_being_sampled, int num_to_sample, int num_being_sampled, int num_mc,
                           const double* best_so_far, const double* normals, int do_gradient_ascent, double* best_points,
                           double* best_ei, int* found, moe_error_t* err);

/* ---- log marginal likelihood of the data under the GP prior: compute_log_likelihood /
 * evaluate_log_likelihood_at_hyperparameter_list (gpp_python_model_selection.cpp:43-69, 270-340 ->
 * LogMarginalLikelihoodEvaluator, gpp_model_selection.cpp:540-612).  A handle keeps the data and the device buffers so
 * that a hyper-parameter sampler's thousands of evaluations on the same data re-use them.
 * hyperparameters_all[num_sets][1 + dim + 1 + num_derivatives] = (alpha, lengths[dim], noise_variance[1 + g]) per set, the
 * layout of the reference's hyperparameter lists (gpp_python_model_selection.cpp:301-303).  Like the reference, 1e-6 is
 * added to the diagonal of K + noise before it is factored (gpp_model_selection.cpp:546-549).  Where the reference
 * ignores a failed factorisation (:551-553, "TODO(GH-211)") and returns a meaningless number, values[i] is -infinity.
 * The sets of one call are factorised together (up to 64 per device pass): a sampler that proposes for all its walkers at
 * once should pass them in one call. */
typedef struct moe_ll moe_ll_t;
int moe_ll_create(int cov_type, const double* points_sampled, const double* points_sampled_value, const int* derivatives,
                  int num_derivatives, int dim, int num_sampled, int device, moe_ll_t** ll_out, moe_error_t* err);
int moe_ll_destroy(moe_ll_t* ll);
int moe_ll_evaluate(moe_ll_t* ll, const double* hyperparameters_all, int num_sets, double* values, moe_error_t* err);
/* compute_hyperparameter_grad_log_likelihood (gpp_python_model_selection.cpp:88-135 ->
 * LogMarginalLikelihoodEvaluator::ComputeGradLogLikelihood, gpp_model_selection.cpp:629-677):
 * grad[1 + dim + 1 + g] = d log p(y | X, theta) / d (alpha, lengths[dim], noise_variance[1 + g]) at ONE hyper-parameter set
 * (layout as above) = 1/2 a^T (dK/dtheta) a - 1/2 tr(K^-1 dK/dtheta), a = K^-1 (y - mean).  The Matern-5/2 kernel follows
 * the reference's convention that only the function-value block of K depends on (alpha, lengths)
 * (MaternNu2p5::HyperparameterGradCovariance, gpp_covariance.cpp:461-487, fills that block alone); the squared
 * exponential is provided without derivative observations.  A singular K + noise is MOE_ERR_SINGULAR. */
int moe_ll_grad(moe_ll_t* ll, const double* hyperparameters, double* grad, moe_error_t* err);
/* r5 -- multistart_hyperparameter_optimization / restarted_hyperparameter_optimization (gpp_python_model_selection.cpp:428-474 ->
 * MultistartGradientDescentHyperparameterOptimization / RestartedGradientDescentHyperparameterOptimizationTensor,
 * gpp_model_selection.hpp:967-1103) from caller-supplied initial guesses: the maximum-likelihood hyper-parameters by restarted gradient
 * ascent (GradientDescentOptimizer, gpp_optimization.hpp:619-705, 1144-1185) on log p(y | X, theta) over
 * theta = (alpha, lengths[dim], noise_variance[1 + g]) in LINEAR space, inside the tensor-product domain domain_log10[n_hyper][2] given
 * in LOG-10 space (as the reference's boundary takes it).  initial_guesses[num_starts][n_hyper] are linear-space points (the reference
 * draws a Latin hypercube in log space and exponentiates, :841-858; moe_latin_hypercube reproduces the generator).  As in the reference
 * the best initial guess seeds the result (InitializeBestKnownPoint, :911-930) and *found reports whether an optimised end point beat
 * it.  num_starts = 1 is the restarted (single-start) optimiser.  Every ascent step evaluates the gradients of all running starts. */
int moe_ll_multistart(moe_ll_t* ll, const moe_gd_params_t* gd_params, const double* domain_log10, const double* initial_guesses,
                      int num_starts, double* best_hyperparameters, double* best_value, int* found, moe_error_t* err);
/* restarted_hyperparameter_optimization (RestartedGradientDescentHyperparameterOptimizationTensor, gpp_model_selection.hpp:989-1012):
 * the point the restarted ascent from x0[n_hyper] (linear space) ENDS at -- the reference returns the state's current point, whether
 * or not it improved on the start. */
int moe_ll_ascend(moe_ll_t* ll, const moe_gd_params_t* gd_params, const double* domain_log10, const double* x0, double* end_point,
                  moe_error_t* err);

/* ---- covariance assembly (exposed for parity tests and the HBM-roofline measurement) ----
 * BuildMixCovarianceMatrix (gpp_math.cpp:309-335, 469-479): out[N x num_pts*(1+g2)] col-major = K(X, pts) with
 * derivative blocks; derivs2[g2] are the derivative observations carried by `pts`. */
int moe_gp_mix_covariance(const moe_gp_t* gp, const double* pts, int num_pts, const int* derivs2, int g2, double* out,
                          moe_error_t* err);
/* Timing probe: builds K(X, pts) [N x num_pts] on device `repeat` times without copying it back; returns the average
 * kernel milliseconds (HIP events) and the algorithmic bytes per launch. */
int moe_cov_build_probe(const moe_gp_t* gp, const double* pts, int num_pts, int repeat, double* avg_ms,
                        double* bytes_per_launch, moe_error_t* err);

/* Timing probe of the GP's own covariance assembly: K(X, X) + noise on the diagonal, N x N with the derivative-observation
 * blocks (BuildCovarianceMatrixWithNoiseVariance, gpp_math.cpp:391-455) -- the launch GaussianProcess construction makes --
 * `repeat` times into a scratch matrix; average kernel ms (HIP events) and the algorithmic bytes 8 [2 n d + N^2]. */
int moe_kxx_build_probe(const moe_gp_t* gp, int repeat, double* avg_ms, double* bytes_per_launch, moe_error_t* err);
/* What the chip sustains on nothing but dependent FP64 FMA chains (8 per lane, 16 wavefronts per CU), TFLOP/s: the rate the
 * FP64-bound kernels can be held against next to the 78.6 TFLOP/s of the data sheet (the clock does not hold 2.4 GHz under
 * FP64 load). */
int moe_debug_fp64_rate(int device, double* tflops, moe_error_t* err);

/* Parity probe of the device factorisation used by moe_gp_create: factors the SPD matrix a[n*n] (column-major, lower
 * triangle read) with the blocked device Cholesky that replaces ComputeCholeskyFactorL (gpp_linear_algebra.cpp:109-148);
 * writes the factor to chol[n*n] (strict upper = 0) and its explicit inverse to chol_inv[n*n]; *info = 0 or the failing
 * leading-minor index (pivot <= 1e-16), in which case MOE_ERR_SINGULAR is returned. */
int moe_debug_cholesky(int n, const double* a, int device, double* chol, double* chol_inv, int* info, moe_error_t* err);

/* Accuracy probe of the device exp / sqrt used inside the covariance loops (csrc/fastmath.hpp):
 * exp_neg[i] = exp(-x[i]), sqrt_out[i] = sqrt(x[i]) for x[i] >= 0. */
int moe_debug_math(int n, const double* x, int device, double* exp_neg, double* sqrt_out, moe_error_t* err);

/* Timing of the last moe_kg / moe_kg_batch call's dominant kernels, measured with HIP events on the library's stream:
 * out[0] = MC inner-optimisation kernel ms, out[1] = N x M covariance-build ms, out[2] = tail contraction ms,
 * out[3] = state set-up ms, out[4] = total device ms. */
int moe_last_kernel_ms(const moe_gp_t* gp, double* out5);
/* Which Monte-Carlo kernel the last moe_kg* call on this handle launched (diagnostics; the tests use it to assert that a
 * fixture really exercised the path it was built for): out[0] = variant (0 wave-per-sample, 1 workgroup-per-sample, 2 streamed-weights wave-per-sample),
 * out[1] = bit 0: coordinate table in LDS; bit 1: FAR FRAME -- the training set, the points being sampled or the inner domain box
 *          span more than 100 length scales from the training-set mean, so the evaluation took the direct-difference kernels with
 *          single-trial passes (exact, slower: typical triggers are the short length scales of a hyper-parameter MCMC ensemble or a
 *          search box far wider than the data); bit 2: coordinates streamed from L2 (far frame, or d > 16); bit 3 (r5): the lane-parked
 *          form of the LDS-table kernel (kg_mc_lane.hpp; MOE_KG_LANE=0 selects the round-4 form -- same results bit for bit),
 * out[2] = wavefronts per workgroup, out[3] = register tiles per wavefront (variant 1) or
 * leading tiles of the paired-row table kept in LDS (variant 0, d > 16),
 * out[4] = streamed per-sample weight table, out[5] = T-free gradient tail, out[6] = workgroups, out[7] = sample pre-pass. */
int moe_last_kernel_info(const moe_gp_t* gp, int* out8);

#ifdef __cplusplus
}
#endif
#endif /* MOE_HIP_H_ */
