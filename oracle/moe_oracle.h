/* oracle/moe_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the Cornell-MOE GP-posterior + Monte-Carlo acquisition hot path
 * (moe/optimal_learning/cpp/{gpp_covariance,gpp_linear_algebra,gpp_math,gpp_knowledge_gradient_optimization}.cpp,
 * gpp_optimization.hpp, gpp_domain.cpp).  It follows the reference ALGORITHM (including the per-sample fantasy-GP
 * re-solve and the exact control flow of the inner line-search gradient descent), not the device algorithm.
 *
 * Pinning: oracle/_ref (the unmodified reference compiled in place) is the authority; tests/test_oracle_vs_ref.py
 * checks every function here against it, and tests/golden/ holds vectors generated from it (tools/make_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this library; the product path
 * (cornell_moe_amd/) never does.
 *
 * Conventions are the reference's: FP64, matrices column-major (A[j*rows+i]), points [point][dim] contiguous,
 * derivative-observation index lists `derivs[g]`, block index 0 = function value, 1+a = d/dx_{derivs[a]}.
 */
#ifndef MOE_ORACLE_H_
#define MOE_ORACLE_H_

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_COV_SQUARE_EXPONENTIAL 0
#define ORC_COV_MATERN_NU_2P5 1
#define ORC_MAX_DIM 64

typedef struct orc_cov {
  int type;
  int dim;
  double alpha;
  double lengths_sq[ORC_MAX_DIM];
} orc_cov;

typedef struct orc_gp orc_gp;

/* gpp_covariance.cpp:121-164 / 339-387 */
void orc_covariance(const orc_cov* c, const double* p1, const int* d1, int g1, const double* p2, const int* d2, int g2,
                    double* cov);
/* gpp_covariance.cpp:171-234 / 389-459 */
void orc_grad_covariance(const orc_cov* c, const double* p1, const int* d1, int g1, const double* p2, const int* d2, int g2,
                         double* grad_cov);

/* gpp_linear_algebra.cpp:109-148; returns 0 or (failing pivot index + 1) */
int orc_cholesky(int n, double* a);
/* gpp_linear_algebra.cpp:160-187 */
void orc_tri_solve(const double* L, char trans, int n, int lda, double* x);
/* gpp_linear_algebra.hpp:220-250 */
void orc_chol_solve(const double* L, int n, double* b);

/* gpp_math.cpp:553-573, 481-511.  Returns NULL if K is singular (pivot <= 1e-16). */
orc_gp* orc_gp_create(int cov_type, double alpha, const double* lengths, const double* X, const double* y,
                      const double* noise, const int* derivs, int g, int d, int n);
void orc_gp_destroy(orc_gp* gp);
int orc_gp_N(const orc_gp* gp);
void orc_gp_dump(const orc_gp* gp, double* K_chol, double* K_inv_y, double* mean);
/* gpp_math.cpp:309-335 (member :469-479): K(X, pts) [N x k(1+g2)] */
void orc_gp_mix_cov(const orc_gp* gp, const double* pts, int k, const int* d2, int g2, double* out);
/* gpp_math.cpp:688-710 */
void orc_gp_additional_mean(const orc_gp* gp, const double* pts, int k, const int* d2, int g2, double* out);
/* gpp_math.cpp:728-757; out[d * k(1+g2)] */
void orc_gp_grad_additional_mean(const orc_gp* gp, const double* pts, int k, const int* d2, int g2, double* out);
/* Python-boundary queries (gpp_python_gaussian_process.cpp:64-236): points carry the GP's own derivative list. */
void orc_gp_mean(const orc_gp* gp, const double* pts, int k, double* out);            /* function values only */
void orc_gp_grad_mean(const orc_gp* gp, const double* pts, int k, double* out);       /* [d][k(1+g)] */
void orc_gp_var(const orc_gp* gp, const double* pts, int k, double* out);             /* [m][m], m=k(1+g) */
int orc_gp_chol_var(const orc_gp* gp, const double* pts, int k, double* out);         /* in-place chol of var */
void orc_gp_grad_var(const orc_gp* gp, const double* pts, int k, int nd, double* out); /* [nd][m][m][d] */
int orc_gp_grad_chol_var(const orc_gp* gp, const double* pts, int k, int nd, double* out);

/* gpp_math.cpp:1991-2126.  normals[M][q+p].  grad may be NULL.  Returns 0 / leading-minor index on singular variance. */
int orc_ei(const orc_gp* gp, const double* Xq, const double* Xp, int q, int p, int M, double best_so_far,
           const double* normals, double* ei, double* grad);

/* log marginal likelihood (gpp_model_selection.cpp:540-612; +1e-6 diagonal jitter like the reference).  rc != 0: K singular. */
int orc_log_likelihood(int cov_type, double alpha, const double* lengths, const double* X, const double* y,
                       const double* noise, const int* derivs, int g, int d, int n, double* value);
/* d log p / d (alpha, lengths[d], noise[1 + g]) -> grad[1 + d + 1 + g]; rc as orc_log_likelihood, -2 = not restated. */
int orc_log_likelihood_grad(int cov_type, double alpha, const double* lengths, const double* X, const double* y,
                            const double* noise, const int* derivs, int g, int d, int n, double* grad);

/* analytic 1,0-EI and its gradient [dim] (gpp_math.cpp:2195-2259); either output may be NULL. */
int orc_ei_analytic(const orc_gp* gp, const double* pt, double best_so_far, double* ei, double* grad);

/* gpp_knowledge_gradient_optimization.cpp:69-227 (+ :420-472, gpp_optimization.hpp:708-828, 1242-1283, gpp_domain.cpp:64-105).
 * gd[8] = {num_multistarts, max_num_steps, max_num_restarts, num_steps_averaged, gamma, pre_mult, max_relative_change,
 * tolerance}; bounds[2*(d-f)]; discrete[P][d-f]; normals[ceil(M/2)][m] (antithetic pairs); grad / best_point may be NULL.
 * counters (may be NULL): [0] posterior-mean evaluations, [1] gradient evaluations. */
int orc_kg(const orc_gp* gp, int num_fidelity, const double* gd, const double* bounds, const double* discrete, int P,
           const double* Xq, const double* Xp, int q, int p, int M, double best_so_far, const double* normals,
           int want_grad, double* kg, double* grad, double* best_point, long* counters);
/* orc_kg on a state built at head_q and then moved to Xq with SetCurrentPoint (the reference's multistart drivers): the
 * discretised set keeps head_q's points.  head_q == NULL: orc_kg. */
int orc_kg_head(const orc_gp* gp, int num_fidelity, const double* gd, const double* bounds, const double* discrete, int P,
           const double* Xq, const double* Xp, int q, int p, int M, double best_so_far, const double* normals,
           int want_grad, double* kg, double* grad, double* best_point, long* counters, const double* head_q);

#ifdef __cplusplus
}
#endif
#endif /* MOE_ORACLE_H_ */
