/* oracle/moe_oracle.c -- TEST INFRASTRUCTURE ONLY (see moe_oracle.h).
 *
 * Plain-C99 restatement of the reference algorithm for the GP-posterior + MC-acquisition hot path.
 * Every function cites the reference file:line (under moe/optimal_learning/cpp/) it follows.
 * Pinned against oracle/_ref (the unmodified reference) by tests/test_oracle_vs_ref.py and tests/golden/.
 */
#include "moe_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define SQ(x) ((x) * (x))
static const double kSqrt5 = 2.236067977499789696409173668731276235;

static double* dalloc(size_t n) { return (double*)calloc(n ? n : 1, sizeof(double)); }

/* ------------------------------------------------------------------------------------------------------------------
 * Covariance (gpp_covariance.cpp)
 * ---------------------------------------------------------------------------------------------------------------- */

/* gpp_covariance.cpp:46-56 NormSquaredWithInverseWeights */
static double norm_sq(const orc_cov* c, const double* p1, const double* p2) {
  double norm = 0.0;
  for (int i = 0; i < c->dim; ++i) norm += SQ(p1[i] - p2[i]) / c->lengths_sq[i];
  return norm;
}

/* SquareExponential::Covariance gpp_covariance.cpp:121-164; MaternNu2p5::Covariance :339-387 */
void orc_covariance(const orc_cov* c, const double* p1, const int* d1, int g1, const double* p2, const int* d2, int g2,
                    double* cov) {
  const double r2 = norm_sq(c, p1, p2);
  double u[ORC_MAX_DIM], v[ORC_MAX_DIM];
  double base, first, second;
  if (c->type == ORC_COV_SQUARE_EXPONENTIAL) {
    base = c->alpha * exp(-0.5 * r2);
    first = base;
    second = base;
  } else {
    const double arg = kSqrt5 * sqrt(r2);
    const double e = exp(-arg);
    base = c->alpha * e * (1.0 + arg + 5.0 / 3.0 * r2);
    first = 5.0 / 3.0 * c->alpha * e * (arg + 1.0);
    second = 25.0 / 3.0 * c->alpha * e;
  }
  cov[0] = base;
  for (int m = 0; m < g1; ++m) {
    u[m] = (p2[d1[m]] - p1[d1[m]]) / c->lengths_sq[d1[m]];
    cov[m + 1] = first * u[m];
  }
  for (int n = 0; n < g2; ++n) {
    v[n] = (p1[d2[n]] - p2[d2[n]]) / c->lengths_sq[d2[n]];
    cov[(n + 1) * (1 + g1)] = first * v[n];
  }
  for (int i = 0; i < g1; ++i) {
    for (int j = 0; j < g2; ++j) {
      double val = u[i] * v[j] * second;
      if (d1[i] == d2[j]) val += first / c->lengths_sq[d2[j]];
      cov[(i + 1) + (j + 1) * (1 + g1)] = val;
    }
  }
}

/* SquareExponential::GradCovariance gpp_covariance.cpp:171-234; MaternNu2p5::GradCovariance :389-459.
 * grad_cov[i + a*dim + b*dim*(1+g1)] = d cov[a,b] / d p1_i */
void orc_grad_covariance(const orc_cov* c, const double* p1, const int* d1, int g1, const double* p2, const int* d2, int g2,
                         double* gc) {
  const int dim = c->dim;
  const double r2 = norm_sq(c, p1, p2);
  double u[ORC_MAX_DIM], v[ORC_MAX_DIM];
  for (int m = 0; m < g1; ++m) u[m] = (p2[d1[m]] - p1[d1[m]]) / c->lengths_sq[d1[m]];
  for (int n = 0; n < g2; ++n) v[n] = (p1[d2[n]] - p2[d2[n]]) / c->lengths_sq[d2[n]];
  if (c->type == ORC_COV_SQUARE_EXPONENTIAL) {
    const double kernel = c->alpha * exp(-0.5 * r2);
    for (int i = 0; i < dim; ++i) {
      const double di = (p2[i] - p1[i]) / c->lengths_sq[i];
      gc[i] = di * kernel;
      for (int m = 0; m < g1; ++m) {
        gc[i + (m + 1) * dim] = di * u[m] * kernel;
        if (i == d1[m]) gc[i + (m + 1) * dim] -= kernel / c->lengths_sq[d1[m]];
      }
      for (int n = 0; n < g2; ++n) {
        gc[i + (n + 1) * dim * (g1 + 1)] = di * v[n] * kernel;
        if (i == d2[n]) gc[i + (n + 1) * dim * (g1 + 1)] += kernel / c->lengths_sq[d2[n]];
      }
      for (int m = 0; m < g1; ++m) {
        for (int n = 0; n < g2; ++n) {
          double t = u[m] * v[n];
          if (d1[m] == d2[n]) t += 1.0 / c->lengths_sq[d1[m]];
          t *= di;
          if (d1[m] == i) t -= v[n] / c->lengths_sq[d1[m]];
          if (d2[n] == i) t += u[m] / c->lengths_sq[d2[n]];
          gc[i + (m + 1) * dim + (n + 1) * dim * (g1 + 1)] = t * kernel;
        }
      }
    }
  } else {
    const double arg = kSqrt5 * sqrt(r2);
    const double e = exp(-arg);
    const double first = 5.0 / 3.0 * c->alpha * e * (arg + 1.0);
    const double second = 25.0 / 3.0 * c->alpha * e;
    for (int i = 0; i < dim; ++i) {
      const double di = (p2[i] - p1[i]) / c->lengths_sq[i];
      gc[i] = di * first;
      for (int m = 0; m < g1; ++m) {
        gc[i + (m + 1) * dim] = second * di * u[m];
        if (i == d1[m]) gc[i + (m + 1) * dim] -= first / c->lengths_sq[d1[m]];
      }
      for (int n = 0; n < g2; ++n) {
        gc[i + (n + 1) * dim * (g1 + 1)] = second * di * v[n];
        if (i == d2[n]) gc[i + (n + 1) * dim * (g1 + 1)] += first / c->lengths_sq[d2[n]];
      }
      for (int m = 0; m < g1; ++m) {
        for (int n = 0; n < g2; ++n) {
          double t = 0.0;
          if (r2 > 0.0) {
            t = second * u[m] * v[n];
            t *= kSqrt5 * di / sqrt(r2);
            if (d1[m] == i) t -= second * v[n] / c->lengths_sq[d1[m]];
            if (d2[n] == i) t += second * u[m] / c->lengths_sq[d2[n]];
            if (d1[m] == d2[n]) t += second * di / c->lengths_sq[d1[m]];
          }
          gc[i + (m + 1) * dim + (n + 1) * dim * (g1 + 1)] = t;
        }
      }
    }
  }
}

/* ------------------------------------------------------------------------------------------------------------------
 * Linear algebra (gpp_linear_algebra.cpp)
 * ---------------------------------------------------------------------------------------------------------------- */

/* ComputeCholeskyFactorL gpp_linear_algebra.cpp:109-148 (outer-product form, pivot threshold 1e-16) */
int orc_cholesky(int n, double* a) {
  for (int k = 0; k < n; ++k) {
    double* col = a + (size_t)k * n;
    if (col[k] > 1.0e-16) {
      const double akk = sqrt(col[k]);
      col[k] = akk;
      for (int j = k + 1; j < n; ++j) col[j] /= akk;
      for (int j = k + 1; j < n; ++j) {
        double* cj = a + (size_t)j * n;
        const double ljk = col[j];
        for (int i = j; i < n; ++i) cj[i] = cj[i] - col[i] * ljk;
      }
    } else {
      return k + 1;
    }
  }
  return 0;
}

/* TriangularMatrixVectorSolve gpp_linear_algebra.cpp:160-187 */
void orc_tri_solve(const double* A, char trans, int n, int lda, double* x) {
  if (trans == 'N') {
    for (int j = 0; j < n; ++j) {
      if (x[j] != 0.0) {
        x[j] /= A[j];
        const double t = x[j];
        for (int i = j + 1; i < n; ++i) x[i] = x[i] - t * A[i];
      }
      A += lda;
    }
  } else {
    A += (size_t)lda * (n - 1);
    for (int j = n - 1; j >= 0; --j) {
      double t = x[j];
      for (int i = n - 1; i >= j + 1; --i) t -= A[i] * x[i];
      t /= A[j];
      x[j] = t;
      A -= lda;
    }
  }
}

static void tri_solve_mat(const double* A, char trans, int n, int ncols, double* X) {
  for (int k = 0; k < ncols; ++k) orc_tri_solve(A, trans, n, n, X + (size_t)k * n);
}

/* CholeskyFactorLMatrixVectorSolve gpp_linear_algebra.hpp:220-250 */
void orc_chol_solve(const double* L, int n, double* b) {
  orc_tri_solve(L, 'N', n, n, b);
  orc_tri_solve(L, 'T', n, n, b);
}

static void chol_solve_mat(const double* L, int n, int ncols, double* B) {
  for (int k = 0; k < ncols; ++k) orc_chol_solve(L, n, B + (size_t)k * n);
}

/* TriangularMatrixVectorMultiply('N') gpp_linear_algebra.cpp:257-273, including its quirk: the column pointer is only
 * decremented when x[j] != 0. */
static void tri_mat_vec_N(const double* A, int n, double* x) {
  A += (size_t)n * (n - 1);
  for (int j = n - 1; j >= 0; --j) {
    if (x[j] != 0.0) {
      const double t = x[j];
      for (int i = n - 1; i >= j + 1; --i) x[i] += t * A[i];
      x[j] *= A[j];
      A -= n;
    }
  }
}

/* y = alpha * op(A) x + beta * y  (GeneralMatrixVectorMultiply gpp_linear_algebra.cpp:340-372) */
static void gemv(const double* A, char trans, const double* x, double alpha, double beta, int m, int n, int lda, double* y) {
  if (trans == 'N') {
    for (int i = 0; i < m; ++i) y[i] *= beta;
    for (int j = 0; j < n; ++j) {
      const double t = alpha * x[j];
      for (int i = 0; i < m; ++i) y[i] += A[i + (size_t)j * lda] * t;
    }
  } else {
    for (int j = 0; j < n; ++j) {
      double t = 0.0;
      for (int i = 0; i < m; ++i) t += A[i + (size_t)j * lda] * x[i];
      y[j] = beta * y[j] + alpha * t;
    }
  }
}

/* C = alpha * op(A) B + beta * C; A is (m x k) ['N'] or stored (k x m) ['T']; B (k x n); C (m x n)
 * (GeneralMatrixMatrixMultiply gpp_linear_algebra.cpp:374-398) */
static void gemm(const double* A, char transA, const double* B, double alpha, double beta, int m, int k, int n, double* C) {
  for (int j = 0; j < n; ++j) {
    for (int i = 0; i < m; ++i) {
      double t = 0.0;
      if (transA == 'N') {
        for (int l = 0; l < k; ++l) t += A[i + (size_t)l * m] * B[l + (size_t)j * k];
      } else {
        for (int l = 0; l < k; ++l) t += A[l + (size_t)i * k] * B[l + (size_t)j * k];
      }
      C[i + (size_t)j * m] = beta * C[i + (size_t)j * m] + alpha * t;
    }
  }
}

/* VectorNorm gpp_linear_algebra.cpp:53-72 */
static double vector_norm(const double* v, int n) {
  if (n == 1) return fabs(v[0]);
  double scale = 0.0, scaled = 1.0;
  for (int i = 0; i < n; ++i) {
    if (v[i] != 0.0) {
      const double a = fabs(v[i]);
      if (scale < a) {
        const double t = scale / a;
        scaled = 1.0 + scaled * (t * t);
        scale = a;
      } else {
        const double t = a / scale;
        scaled += t * t;
      }
    }
  }
  return scale * sqrt(scaled);
}

/* ------------------------------------------------------------------------------------------------------------------
 * Gaussian process (gpp_math.cpp)
 * ---------------------------------------------------------------------------------------------------------------- */

struct orc_gp {
  orc_cov cov;
  int dim, n, g, N;
  int derivs[ORC_MAX_DIM];
  double* X;      /* [n][dim] */
  double* y;      /* [n][1+g] */
  double* noise;  /* [1+g] */
  double* K_chol; /* [N][N] col-major, lower */
  double* K_inv_y;
  double mean;
};

/* BuildMixCovarianceMatrix gpp_math.cpp:309-335: rows over A (with gA blocks), cols over B (with gB blocks) */
static void build_mix(const orc_cov* c, const double* A, const double* B, int nA, int nB, const int* dA, int gA,
                      const int* dB, int gB, double* out) {
  double blk[(1 + ORC_MAX_DIM) * (1 + ORC_MAX_DIM)];
  const size_t rows = (size_t)nA * (gA + 1);
  for (int j = 0; j < nB; ++j) {
    for (int i = 0; i < nA; ++i) {
      orc_covariance(c, A + (size_t)i * c->dim, dA, gA, B + (size_t)j * c->dim, dB, gB, blk);
      for (int m = 0; m < gA + 1; ++m)
        for (int n = 0; n < gB + 1; ++n)
          out[(size_t)(i * (gA + 1) + m) + (size_t)(j * (gB + 1) + n) * rows] = blk[m + n * (gA + 1)];
    }
  }
}

/* BuildCovarianceMatrixWithNoiseVariance gpp_math.cpp:426-455 (lower triangle) */
static void build_K_with_noise(const orc_gp* gp, double* K) {
  double blk[(1 + ORC_MAX_DIM) * (1 + ORC_MAX_DIM)];
  const int g = gp->g;
  const size_t N = gp->N;
  for (int i = 0; i < gp->n; ++i) {
    for (int j = i; j < gp->n; ++j) {
      orc_covariance(&gp->cov, gp->X + (size_t)j * gp->dim, gp->derivs, g, gp->X + (size_t)i * gp->dim, gp->derivs, g, blk);
      for (int m = 0; m < g + 1; ++m) {
        for (int n = 0; n < g + 1; ++n) {
          const size_t row = (size_t)j * (g + 1) + m, col = (size_t)i * (g + 1) + n;
          if (row >= col) K[row + col * N] = blk[m + n * (g + 1)];
          if (row == col) K[row + col * N] += gp->noise[m];
        }
      }
    }
  }
}

/* RecomputeMeanVariables gpp_math.cpp:531-551 */
static void recompute_mean_variables(orc_gp* gp, int mean_change) {
  if (mean_change) {
    gp->mean = 0.0;
    for (int i = 0; i < gp->n; ++i) gp->mean += gp->y[(size_t)i * (gp->g + 1)];
    gp->mean /= gp->n;
  }
  memcpy(gp->K_inv_y, gp->y, sizeof(double) * gp->N);
  for (int i = 0; i < gp->n; ++i) gp->K_inv_y[(size_t)i * (gp->g + 1)] -= gp->mean;
  orc_chol_solve(gp->K_chol, gp->N, gp->K_inv_y);
}

/* RecomputeCholeskyVariables gpp_math.cpp:513-529 */
static int recompute_cholesky_variables(orc_gp* gp) {
  memset(gp->K_chol, 0, sizeof(double) * (size_t)gp->N * gp->N);
  build_K_with_noise(gp, gp->K_chol);
  return orc_cholesky(gp->N, gp->K_chol);
}

static orc_gp* gp_alloc(const orc_cov* cov, int g, const int* derivs, int d, int n) {
  orc_gp* gp = (orc_gp*)calloc(1, sizeof(orc_gp));
  gp->cov = *cov;
  gp->dim = d;
  gp->n = n;
  gp->g = g;
  gp->N = n * (1 + g);
  for (int i = 0; i < g; ++i) gp->derivs[i] = derivs[i];
  gp->X = dalloc((size_t)n * d);
  gp->y = dalloc((size_t)gp->N);
  gp->noise = dalloc((size_t)1 + g);
  gp->K_chol = dalloc((size_t)gp->N * gp->N);
  gp->K_inv_y = dalloc((size_t)gp->N);
  return gp;
}

/* LogMarginalLikelihoodEvaluator::FillLogLikelihoodState + ComputeLogLikelihood (gpp_model_selection.cpp:540-612):
 * K + noise, +1e-6 on the diagonal, Cholesky (a failing pivot is ignored by the reference; here it returns rc != 0 and
 * *value is left untouched), y centred by the mean of the function values, LML = -1/2 y^T K^-1 y - sum log L_ii - N/2 log 2pi. */
int orc_log_likelihood(int cov_type, double alpha, const double* lengths, const double* X, const double* y,
                       const double* noise, const int* derivs, int g, int d, int n, double* value) {
  orc_cov cov;
  cov.type = cov_type;
  cov.dim = d;
  cov.alpha = alpha;
  for (int i = 0; i < d; ++i) cov.lengths_sq[i] = SQ(lengths[i]);
  orc_gp* gp = gp_alloc(&cov, g, derivs, d, n);
  memcpy(gp->X, X, sizeof(double) * (size_t)n * d);
  memcpy(gp->y, y, sizeof(double) * (size_t)gp->N);
  memcpy(gp->noise, noise, sizeof(double) * (size_t)(1 + g));
  const size_t N = (size_t)gp->N;
  memset(gp->K_chol, 0, sizeof(double) * N * N);
  build_K_with_noise(gp, gp->K_chol);
  for (size_t i = 0; i < N; ++i) gp->K_chol[i + i * N] += 1.0e-6;
  const int rc = orc_cholesky((int)N, gp->K_chol);
  if (rc == 0) {
    recompute_mean_variables(gp, 1); /* K_inv_y = K^-1 (y - mean on the value rows) */
    double term1 = 0.0, term2 = 0.0;
    for (size_t i = 0; i < N; ++i) {
      const double yc = gp->y[i] - ((i % (size_t)(g + 1)) == 0 ? gp->mean : 0.0);
      term1 += yc * gp->K_inv_y[i];
      term2 -= log(gp->K_chol[i + i * N]);
    }
    *value = -0.5 * term1 + term2 - 0.5 * (double)N * 1.8378770664093454835607;
  }
  orc_gp_destroy(gp);
  return rc;
}

/* CovarianceInterface::HyperparameterGradCovariance for the function-value entry: out[1 + d] = d cov(p1, p2) / d (alpha,
 * lengths).  SquareExponential gpp_covariance.cpp:245-259; MaternNu2p5 :461-487 (which fills nothing but this entry: with
 * derivative observations the other blocks of the matrix below stay at the zeros their buffer was created with,
 * gpp_model_selection.cpp:398). */
static void hyper_grad_cov_value(const orc_cov* c, const double* p1, const double* p2, double* out) {
  const double nsq = norm_sq(c, p1, p2);
  if (c->type == 0) {
    const double cov = c->alpha * exp(-0.5 * nsq);
    out[0] = cov / c->alpha;
    for (int i = 0; i < c->dim; ++i) {
      const double len = sqrt(c->lengths_sq[i]);
      out[1 + i] = cov * SQ((p1[i] - p2[i]) / len) / len;
    }
    return;
  }
  if (nsq == 0.0) {
    out[0] = 1.0;
    for (int i = 0; i < c->dim; ++i) out[1 + i] = 0.0;
    return;
  }
  const double matern_arg = kSqrt5 * sqrt(nsq);
  const double poly_part = matern_arg + 5.0 / 3.0 * nsq;
  const double exp_part = exp(-matern_arg);
  out[0] = (1.0 + poly_part) * exp_part;
  for (int i = 0; i < c->dim; ++i) {
    const double len = sqrt(c->lengths_sq[i]);
    const double dr2_dleni = -2.0 * SQ((p1[i] - p2[i]) / len) / len;
    const double dr_dleni = 0.5 * dr2_dleni / sqrt(nsq);
    out[1 + i] = c->alpha * exp_part * (5.0 / 3.0 * dr2_dleni - poly_part * kSqrt5 * dr_dleni);
  }
}

/* LogMarginalLikelihoodEvaluator::ComputeGradLogLikelihood (gpp_model_selection.cpp:629-677) on the state of
 * orc_log_likelihood: for every hyper-parameter theta in (alpha, lengths[d], noise variances[1 + g]) the matrix dK/dtheta
 * (BuildHyperparameterGradCovarianceMatrix, :386-446), then  grad = 1/2 a^T dK a - 1/2 tr(K^-1 dK),  a = K^-1 (y - mean),
 * with K^-1 dK by two triangular solves per column as the reference does (OL_USE_INVERSE 0).
 * The squared exponential is restated for g = 0 only (returns -2 otherwise). */
int orc_log_likelihood_grad(int cov_type, double alpha, const double* lengths, const double* X, const double* y,
                            const double* noise, const int* derivs, int g, int d, int n, double* grad) {
  if (cov_type == 0 && g > 0) return -2;
  orc_cov cov;
  cov.type = cov_type;
  cov.dim = d;
  cov.alpha = alpha;
  for (int i = 0; i < d; ++i) cov.lengths_sq[i] = SQ(lengths[i]);
  orc_gp* gp = gp_alloc(&cov, g, derivs, d, n);
  memcpy(gp->X, X, sizeof(double) * (size_t)n * d);
  memcpy(gp->y, y, sizeof(double) * (size_t)gp->N);
  memcpy(gp->noise, noise, sizeof(double) * (size_t)(1 + g));
  const size_t N = (size_t)gp->N;
  const int g1 = 1 + g, nh = 1 + d + g1;
  memset(gp->K_chol, 0, sizeof(double) * N * N);
  build_K_with_noise(gp, gp->K_chol);
  for (size_t i = 0; i < N; ++i) gp->K_chol[i + i * N] += 1.0e-6;
  const int rc = orc_cholesky((int)N, gp->K_chol);
  if (rc == 0) {
    recompute_mean_variables(gp, 1);
    double* dK = dalloc(N * N);
    double* tmp = dalloc(N);
    double* hg = dalloc((size_t)(1 + d));
    for (int h = 0; h < nh; ++h) {
      memset(dK, 0, sizeof(double) * N * N);
      if (h < 1 + d) {
        for (int i = 0; i < n; ++i)
          for (int j = 0; j < n; ++j) {
            hyper_grad_cov_value(&cov, X + (size_t)j * d, X + (size_t)i * d, hg);
            dK[(size_t)j * g1 + (size_t)i * g1 * N] = hg[h];
          }
      } else {
        const int m = h - 1 - d;
        for (int i = 0; i < n; ++i) {
          const size_t r = (size_t)i * g1 + m;
          dK[r + r * N] = 1.0;
        }
      }
      gemv(dK, 'N', gp->K_inv_y, 1.0, 0.0, (int)N, (int)N, (int)N, tmp);
      double quad = 0.0;
      for (size_t i = 0; i < N; ++i) quad += gp->K_inv_y[i] * tmp[i];
      chol_solve_mat(gp->K_chol, (int)N, (int)N, dK);
      double tr = 0.0;
      for (size_t i = 0; i < N; ++i) tr += dK[i + i * N];
      grad[h] = 0.5 * quad - 0.5 * tr;
    }
    free(dK);
    free(tmp);
    free(hg);
  }
  orc_gp_destroy(gp);
  return rc;
}

/* GaussianProcess ctor gpp_math.cpp:553-573 + RecomputeDerivedVariables :481-511 */
orc_gp* orc_gp_create(int cov_type, double alpha, const double* lengths, const double* X, const double* y,
                      const double* noise, const int* derivs, int g, int d, int n) {
  orc_cov cov;
  cov.type = cov_type;
  cov.dim = d;
  cov.alpha = alpha;
  for (int i = 0; i < d; ++i) cov.lengths_sq[i] = SQ(lengths[i]); /* gpp_covariance.cpp:85-92 */
  orc_gp* gp = gp_alloc(&cov, g, derivs, d, n);
  memcpy(gp->X, X, sizeof(double) * (size_t)n * d);
  memcpy(gp->y, y, sizeof(double) * (size_t)gp->N);
  memcpy(gp->noise, noise, sizeof(double) * (size_t)(1 + g));
  if (recompute_cholesky_variables(gp) != 0) {
    orc_gp_destroy(gp);
    return NULL;
  }
  recompute_mean_variables(gp, 1);
  return gp;
}

void orc_gp_destroy(orc_gp* gp) {
  if (!gp) return;
  free(gp->X);
  free(gp->y);
  free(gp->noise);
  free(gp->K_chol);
  free(gp->K_inv_y);
  free(gp);
}

int orc_gp_N(const orc_gp* gp) { return gp->N; }

void orc_gp_dump(const orc_gp* gp, double* K_chol, double* K_inv_y, double* mean) {
  if (K_chol) memcpy(K_chol, gp->K_chol, sizeof(double) * (size_t)gp->N * gp->N);
  if (K_inv_y) memcpy(K_inv_y, gp->K_inv_y, sizeof(double) * gp->N);
  if (mean) *mean = gp->mean;
}

static void gp_mix(const orc_gp* gp, const double* pts, int k, const int* d2, int g2, double* out) {
  build_mix(&gp->cov, gp->X, pts, gp->n, k, gp->derivs, gp->g, d2, g2, out);
}

/* GaussianProcess::BuildMixCovarianceMatrix gpp_math.cpp:469-479 */
void orc_gp_mix_cov(const orc_gp* gp, const double* pts, int k, const int* d2, int g2, double* out) {
  gp_mix(gp, pts, k, d2, g2, out);
}

/* ComputeMeanOfAdditionalPoints gpp_math.cpp:688-710 */
void orc_gp_additional_mean(const orc_gp* gp, const double* pts, int k, const int* d2, int g2, double* out) {
  const int cols = k * (g2 + 1);
  double* kt = dalloc((size_t)gp->N * cols);
  gp_mix(gp, pts, k, d2, g2, kt);
  for (int i = 0; i < k; ++i)
    for (int j = 0; j < g2 + 1; ++j) out[i * (g2 + 1) + j] = (j == 0) ? gp->mean : 0.0;
  gemv(kt, 'T', gp->K_inv_y, 1.0, 1.0, gp->N, cols, gp->N, out);
  free(kt);
}

/* grad_K_star fill, gpp_math.cpp:616-637 (also :736-752): gKs[dd + row*dim + col*dim*N],
 * row = training entry, col = point-block entry of the differentiated point */
static void fill_grad_K_star(const orc_gp* gp, const double* pts, int npts, const int* d2, int g2, double* gKs) {
  const int dim = gp->dim, g = gp->g;
  double* tmp = dalloc((size_t)dim * (g2 + 1) * (g + 1));
  for (int i = 0; i < npts; ++i) {
    for (int j = 0; j < gp->n; ++j) {
      orc_grad_covariance(&gp->cov, pts + (size_t)i * dim, d2, g2, gp->X + (size_t)j * dim, gp->derivs, g, tmp);
      for (int m = 0; m < g2 + 1; ++m)
        for (int n = 0; n < g + 1; ++n) {
          const size_t row = (size_t)n + (size_t)j * (g + 1), col = (size_t)m + (size_t)i * (g2 + 1);
          for (int d = 0; d < dim; ++d)
            gKs[d + row * dim + col * dim * gp->N] = tmp[d + m * dim + n * dim * (g2 + 1)];
        }
    }
  }
  free(tmp);
}

/* SpecialTensorVectorMultiply gpp_math.cpp:357-366 */
static void special_tensor_vec(const double* tensor, const double* vec, int dim_one, int dim_two, int dim_three, double* ans) {
  for (int i = 0; i < dim_one; ++i) {
    for (int d = 0; d < dim_three; ++d) ans[d] = 0.0;
    gemv(tensor, 'N', vec, 1.0, 0.0, dim_three, dim_two, dim_three, ans);
    tensor += (size_t)dim_two * dim_three;
    ans += dim_three;
  }
}

/* ComputeGradMeanOfAdditionalPoints gpp_math.cpp:728-757 */
void orc_gp_grad_additional_mean(const orc_gp* gp, const double* pts, int k, const int* d2, int g2, double* out) {
  double* gKs = dalloc((size_t)gp->dim * gp->N * k * (g2 + 1));
  fill_grad_K_star(gp, pts, k, d2, g2, gKs);
  special_tensor_vec(gKs, gp->K_inv_y, k * (g2 + 1), gp->N, gp->dim, out);
  free(gKs);
}

/* PointsToSampleState (gpp_math.hpp:961-982) + FillPointsToSampleState (gpp_math.cpp:600-653) */
typedef struct {
  int k, gt, nd, m, precomputed, precomputed_grad;
  int gradients[ORC_MAX_DIM];
  const double* pts;
  double *K_star, *K_inv_K_star, *grad_K_star, *grad_K_inv_K_star;
} pts_state;

static void pts_state_free(pts_state* s) {
  free(s->K_star);
  free(s->K_inv_K_star);
  free(s->grad_K_star);
  free(s->grad_K_inv_K_star);
}

static void pts_state_fill(const orc_gp* gp, pts_state* s, const double* pts, int k, const int* gradients, int gt, int nd,
                           int precomputed, int precomputed_grad) {
  const int N = gp->N, dim = gp->dim;
  memset(s, 0, sizeof(*s));
  s->k = k;
  s->gt = gt;
  s->nd = nd;
  s->m = k * (gt + 1);
  s->precomputed = precomputed;
  s->precomputed_grad = precomputed_grad;
  s->pts = pts;
  for (int i = 0; i < gt; ++i) s->gradients[i] = gradients[i];
  s->K_star = dalloc((size_t)N * s->m);
  gp_mix(gp, pts, k, s->gradients, gt, s->K_star);
  if (precomputed) {
    s->K_inv_K_star = dalloc((size_t)N * s->m);
    memcpy(s->K_inv_K_star, s->K_star, sizeof(double) * (size_t)N * s->m);
    chol_solve_mat(gp->K_chol, N, s->m, s->K_inv_K_star);
  }
  if (nd > 0) {
    s->grad_K_star = dalloc((size_t)dim * N * nd * (gt + 1));
    fill_grad_K_star(gp, pts, nd, s->gradients, gt, s->grad_K_star);
    if (precomputed_grad) {
      /* gpp_math.cpp:639-651: K^-1 applied along `row` for every (col, dd) */
      const int cols = nd * (gt + 1);
      s->grad_K_inv_K_star = dalloc((size_t)dim * N * cols);
      double* tr = dalloc((size_t)N * dim);
      for (int c = 0; c < cols; ++c) {
        const double* src = s->grad_K_star + (size_t)c * N * dim;
        double* dst = s->grad_K_inv_K_star + (size_t)c * N * dim;
        for (int r = 0; r < N; ++r)
          for (int d = 0; d < dim; ++d) tr[r + (size_t)d * N] = src[d + (size_t)r * dim];
        chol_solve_mat(gp->K_chol, N, dim, tr);
        for (int r = 0; r < N; ++r)
          for (int d = 0; d < dim; ++d) dst[d + (size_t)r * dim] = tr[r + (size_t)d * N];
      }
      free(tr);
    }
  }
}

/* ComputeMeanOfPoints gpp_math.cpp:662-678 */
static void mean_of_points(const orc_gp* gp, const pts_state* s, double* out) {
  for (int i = 0; i < s->k; ++i)
    for (int j = 0; j < s->gt + 1; ++j) out[i * (s->gt + 1) + j] = (j == 0) ? gp->mean : 0.0;
  gemv(s->K_star, 'T', gp->K_inv_y, 1.0, 1.0, gp->N, s->m, gp->N, out);
}

/* ComputeGradMeanOfPoints gpp_math.cpp:721-726 */
static void grad_mean_of_points(const orc_gp* gp, const pts_state* s, double* out) {
  special_tensor_vec(s->grad_K_star, gp->K_inv_y, s->nd * (s->gt + 1), gp->N, gp->dim, out);
}

/* ComputeVarianceOfPoints gpp_math.cpp:924-970 */
static void variance_of_points(const orc_gp* gp, const pts_state* s, const int* g2list, int g2, double* var) {
  const int N = gp->N, m = s->m, m2 = s->k * (g2 + 1);
  build_mix(&gp->cov, s->pts, s->pts, s->k, s->k, s->gradients, s->gt, g2list, g2, var);
  double* part2 = dalloc((size_t)N * m2);
  gp_mix(gp, s->pts, s->k, g2list, g2, part2);
  if (!s->precomputed) {
    double* V = dalloc((size_t)N * m);
    memcpy(V, s->K_star, sizeof(double) * (size_t)N * m);
    tri_solve_mat(gp->K_chol, 'N', N, m, V);
    tri_solve_mat(gp->K_chol, 'N', N, m2, part2);
    gemm(V, 'T', part2, -1.0, 1.0, m, N, m2, var);
    free(V);
  } else {
    gemm(s->K_inv_K_star, 'T', part2, -1.0, 1.0, m, N, m2, var);
  }
  free(part2);
}

/* ComputeGradVarianceOfPointsPerPoint gpp_math.cpp:1267-1357; grad_var[d + row*dim + col*dim*m] */
static void grad_variance_per_point(const orc_gp* gp, const pts_state* s, int diff, double* gv) {
  const int dim = gp->dim, N = gp->N, gt = s->gt, m = s->m, k = s->k;
  memset(gv, 0, sizeof(double) * (size_t)dim * m * m);
  double* tgt = gv + (size_t)dim * m * diff * (gt + 1);
  for (int i = 0; i < gt + 1; ++i) {
    const int col = diff * (gt + 1) + i;
    gemm(s->grad_K_star + (size_t)col * dim * N, 'N', s->K_inv_K_star, 1.0, 0.0, dim, N, m, tgt);
    for (int j = 0; j < m * dim; ++j) tgt[j] *= -1.0;
    tgt += (size_t)m * dim;
  }
  for (int a = 0; a < gt + 1; ++a) {
    for (int b = a; b < gt + 1; ++b) {
      for (int d = 0; d < dim; ++d) {
        const size_t row = (size_t)diff * (gt + 1) + a, col = (size_t)diff * (gt + 1) + b;
        gv[d + row * dim + col * dim * m] += gv[d + col * dim + row * dim * m];
        gv[d + col * dim + row * dim * m] = gv[d + row * dim + col * dim * m];
      }
    }
  }
  double* tmp = dalloc((size_t)dim * SQ(gt + 1));
  for (int j = 0; j < k; ++j) {
    orc_grad_covariance(&gp->cov, s->pts + (size_t)diff * dim, s->gradients, gt, s->pts + (size_t)j * dim, s->gradients, gt,
                        tmp);
    for (int a = 0; a < gt + 1; ++a) {
      for (int b = 0; b < gt + 1; ++b) {
        const size_t row = (size_t)j * (gt + 1) + a, col = (size_t)diff * (gt + 1) + b;
        for (int d = 0; d < dim; ++d) {
          if (j == diff) {
            gv[d + row * dim + col * dim * m] += tmp[d + b * dim + a * dim * (gt + 1)] + tmp[d + a * dim + b * dim * (gt + 1)];
          } else {
            gv[d + row * dim + col * dim * m] += tmp[d + b * dim + a * dim * (gt + 1)];
          }
        }
      }
    }
  }
  free(tmp);
  for (int i = 0; i < gt + 1; ++i) {
    const size_t row = (size_t)diff * (gt + 1) + i;
    for (int j = 0; j < k; ++j) {
      for (int b = 0; b < gt + 1; ++b) {
        const size_t col = (size_t)j * (gt + 1) + b;
        if (j != diff)
          for (int d = 0; d < dim; ++d) gv[d + dim * row + (size_t)dim * m * col] = gv[d + dim * col + (size_t)dim * m * row];
      }
    }
  }
}

/* ComputeGradCholeskyVarianceOfPointsPerPoint gpp_math.cpp:1389-1452 (Smith 1995);
 * gc[d + c*dim + r*dim*m] = dL[r][c]/dXs_{d,diff} for r >= c */
static void grad_cholesky_per_point(const orc_gp* gp, const pts_state* s, int diff, const double* chol, double* gc) {
  const int dim = gp->dim, m = s->m;
  const double kMinimumStdDev = 2.220446049250313e-16; /* gpp_math.hpp:291 std::numeric_limits<double>::epsilon() */
  grad_variance_per_point(gp, s, diff, gc);
  for (int i = 0; i < m; ++i) {
    double* col = gc + (size_t)i * m * dim;
    for (int j = (i + 1) * dim; j < dim * m; ++j) col[j] = 0.0;
  }
#define CH(i, j) chol[(size_t)(j) * m + (i)]
#define GC(d, i, j) gc[(size_t)(j) * m * dim + (size_t)(i) * dim + (d)]
  for (int k = 0; k < m; ++k) {
    const double Lkk = CH(k, k);
    if (Lkk > kMinimumStdDev) {
      for (int d = 0; d < dim; ++d) GC(d, k, k) = 0.5 * GC(d, k, k) / Lkk;
      for (int j = k + 1; j < m; ++j)
        for (int d = 0; d < dim; ++d) GC(d, k, j) = (GC(d, k, j) - CH(j, k) * GC(d, k, k)) / Lkk;
      for (int j = k + 1; j < m; ++j)
        for (int i = j; i < m; ++i)
          for (int d = 0; d < dim; ++d) GC(d, j, i) = GC(d, j, i) - GC(d, k, i) * CH(j, k) - CH(i, k) * GC(d, k, j);
    }
  }
#undef CH
#undef GC
}

/* ---- Python-boundary queries (gpp_python_gaussian_process.cpp:64-236) ---- */

void orc_gp_mean(const orc_gp* gp, const double* pts, int k, double* out) {
  pts_state s;
  pts_state_fill(gp, &s, pts, k, NULL, 0, 0, 1, 0);
  mean_of_points(gp, &s, out);
  pts_state_free(&s);
}

void orc_gp_grad_mean(const orc_gp* gp, const double* pts, int k, double* out) {
  pts_state s;
  pts_state_fill(gp, &s, pts, k, gp->derivs, gp->g, k, 1, 0);
  grad_mean_of_points(gp, &s, out);
  pts_state_free(&s);
}

void orc_gp_var(const orc_gp* gp, const double* pts, int k, double* out) {
  pts_state s;
  pts_state_fill(gp, &s, pts, k, gp->derivs, gp->g, 0, 1, 0);
  variance_of_points(gp, &s, gp->derivs, gp->g, out);
  pts_state_free(&s);
}

int orc_gp_chol_var(const orc_gp* gp, const double* pts, int k, double* out) {
  orc_gp_var(gp, pts, k, out);
  return orc_cholesky(k * (1 + gp->g), out);
}

void orc_gp_grad_var(const orc_gp* gp, const double* pts, int k, int nd, double* out) {
  pts_state s;
  const int m = k * (1 + gp->g);
  pts_state_fill(gp, &s, pts, k, gp->derivs, gp->g, nd, 1, 0);
  for (int p = 0; p < nd; ++p) grad_variance_per_point(gp, &s, p, out + (size_t)p * gp->dim * m * m);
  pts_state_free(&s);
}

int orc_gp_grad_chol_var(const orc_gp* gp, const double* pts, int k, int nd, double* out) {
  pts_state s;
  const int m = k * (1 + gp->g);
  double* chol = dalloc((size_t)m * m);
  pts_state_fill(gp, &s, pts, k, gp->derivs, gp->g, nd, 1, 0);
  variance_of_points(gp, &s, gp->derivs, gp->g, chol);
  int rc = orc_cholesky(m, chol);
  if (rc == 0)
    for (int p = 0; p < nd; ++p) grad_cholesky_per_point(gp, &s, p, chol, out + (size_t)p * gp->dim * m * m);
  pts_state_free(&s);
  free(chol);
  return rc;
}

/* ------------------------------------------------------------------------------------------------------------------
 * q,p-EI by Monte Carlo (gpp_math.cpp:1991-2126)
 * ---------------------------------------------------------------------------------------------------------------- */
int orc_ei(const orc_gp* gp, const double* Xq, const double* Xp, int q, int p, int M, double best_so_far,
           const double* normals, double* ei_out, double* grad) {
  const int dim = gp->dim, u = q + p;
  double* U = dalloc((size_t)u * dim);
  memcpy(U, Xq, sizeof(double) * (size_t)q * dim);
  if (p > 0) memcpy(U + (size_t)q * dim, Xp, sizeof(double) * (size_t)p * dim);
  double* mu = dalloc(u);
  double* chol = dalloc((size_t)u * u);
  double* y = dalloc(u);
  int rc = 0;

  /* value: state built with configure_for_gradients=false -> num_derivatives=0, precomputed=false (:2149-2150) */
  if (ei_out) {
    pts_state s;
    pts_state_fill(gp, &s, U, u, NULL, 0, 0, 0, 0);
    mean_of_points(gp, &s, mu);
    variance_of_points(gp, &s, NULL, 0, chol);
    for (int i = 0; i < u; ++i) chol[i + (size_t)i * u] += 1.0e-6;
    rc = orc_cholesky(u, chol);
    if (rc == 0) {
      double agg = 0.0;
      for (int i = 0; i < M; ++i) {
        memcpy(y, normals + (size_t)i * u, sizeof(double) * u);
        tri_mat_vec_N(chol, u, y);
        double imp = 0.0;
        for (int j = 0; j < u; ++j) {
          const double t = best_so_far - (mu[j] + y[j]);
          if (t > imp) imp = t;
        }
        if (imp > 0.0) agg += imp;
      }
      *ei_out = agg / (double)M;
    }
    pts_state_free(&s);
  }
  /* gradient: num_derivatives = q, precomputed = true (:2134-2135) */
  if (grad && rc == 0) {
    pts_state s;
    pts_state_fill(gp, &s, U, u, NULL, 0, q, 1, 0);
    double* grad_mu = dalloc((size_t)dim * q);
    double* gchol = dalloc((size_t)dim * u * u * q);
    double* agg = dalloc((size_t)dim * q);
    mean_of_points(gp, &s, mu);
    grad_mean_of_points(gp, &s, grad_mu);
    variance_of_points(gp, &s, NULL, 0, chol);
    for (int i = 0; i < u; ++i) chol[i + (size_t)i * u] += 1.0e-6;
    rc = orc_cholesky(u, chol);
    if (rc == 0) {
      for (int k = 0; k < q; ++k) grad_cholesky_per_point(gp, &s, k, chol, gchol + (size_t)k * dim * u * u);
      for (int i = 0; i < M; ++i) {
        const double* z = normals + (size_t)i * u;
        memcpy(y, z, sizeof(double) * u);
        tri_mat_vec_N(chol, u, y);
        double imp = 0.0;
        int winner = u + 1;
        for (int j = 0; j < u; ++j) {
          const double t = best_so_far - (mu[j] + y[j]);
          if (t > imp) {
            imp = t;
            winner = j;
          }
        }
        if (imp > 0.0) {
          if (winner < q)
            for (int d = 0; d < dim; ++d) agg[winner * dim + d] -= grad_mu[winner * dim + d];
          const double* blk = gchol + (size_t)winner * dim * u;
          for (int k = 0; k < q; ++k) {
            gemv(blk, 'N', z, -1.0, 1.0, dim, u, dim, agg + (size_t)k * dim);
            blk += (size_t)dim * u * u;
          }
        }
      }
      for (int k = 0; k < q * dim; ++k) grad[k] = agg[k] / (double)M;
    }
    free(grad_mu);
    free(gchol);
    free(agg);
    pts_state_free(&s);
  }
  free(U);
  free(mu);
  free(chol);
  free(y);
  return rc;
}

/* ------------------------------------------------------------------------------------------------------------------
 * analytic 1,0-EI (OnePotentialSampleExpectedImprovementEvaluator, gpp_math.cpp:2195-2259): the posterior at one point is a
 * scalar Gaussian, EI = (best - mu) Phi(c) + sigma phi(c), c = (best - mu) / sigma.  Phi via erfc, like the shimmed
 * boost::math::cdf the reference build under oracle/_ref uses.
 * ---------------------------------------------------------------------------------------------------------------- */
static double std_normal_pdf(double z) { return exp(-0.5 * z * z) / 2.5066282746310002; }
static double std_normal_cdf(double z) { return 0.5 * erfc(-z * 0.70710678118654752440); }

int orc_ei_analytic(const orc_gp* gp, const double* pt, double best_so_far, double* ei_out, double* grad) {
  const int dim = gp->dim;
  const double kMinVarEI = 2.2250738585072014e-308;                                   /* gpp_math.hpp:1316 */
  const double kMinVarGradEI = 150.0 * 2.220446049250313e-16 * 2.220446049250313e-16; /* gpp_math.hpp:1323 */
  if (ei_out) { /* state without gradients (:2271-2281) */
    pts_state s;
    double mu, var;
    pts_state_fill(gp, &s, pt, 1, NULL, 0, 0, 0, 0);
    mean_of_points(gp, &s, &mu);
    variance_of_points(gp, &s, NULL, 0, &var);
    const double sigma = sqrt(fmax(kMinVarEI, var));
    const double t = best_so_far - mu;
    const double ei = t * std_normal_cdf(t / sigma) + sigma * std_normal_pdf(t / sigma);
    *ei_out = fmax(0.0, ei);
    pts_state_free(&s);
  }
  if (grad) {
    pts_state s;
    double mu, var;
    double* grad_mu = dalloc(dim);
    double* gchol = dalloc(dim);
    pts_state_fill(gp, &s, pt, 1, NULL, 0, 1, 1, 0);
    mean_of_points(gp, &s, &mu);
    grad_mean_of_points(gp, &s, grad_mu);
    variance_of_points(gp, &s, NULL, 0, &var);
    var = fmax(kMinVarGradEI, var);
    double sigma = sqrt(var);
    grad_cholesky_per_point(gp, &s, 0, &sigma, gchol);
    const double mu_diff = best_so_far - mu;
    const double c = mu_diff / sigma;
    const double pdf_c = std_normal_pdf(c), cdf_c = std_normal_cdf(c);
    for (int i = 0; i < dim; ++i) {
      const double d_c = (-sigma * grad_mu[i] - gchol[i] * mu_diff) / var;
      const double d_a = -grad_mu[i] * cdf_c + mu_diff * pdf_c * d_c;
      const double d_b = gchol[i] * pdf_c + sigma * (-c) * pdf_c * d_c;
      grad[i] = d_a + d_b;
    }
    free(grad_mu);
    free(gchol);
    pts_state_free(&s);
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------------------------------
 * q-KG / d-KG by Monte Carlo (gpp_knowledge_gradient_optimization.cpp)
 * ---------------------------------------------------------------------------------------------------------------- */

typedef struct {
  const orc_gp* gp; /* the fantasy GP */
  int f;            /* num_fidelity */
  double pt[ORC_MAX_DIM];
  long n_val, n_grad;
} pm_state;

/* PosteriorMeanState::SetCurrentPoint .cpp:353-357 */
static void pm_set(pm_state* s, const double* x) {
  const int dim = s->gp->dim;
  for (int i = 0; i < dim - s->f; ++i) s->pt[i] = x[i];
  for (int i = dim - s->f; i < dim; ++i) s->pt[i] = 1.0;
}

/* PosteriorMeanEvaluator::ComputePosteriorMean .cpp:324-331 */
static double pm_value(pm_state* s) {
  double mu;
  orc_gp_additional_mean(s->gp, s->pt, 1, NULL, 0, &mu);
  s->n_val++;
  return -mu;
}

/* PosteriorMeanEvaluator::ComputeGradPosteriorMean .cpp:341-351 */
static void pm_grad(pm_state* s, double* grad) {
  double gmu[ORC_MAX_DIM];
  orc_gp_grad_additional_mean(s->gp, s->pt, 1, NULL, 0, gmu);
  for (int i = 0; i < s->gp->dim - s->f; ++i) grad[i] = -gmu[i];
  s->n_grad++;
}

/* TensorProductDomain::LimitUpdate gpp_domain.cpp:64-105 */
static void limit_update(const double* bounds, int size, double max_relative_change, const double* x, double* step) {
  const double kInvalidStepScaleFactor = 0.5; /* gpp_domain.hpp */
  for (int j = 0; j < size; ++j) {
    const double lo = bounds[2 * j], hi = bounds[2 * j + 1];
    double desired = step[j];
    double dist = fmin(x[j] - lo, hi - x[j]);
    if (fabs(desired) > max_relative_change * dist) desired = copysign(max_relative_change * dist, desired);
    const double next = x[j] + desired;
    if (next < lo || next > hi) {
      if (next < lo) {
        dist = lo - x[j];
        if (x[j] + desired * kInvalidStepScaleFactor < lo)
          desired = dist * kInvalidStepScaleFactor;
        else
          desired *= kInvalidStepScaleFactor;
      } else {
        dist = hi - x[j];
        if (x[j] + desired * kInvalidStepScaleFactor > hi)
          desired = dist * kInvalidStepScaleFactor;
        else
          desired *= kInvalidStepScaleFactor;
      }
    }
    step[j] = desired;
  }
}

/* GradientDescentOptimizationLineSearch gpp_optimization.hpp:708-828 */
static void gd_line_search(pm_state* s, const double* gd, const double* bounds) {
  const int size = s->gp->dim - s->f;
  const int max_num_steps = (int)gd[1];
  const double gamma = gd[4], pre_mult = gd[5], max_relative_change = gd[6], tol = gd[7];
  double grad[ORC_MAX_DIM], step[ORC_MAX_DIM], trial[ORC_MAX_DIM], next[ORC_MAX_DIM];
  for (int j = 0; j < size; ++j) next[j] = s->pt[j];
  const double decrease_rate = 0.5, tolerance = 0.5;
  const double step_tolerance = tol / (double)max_num_steps;
  for (int i = 0; i < max_num_steps; ++i) {
    const double f0 = pm_value(s);
    double obj = 0.0;
    double alpha = pre_mult * pow((double)(i + 1), -gamma);
    pm_grad(s, grad);
    double norm = 0.0;
    for (int j = 0; j < size; ++j) norm += grad[j] * grad[j];
    const int max_search = 30;
    int search = 0;
    while (search < max_search) {
      for (int j = 0; j < size; ++j) step[j] = alpha * grad[j];
      for (int j = 0; j < size; ++j) trial[j] = next[j] + step[j];
      pm_set(s, trial);
      obj = pm_value(s);
      if (obj - f0 > tolerance * alpha * norm) break;
      alpha *= decrease_rate;
      search += 1;
    }
    for (int j = 0; j < size; ++j) step[j] = alpha * grad[j];
    limit_update(bounds, size, max_relative_change, next, step);
    for (int j = 0; j < size; ++j) trial[j] = next[j] + step[j];
    pm_set(s, trial);
    obj = pm_value(s);
    if (obj <= f0 || search == max_search) {
      pm_set(s, next);
      break;
    }
    for (int j = 0; j < size; ++j) next[j] += step[j];
    pm_set(s, next);
    if (vector_norm(step, size) < step_tolerance) break;
  }
}

/* GradientDescentOptimizerLineSearch::Optimize gpp_optimization.hpp:1242-1283 */
static void gd_optimize(pm_state* s, const double* gd, const double* bounds) {
  const int size = s->gp->dim - s->f;
  const int max_num_restarts = (int)gd[2];
  if (max_num_restarts <= 0) return;
  double cur[ORC_MAX_DIM];
  for (int i = 0; i < max_num_restarts; ++i) {
    for (int j = 0; j < size; ++j) cur[j] = s->pt[j];
    gd_line_search(s, gd, bounds);
    for (int j = 0; j < size; ++j) cur[j] -= s->pt[j];
    if (vector_norm(cur, size) <= gd[7]) break;
  }
}

/* ComputeOptimalPosteriorMean .cpp:420-472 (k = min(1, num_starts): best single start) */
static void optimal_posterior_mean(const orc_gp* gp_after, int f, const double* gd, const double* bounds,
                                   const double* starts, int num_starts, double* best_point, double* best_value,
                                   long* counters) {
  if ((int)gd[2] <= 0) return;
  const int size = gp_after->dim - f;
  pm_state s;
  s.gp = gp_after;
  s.f = f;
  s.n_val = s.n_grad = 0;
  int best_i = -1;
  double top = 0.0; /* priority_queue of (-val): keeps the entry with the smallest -val */
  for (int i = 0; i < num_starts; ++i) {
    pm_set(&s, starts + (size_t)i * size);
    const double val = pm_value(&s);
    if (i < 1) {
      top = -val;
      best_i = i;
    } else if (top > -val) {
      top = -val;
      best_i = i;
    }
  }
  *best_value = -INFINITY;
  if (best_i >= 0) {
    pm_set(&s, starts + (size_t)best_i * size);
    gd_optimize(&s, gd, bounds);
    const double v = pm_value(&s);
    if (v > *best_value) {
      *best_value = v;
      for (int j = 0; j < size; ++j) best_point[j] = s.pt[j];
    }
  }
  if (counters) {
    counters[0] += s.n_val;
    counters[1] += s.n_grad;
  }
}

/* ComputeGradCovarianceOfPointsPerPoint gpp_math.cpp:1063-1115 with num_gradients_discrete_pts = 0 and the
 * "precomputed=false" branch taken by the KG tail (grad_K_inv_times_K_star x kt). gcov[d + row*dim + col*dim*m] */
static void grad_cov_per_point(const orc_gp* gp, const pts_state* s, int diff, const double* disc, int num_pts,
                               const double* kt, double* gcov) {
  const int dim = gp->dim, N = gp->N, gt = s->gt, m = s->m;
  memset(gcov, 0, sizeof(double) * (size_t)dim * m * num_pts);
  double* temp = dalloc((size_t)dim * num_pts * (gt + 1));
  for (int i = 0; i < gt + 1; ++i) {
    const int index = diff * (gt + 1) + i;
    gemm(s->grad_K_inv_K_star + (size_t)index * dim * N, 'N', kt, 1.0, 0.0, dim, N, num_pts, temp + (size_t)i * dim * num_pts);
  }
  double* tmp = dalloc((size_t)dim * (gt + 1));
  for (int j = 0; j < num_pts; ++j) {
    orc_grad_covariance(&gp->cov, s->pts + (size_t)diff * dim, s->gradients, gt, disc + (size_t)j * dim, NULL, 0, tmp);
    for (int a = 0; a < gt + 1; ++a) {
      const size_t row = (size_t)a + (size_t)diff * (gt + 1);
      for (int d = 0; d < dim; ++d)
        gcov[d + row * dim + (size_t)j * dim * m] = tmp[d + dim * a] - temp[d + (size_t)dim * j + (size_t)dim * num_pts * a];
    }
  }
  free(tmp);
  free(temp);
}

int orc_kg(const orc_gp* gp, int f, const double* gd, const double* bounds, const double* discrete, int P,
           const double* Xq, const double* Xp, int q, int p, int M, double best_so_far, const double* normals,
           int want_grad, double* kg_out, double* grad_out, double* best_point_out, long* counters) {
  return orc_kg_head(gp, f, gd, bounds, discrete, P, Xq, Xp, q, p, M, best_so_far, normals, want_grad, kg_out, grad_out,
                     best_point_out, counters, NULL);
}

/* The same evaluation on a state whose discretised set was frozen when the state was BUILT: KnowledgeGradientState's
 * constructor fills discretized_set = [union points without fidelity dims ; discrete points] (.cpp:259-261), and
 * SetCurrentPoint (.cpp:232-243) refreshes union_of_points and the GP state but NOT discretized_set.  The multistart /
 * point-list drivers build their states at the FIRST start (gpp_knowledge_gradient_optimization.hpp:886-889, 1117-1120) and
 * then move them with SetCurrentPoint, so every evaluation they make scores and starts its inner optimisation from
 * head_q = that first start's q points (+ points_being_sampled, which never change).  head_q[q][dim] or NULL (= Xq: a
 * fresh state per evaluation, what the single-evaluation Python entry points do). */
int orc_kg_head(const orc_gp* gp, int f, const double* gd, const double* bounds, const double* discrete, int P,
                const double* Xq, const double* Xp, int q, int p, int M, double best_so_far, const double* normals,
                int want_grad, double* kg_out, double* grad_out, double* best_point_out, long* counters,
                const double* head_q) {
  const int dim = gp->dim, g = gp->g, u = q + p, m = u * (1 + g), N = gp->N;
  const int nd = want_grad ? q : 0;
  int rc = 0;
  if (counters) counters[0] = counters[1] = 0;
  /* KnowledgeGradientState ctor .cpp:246-275 */
  double* U = dalloc((size_t)u * dim);
  memcpy(U, Xq, sizeof(double) * (size_t)q * dim);
  if (p > 0) memcpy(U + (size_t)q * dim, Xp, sizeof(double) * (size_t)p * dim);
  const int A = u + P, size = dim - f;
  double* disc_set = dalloc((size_t)A * size);
  for (int i = 0; i < u; ++i)
    memcpy(disc_set + (size_t)i * size, (head_q && i < q ? head_q : U) + (size_t)i * dim, sizeof(double) * size);
  memcpy(disc_set + (size_t)u * size, discrete, sizeof(double) * (size_t)P * size);
  pts_state s;
  pts_state_fill(gp, &s, U, u, gp->derivs, g, nd, 1, want_grad ? 1 : 0);
  /* PreCompute .cpp:292-317 */
  double* mu = dalloc(m);
  double* chol = dalloc((size_t)m * m);
  orc_gp_additional_mean(gp, U, u, gp->derivs, g, mu);
  variance_of_points(gp, &s, gp->derivs, g, chol);
  for (int i = 0; i < u; ++i)
    for (int j = 0; j < 1 + g; ++j) {
      const size_t row = (size_t)i * (1 + g) + j;
      chol[row + row * m] += gp->noise[j];
    }
  rc = orc_cholesky(m, chol);
  double* normals_full = dalloc((size_t)M * m);
  double* best_point = dalloc((size_t)M * dim);
  double* make_up = dalloc(m);
  double* grad_mu = NULL;
  double* gchol = NULL;
  double* agg = NULL;
  orc_gp* after = NULL;
  if (rc != 0) goto done;
  for (int c = 0; c < m; ++c) /* ZeroUpperTriangle gpp_linear_algebra.cpp:83-90 */
    for (int r = 0; r < c; ++r) chol[r + (size_t)c * m] = 0.0;

  int winner = -1;
  double best_posterior = best_so_far;
  for (int j = 0; j < u; ++j) {
    if (mu[j * (1 + g)] < best_posterior) {
      winner = j;
      best_posterior = mu[j * (1 + g)];
    }
  }
  if (want_grad) {
    /* .cpp:134-161 */
    double* gm = dalloc((size_t)dim * q * (1 + g));
    grad_mu = dalloc((size_t)dim * q);
    grad_mean_of_points(gp, &s, gm);
    for (int i = 0; i < q; ++i)
      for (int d = 0; d < dim; ++d) grad_mu[d + i * dim] = gm[d + (size_t)i * (1 + g) * dim];
    free(gm);
    gchol = dalloc((size_t)dim * m * m * q);
    for (int k = 0; k < q; ++k) grad_cholesky_per_point(gp, &s, k, chol, gchol + (size_t)k * dim * m * m);
    agg = dalloc((size_t)dim * q);
    if (winner >= 0 && winner < q)
      for (int d = 0; d < dim; ++d) agg[winner * dim + d] += M * grad_mu[winner * dim + d];
    for (size_t i = 0; i < (size_t)M * dim; ++i) best_point[i] = 1.0;
  }

  /* fantasy GP: copy + AddSampledPointsToGP(union, zeros) gpp_math.cpp:1720-1737 */
  after = gp_alloc(&gp->cov, g, gp->derivs, dim, gp->n + u);
  memcpy(after->X, gp->X, sizeof(double) * (size_t)gp->n * dim);
  memcpy(after->X + (size_t)gp->n * dim, U, sizeof(double) * (size_t)u * dim);
  memcpy(after->y, gp->y, sizeof(double) * (size_t)N);
  memcpy(after->noise, gp->noise, sizeof(double) * (size_t)(1 + g));
  after->mean = gp->mean;
  rc = recompute_cholesky_variables(after);
  if (rc != 0) goto done;

  double aggregate = 0.0;
  for (int i = 0; i < M; ++i) {
    double* z = normals_full + (size_t)i * m;
    if (i % 2 == 1) {
      for (int j = 0; j < m; ++j) z[j] = -normals_full[(size_t)(i - 1) * m + j];
    } else {
      for (int j = 0; j < m; ++j) z[j] = normals[(size_t)(i / 2) * m + j];
    }
    memcpy(make_up, mu, sizeof(double) * m);
    gemv(chol, 'N', z, 1.0, 1.0, m, m, m, make_up);
    /* NewSampledValue(values, num_union, num_sampled, mean_change=false) gpp_math.cpp:1739-1747 */
    memcpy(after->y + (size_t)gp->n * (1 + g), make_up, sizeof(double) * m);
    recompute_mean_variables(after, 0);
    double best_value = 0.0;
    optimal_posterior_mean(after, f, gd, bounds, disc_set, A, best_point + (size_t)i * dim, &best_value, counters);
    aggregate += best_posterior + best_value;
  }
  if (kg_out) *kg_out = aggregate / (double)M;

  if (want_grad) {
    /* .cpp:199-225 */
    double* cic = dalloc((size_t)m * M);
    build_mix(&gp->cov, U, best_point, u, M, s.gradients, g, NULL, 0, cic);
    double* kt = dalloc((size_t)N * M);
    gp_mix(gp, best_point, M, NULL, 0, kt);
    gemm(s.K_inv_K_star, 'T', kt, -1.0, 1.0, m, N, M, cic);
    tri_solve_mat(chol, 'N', m, M, cic);
    /* ComputeGradInverseCholeskyCovarianceOfPointsPerPoint gpp_math.cpp:1601-1651 */
    double* gcov = dalloc((size_t)dim * m * M);
    double* temp = dalloc((size_t)m * M);
    double* temp_chol = dalloc((size_t)m * m);
    double* gic = dalloc((size_t)dim * m * M);
    for (int k = 0; k < q; ++k) {
      grad_cov_per_point(gp, &s, k, best_point, M, kt, gcov);
      const double* gck = gchol + (size_t)k * dim * m * m;
      for (int d = 0; d < dim; ++d) {
        for (int j = 0; j < m; ++j)
          for (int l = 0; l < M; ++l) temp[j + (size_t)l * m] = gcov[d + (size_t)j * dim + (size_t)l * dim * m];
        tri_solve_mat(chol, 'N', m, M, temp);
        memset(temp_chol, 0, sizeof(double) * (size_t)m * m);
        for (int j = 0; j < m; ++j)
          for (int l = j; l < m; ++l) temp_chol[l + (size_t)j * m] = gck[d + (size_t)j * dim + (size_t)l * dim * m];
        tri_solve_mat(chol, 'N', m, m, temp_chol);
        gemm(temp_chol, 'N', cic, -1.0, 1.0, m, m, M, temp);
        for (int j = 0; j < m; ++j)
          for (int l = 0; l < M; ++l) gic[d + (size_t)j * dim + (size_t)l * dim * m] = temp[j + (size_t)l * m];
      }
      for (int i = 0; i < M; ++i)
        gemv(gic + (size_t)i * dim * m, 'N', normals_full + (size_t)i * m, -1.0, 1.0, dim, m, dim, agg + (size_t)k * dim);
    }
    for (int k = 0; k < q * dim; ++k) grad_out[k] = agg[k] / (double)M;
    free(cic);
    free(kt);
    free(gcov);
    free(temp);
    free(temp_chol);
    free(gic);
  }
  if (best_point_out) memcpy(best_point_out, best_point, sizeof(double) * (size_t)M * dim);

done:
  orc_gp_destroy(after);
  free(U);
  free(disc_set);
  pts_state_free(&s);
  free(mu);
  free(chol);
  free(normals_full);
  free(best_point);
  free(make_up);
  free(grad_mu);
  free(gchol);
  free(agg);
  return rc;
}
