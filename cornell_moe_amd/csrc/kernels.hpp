// cornell_moe_amd/csrc/kernels.hpp -- launch wrappers of the hand-written gfx950 kernels (definitions in kernels_*.hip).
//
// Device data layout (everything FP64):
//   points      : row-major [point][DP], DP = dim rounded up to a multiple of 4, zero padded
//                 (a zero pad contributes exactly +0.0 to every squared distance);
//   matrices    : column-major, leading dimension given explicitly (rows are what consecutive lanes walk, so every
//                 global store/load of a matrix column is a coalesced 512 B per wavefront);
//   CovParams   : kernel type, alpha, 1/l^2 per (padded) dimension, passed by value.
#pragma once
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace moe {

constexpr int kMaxDimPadded = 32;  // dims up to 32 (C5 has 12); kernels are instantiated for padded_dim(d) in {4, 8, 12, 16, 24, 32}
constexpr int kMaxDerivs = 16;

struct CovParams {
  int type;  // MOE_COV_*
  int dim;   // true dim
  int dp;    // padded dim (multiple of 4)
  double alpha;
  double inv_l2[kMaxDimPadded];  // 1 / length^2, 0 in padded slots
  double inv_l[kMaxDimPadded];   // 1 / length, 0 in padded slots
  double center[kMaxDimPadded];  // a reference point near the data (the training-set mean; any point works): the value-only
                                 // covariance build subtracts it before scaling so that pre-scaled coordinates keep the
                                 // precision of the differences wherever the domain sits (kernels_cov.hip)
};

struct DerivList {
  int g;
  int idx[kMaxDerivs];
};

// out[(i*(1+gA)+a) + (col0 + j*(1+gB)+b) * ld] = cov(A_i, B_j)[a, b]   (BuildMixCovarianceMatrix, gpp_math.cpp:309-335)
// If diag_noise != nullptr (A == B, gA == gB): adds diag_noise[a] where row == col (gpp_math.cpp:426-455).
// streaming = true: the big N x M cross-covariance builds (gradient tail, the roofline probe) may take the value-only fast path
// (centred pre-scaled coordinates + table exp, <= 2 ulp per entry); the GP's own K(X, X), K* and the posterior queries keep the
// general kernel, whose entries follow the reference's formulas operation for operation (a duplicate point must still be
// reported singular at the same pivot).
// lower_only = true (A == B: the symmetric K(X, X) a factorisation consumes): only entries with row >= column are computed and
// written -- 8 [n d + N (N + 1) / 2] bytes, SURVEY 8(d)'s symmetric count -- and the strict upper triangle of `out` is left untouched.
void launch_cov_build(const CovParams& cp, const double* A, int nA, const DerivList& dA, const double* B, int nB,
                      const DerivList& dB, const double* diag_noise, double* out, long ld, long col0, hipStream_t s,
                      bool streaming = false, bool lower_only = false);
// K(A, [B1 | B2]) without derivative observations, two column ranges, one launch (r5; kernels_cov.hip)
void launch_cov_build_pair(const CovParams& cp, const double* A, int nA, const double* B, int nB1, long col1, int nB2, long col2,
                           double* out, long ld, hipStream_t s);

// E[(j*(1+g)+n) + (col0 + (i*(1+gt)+m)*dim + dd) * ld] = d cov(P_i, X_j)[m, n] / d P_{i,dd}
// (grad_K_star fill, gpp_math.cpp:616-637)
// Posterior mean of the function value at nP points P[nP][DP] (+ gradient): out[nP] or out[nP][1 + DP] = (mu, d mu / d x).
void launch_mean(const CovParams& cp, const double* X, int n, const DerivList& dX, const double* KinvY, const double* P,
                 int nP, double mean, bool want_grad, double* out, hipStream_t s);
void launch_grad_kstar(const CovParams& cp, const double* X, int n, const DerivList& dX, const double* P, int nP,
                       const DerivList& dP, double* out, long ld, long col0, hipStream_t s);

// e[i] = exp_nonpos(-x[i]), r[i] = sqrt_nonneg(x[i])  (accuracy probe of fastmath.hpp)
void launch_debug_math(const double* x, int n, double* e, double* r, hipStream_t s);

// C[N x c] (ldc) = op(T) * B[N x c] (ldb); T lower-triangular N x N (ldt), op = 'N' or 'T'.
void launch_tri_gemm(char op, int N, int c, const double* T, long ldt, const double* B, long ldb, double* C, long ldc,
                     hipStream_t s);
// G (c x c, ldg) = S^T S for S = the columns 0, stride, 2 stride, ... of the lower-triangular T (N x N, explicit zeros above the
// diagonal): with T = L^-1 the rows / columns i stride of K^-1 = L^-T L^-1 -- all the hyper-parameter gradient of the log marginal
// likelihood reads off the diagonal (the function-value rows: 1 / (1 + g)^2 of the matrix).  diag (or NULL): the N squared column
// norms of T = diag(K^-1).
void launch_tri_gram_strided(int N, int c, int stride, const double* T, long ldt, double* G, long ldg, double* diag, hipStream_t s);
// The same product through kernels shaped for c <= 16 columns (falls back to launch_tri_gemm above that).  A different
// summation order: the per-evaluation state set-up keeps to launch_tri_gemm so that its results do not depend on how many
// evaluations share a call; the GP-level solves (K^-1 (y - mean), the block row of an append) use this one.
void launch_tri_gemm_skinny(char op, int N, int c, const double* T, long ldt, const double* B, long ldb, double* C,
                            long ldc, hipStream_t s);

// op(T) B for a right-hand side that is a row of independent problems of `cols_per_problem` columns each (the evaluations of a KG
// batch): the kernel is chosen from N and cols_per_problem alone and every column's summation order is fixed by N, so a problem's
// result does not depend on how many share the call.  <= 16 columns per problem: the skinny kernels; otherwise split K on the matrix
// pipe (kernels_linalg.hip tri_splitk_kernel).  work: tri_cols_work_doubles(N, c) doubles.
size_t tri_cols_work_doubles(int N, int c);
void launch_tri_gemm_cols(char op, int N, int c, int cols_per_problem, const double* T, long ldt, const double* B, long ldb,
                          double* C, long ldc, double* work, hipStream_t s);

// C[m x n] (ldc) = A^T B, A is K x m (lda), B is K x n (ldb); reduction over the K rows.
void launch_gemm_tn(int m, int n, int K, const double* A, long lda, const double* B, long ldb, double* C, long ldc,
                    hipStream_t s);
// The same for `batch` problems of one shape in ONE launch (problem e: A + e sA, B + e sB; its result dense, m x n with leading
// dimension m, at C + e m n), with K cut into `slices` partial products summed in slice order: skinny outputs over a long K.
// work holds slices * batch * m * n doubles (unused for slices == 1).
void launch_gemm_tn_splitk(int m, int n, int K, const double* A, long lda, const double* B, long ldb, double* C, double* work,
                           int slices, hipStream_t s, int batch = 1, long sA = 0, long sB = 0);

// C[i] = sum over the `slices` partial results work[sl * count + i], in slice order.
void launch_sum_slices(const double* work, int slices, long count, double* C, hipStream_t s);

// Batched Gram matrices over column groups of V (K x *, ldv): for eval e, G_e[c x c] (ld c, eval stride c*c) = V_e^T V_e
// where V_e's column `l` is V's column  l < m ? e*m + l : l < m+ng ? E*m + e*ng + (l-m) : E*(m+ng) + e*A + (l-m-ng).
// work (may be NULL: single pass): E * gram_batch_slices(E, c, K) * c * c doubles of scratch for the K-sliced version.
int gram_batch_slices(int E, int c, int K);
// defer_sum (r5): with more than one K slice leave the partial Grams in `work` ([E][slices][c c]) for a consumer that adds them up in
// slice order itself (kg_state.hip); G is then not written.  Returns the slice count (1: G holds the result).
int launch_gram_batch(int E, int m, int ng, int A, int K, const double* V, long ldv, double* G, double* work,
                      hipStream_t s, bool defer_sum = false);

// Batched cross products of the KG state (r4): for eval e, G_e[(ng + A) x m] (ld ng + A, eval stride (ng + A) m) = S_e^T W_e, where
// S_e = the evaluation's gradient and extra columns of S (column map of launch_gram_batch, l >= m) and W_e = columns e m .. of W.
// work: E * gram_cross_slices(m, ng, A, K) * (ng + A) * m doubles.
int gram_cross_slices(int m, int ng, int A, int K);
int launch_gram_cross_batch(int E, int m, int ng, int A, int K, const double* S, long lds, const double* W, long ldw, double* G,
                            double* work, hipStream_t s, bool defer_sum = false);  // (defer_sum, return value: as launch_gram_batch)

// In-place blocked Cholesky of the lower triangle of A (N x N, lda) + explicit inverse of the factor into Linv
// (N x N, ldl, lower; strict upper zeroed).  info (device int): 0 or failing pivot index + 1 (pivot <= 1e-16).
// work: cholesky_work_doubles(N) doubles of device scratch for the recursive inversion.
// upper_is_zero: the caller guarantees that A's strict upper triangle already holds zeros (a lower_only covariance build into a
// cleared buffer); otherwise it is zeroed here after the factorisation.
// side (r5): a second stream for the early-inverse schedule (kernels_linalg.hip) -- the leading half's inverse and the top level's first
// product run on it while `s` factors the trailing half; NULL: everything on `s`, one level after the other.  The caller need not
// synchronise `side`: `s` waits for it before the call's last launches.
void launch_cholesky_and_inverse(int N, double* A, long lda, double* Linv, long ldl, double* work, int* info,
                                 hipStream_t s, bool upper_is_zero = false, hipStream_t side = nullptr);
size_t cholesky_work_doubles(int N);
long early_inverse_split(int N);  // the early-inverse schedule's split row (0: the schedule does not apply at this N)
// scratch of the two-level factorisation's step kernel (two N x 64 column-block buffers), in doubles
size_t chol_scratch_doubles(int N);

// Rank-kk append to a factorisation, in place: L / Linv hold the factor and inverse factor of K11 in their leading
// N0 x N0 blocks and have room for kk more rows and columns (ldl, ldi >= N0 + kk).  Given B = K12 (N0 x kk, ld N0) and
// C = K22 + noise (kk x kk, ld kk; overwritten by L22) the new block row is written:
//   L21 = (L11^-1 K12)^T,  L22 = chol(K22 - L21 L21^T),  X21 = -L22^-1 L21 L11^-1,  X22 = L22^-1
// -- O(N0^2 kk) instead of the O(N^3) refactorisation (the reference's TODO GH-192, gpp_math.cpp:1699-1737).  The old
// block is not touched, so a failed append (info != 0: pivot index within the new block + 1) leaves it valid.
// work: cholesky_append_work_doubles doubles.
size_t cholesky_append_work_doubles(int N0, int kk);
void launch_cholesky_append(int N0, int kk, double* L, long ldl, double* Linv, long ldi, const double* B, double* C,
                            double* work, int* info, hipStream_t s);

// ---- a batch of independent factorisations (the log-likelihood of many hyper-parameter sets on the same data) ----
// In-place blocked Cholesky of `batch` matrices A + b * a_stride (N x N lower triangles, lda); the inverses of the diagonal
// blocks go to Linv + b * l_stride (N x N layout, ldl; only the 64 x 64 diagonal blocks are written -- no full inverse
// factor).  info[b]: 0 or failing pivot + 1.  Every step is ONE launch over the batch (grid.z).
void launch_cholesky_batch(int N, double* A, long lda, long a_stride, double* Linv, long ldl, long l_stride, int* info,
                           int batch, hipStream_t s, double* scratch = nullptr);
// The log-likelihood's quadratic form comes out of the same factorisation: with the centred data as an extra ROW N of the
// matrix (corner 1e100) the factor's row N is v^T = (L^-1 yc)^T, and yc^T K^-1 yc = |v|^2 -- no triangular solve.
// launch_ll_border writes that row; launch_ll_terms_batch returns out[b] = (sum log L_ii, |v|^2) over i, j < N.
void launch_ll_border(double* A, long lda, long a_stride, int N, const double* yc, int batch, hipStream_t s);
void launch_ll_terms_batch(const double* A, long lda, long a_stride, int N, double* out, int batch, hipStream_t s);

}  // namespace moe
