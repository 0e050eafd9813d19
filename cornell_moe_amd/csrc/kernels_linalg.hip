// cornell_moe_amd/csrc/kernels_linalg.hip -- FP64 dense linear algebra for the GP posterior on gfx950.
//
//  * tile_gemm_kernel : LDS-tiled FP64 GEMM (vector FMA; on gfx950 the FP64 MFMA peak equals the FP64 vector peak, so the
//    matrix pipe buys nothing here) with the three operand shapes the path needs:
//      tri 'N'  C = L  B   (L lower triangular)          -> V = L^-1-apply via the explicit inverse factor
//      tri 'T'  C = L^T B
//      tn       C = A^T B  (Gram matrices / posterior-mean dot products / gradient-tail contraction)
//  * blocked right-looking Cholesky (NB = 64) + explicit inverse of the factor: replaces ComputeCholeskyFactorL
//    (gpp_linear_algebra.cpp:109-148) and turns every TriangularMatrixVectorSolve (:160-187) of the reference into a
//    GEMM against L^-1, which is what lets the posterior solves run wide instead of as 1000 dependent steps.
#include "kernels.hpp"

namespace moe {

namespace {

constexpr int TK = 16;

// MODE 0: C = A^T B, A is K x m (element (k,i) at k + i*lda): full k range.
// MODE 1: C = T B,   T lower (element (i,k) at i + k*lda), k < i0 + TM.
// MODE 2: C = T^T B, T lower (element (k,i) at k + i*lda), k >= i0.
// KT = depth of one LDS stage: skinny outputs (few workgroups, latency-bound K loop) use deep stages so the loop has 4x
// fewer global-load / barrier round trips; big outputs keep 16 for occupancy.
template <int TM, int TN, int MODE, int KT>
__global__ __launch_bounds__(256) void tile_gemm_kernel(int M, int Ncols, int K, const double* __restrict__ A, long lda,
                                                       const double* __restrict__ B, long ldb, double* __restrict__ C,
                                                       long ldc) {
  constexpr int RM = TM / 16, RN = TN / 16;
  constexpr int TK = KT;
  __shared__ double As[TK][TM + 1];
  __shared__ double Bs[TK][TN + 1];
  const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;
  const int i0 = blockIdx.x * TM, j0 = blockIdx.y * TN;
  int k_lo = 0, k_hi = K;
  if (MODE == 1) k_hi = min(K, i0 + TM);
  if (MODE == 2) k_lo = (i0 / TK) * TK;
  double acc[RM][RN];
#pragma unroll
  for (int a = 0; a < RM; ++a)
#pragma unroll
    for (int b = 0; b < RN; ++b) acc[a][b] = 0.0;

  for (int k0 = k_lo; k0 < k_hi; k0 += TK) {
    // stage A tile: As[kk][ii] = Aop(i0+ii, k0+kk)
    if (MODE == 1) {
      for (int t = threadIdx.x; t < TK * TM; t += 256) {
        const int ii = t % TM, kk = t / TM;
        const int gi = i0 + ii, gk = k0 + kk;
        As[kk][ii] = (gi < M && gk < K && gk <= gi) ? A[(long)gi + (long)gk * lda] : 0.0;
      }
    } else {
      for (int t = threadIdx.x; t < TK * TM; t += 256) {
        const int kk = t % TK, ii = t / TK;
        const int gi = i0 + ii, gk = k0 + kk;
        bool ok = gi < M && gk < K;
        if (MODE == 2) ok = ok && gk >= gi;
        As[kk][ii] = ok ? A[(long)gk + (long)gi * lda] : 0.0;
      }
    }
    for (int t = threadIdx.x; t < TK * TN; t += 256) {
      const int kk = t % TK, jj = t / TK;
      const int gk = k0 + kk, gj = j0 + jj;
      Bs[kk][jj] = (gk < K && gj < Ncols) ? B[(long)gk + (long)gj * ldb] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < TK; ++kk) {
      double av[RM], bv[RN];
#pragma unroll
      for (int a = 0; a < RM; ++a) av[a] = As[kk][tx + 16 * a];
#pragma unroll
      for (int b = 0; b < RN; ++b) bv[b] = Bs[kk][ty + 16 * b];
#pragma unroll
      for (int a = 0; a < RM; ++a)
#pragma unroll
        for (int b = 0; b < RN; ++b) acc[a][b] = fma(av[a], bv[b], acc[a][b]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < RM; ++a)
#pragma unroll
    for (int b = 0; b < RN; ++b) {
      const int gi = i0 + tx + 16 * a, gj = j0 + ty + 16 * b;
      if (gi < M && gj < Ncols) C[(long)gi + (long)gj * ldc] = acc[a][b];
    }
}

template <int MODE>
void tile_gemm(int M, int Ncols, int K, const double* A, long lda, const double* B, long ldb, double* C, long ldc,
               hipStream_t s) {
  if (M <= 0 || Ncols <= 0) return;
  // Skinny outputs: smaller tiles give the chip more workgroups to place.
  const long blocks64 = (long)((M + 63) / 64) * ((Ncols + 63) / 64);
  if (blocks64 >= 512) {
    dim3 grid((M + 63) / 64, (Ncols + 63) / 64);
    hipLaunchKernelGGL((tile_gemm_kernel<64, 64, MODE, 16>), grid, dim3(256), 0, s, M, Ncols, K, A, lda, B, ldb, C, ldc);
  } else {
    dim3 grid((M + 31) / 32, (Ncols + 15) / 16);
    hipLaunchKernelGGL((tile_gemm_kernel<32, 16, MODE, 64>), grid, dim3(256), 0, s, M, Ncols, K, A, lda, B, ldb, C, ldc);
  }
  MOE_HIP_CHECK(hipGetLastError());
}

// Batched Gram matrices G_e = V_e^T V_e over column groups of V (see kernels.hpp): 32 x 32 output tile per workgroup,
// blockIdx.z = evaluation, only tiles with tile-row >= tile-col are computed and mirrored.
struct GramMap {
  int E, m, ng, A;
  __device__ __forceinline__ long col(int e, int l) const {
    if (l < m) return (long)e * m + l;
    if (l < m + ng) return (long)E * m + (long)e * ng + (l - m);
    return (long)E * (m + ng) + (long)e * A + (l - m - ng);
  }
};

__global__ __launch_bounds__(256) void gram_batch_kernel(GramMap gm, int K, const double* __restrict__ V, long ldv,
                                                        double* __restrict__ G) {
  constexpr int T = 32;
  constexpr int TK = 64;  // deep stages: the grid is tiny (c <= a few hundred), the K loop is latency-bound
  __shared__ double As[TK][T + 1];
  __shared__ double Bs[TK][T + 1];
  if (blockIdx.y > blockIdx.x) return;
  const int c = gm.m + gm.ng + gm.A;
  const int e = blockIdx.z;
  const int i0 = blockIdx.x * T, j0 = blockIdx.y * T;
  const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;
  double acc[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
  for (int k0 = 0; k0 < K; k0 += TK) {
    for (int t = threadIdx.x; t < TK * T; t += 256) {
      const int kk = t % TK, ii = t / TK;
      const int gk = k0 + kk;
      const int gi = i0 + ii, gj = j0 + ii;
      As[kk][ii] = (gi < c && gk < K) ? V[(long)gk + gm.col(e, gi) * ldv] : 0.0;
      Bs[kk][ii] = (gj < c && gk < K) ? V[(long)gk + gm.col(e, gj) * ldv] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < TK; ++kk) {
      const double a0 = As[kk][tx], a1 = As[kk][tx + 16], b0 = Bs[kk][ty], b1 = Bs[kk][ty + 16];
      acc[0][0] = fma(a0, b0, acc[0][0]);
      acc[0][1] = fma(a0, b1, acc[0][1]);
      acc[1][0] = fma(a1, b0, acc[1][0]);
      acc[1][1] = fma(a1, b1, acc[1][1]);
    }
    __syncthreads();
  }
  double* Ge = G + (long)e * c * c;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int gi = i0 + tx + 16 * a, gj = j0 + ty + 16 * b;
      if (gi < c && gj < c) {
        Ge[(long)gi + (long)gj * c] = acc[a][b];
        Ge[(long)gj + (long)gi * c] = acc[a][b];  // symmetric: identical sum, so both triangles hold the same bits
      }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Blocked Cholesky (lower, in place) with explicit inverse factor.
// ------------------------------------------------------------------------------------------------------------------
constexpr int NB = 64;

// Factor the nb x nb diagonal block at (k0,k0); write L_kk back (strict upper zeroed) and its inverse into Linv's
// diagonal block.  One wavefront: lane t owns row t during the factorisation and column t of the inverse.
__global__ __launch_bounds__(64) void chol_diag_kernel(double* __restrict__ A, long lda, double* __restrict__ Linv,
                                                      long ldl, int k0, int nb, int* __restrict__ info) {
  __shared__ double S[NB][NB + 1];
  const int t = threadIdx.x;
  if (*info != 0) return;
  for (int c = 0; c < nb; ++c)
    if (t < nb) S[t][c] = (c <= t) ? A[(long)(k0 + t) + (long)(k0 + c) * lda] : 0.0;
  __syncthreads();
  for (int k = 0; k < nb; ++k) {
    const double piv = S[k][k];
    if (!(piv > 1.0e-16)) {  // gpp_linear_algebra.cpp:118
      if (t == 0) *info = k0 + k + 1;
      return;
    }
    const double lkk = sqrt(piv);
    __syncthreads();
    if (t == k) S[k][k] = lkk;
    if (t > k && t < nb) S[t][k] = S[t][k] / lkk;
    __syncthreads();
    if (t > k && t < nb) {
      const double lik = S[t][k];
      for (int j = k + 1; j <= t; ++j) S[t][j] = S[t][j] - lik * S[j][k];
    }
    __syncthreads();
  }
  for (int c = 0; c < nb; ++c)
    if (t < nb) A[(long)(k0 + t) + (long)(k0 + c) * lda] = S[t][c];  // strict upper written as 0
  // inverse: lane t solves L x = e_t (forward substitution, entries above t are 0)
  if (t < nb) {
    double x[NB];
#pragma unroll 1
    for (int i = 0; i < nb; ++i) {
      if (i < t) {
        x[i] = 0.0;
      } else {
        double sum = (i == t) ? 1.0 : 0.0;
        for (int j = t; j < i; ++j) sum -= S[i][j] * x[j];
        x[i] = sum / S[i][i];
      }
      Linv[(long)(k0 + i) + (long)(k0 + t) * ldl] = x[i];
    }
  }
}

// Panel below the diagonal block: L_ik = A_ik * L_kk^-T, i.e. out[i][c] = sum_j A[i][j] * Linv_kk[c][j].
// One workgroup per 64 panel rows; both operands staged in LDS, 4 x 4 outputs per thread.
__global__ __launch_bounds__(256) void chol_panel_kernel(double* __restrict__ A, long lda, const double* __restrict__ Linv,
                                                        long ldl, int N, int k0, int nb, const int* __restrict__ info) {
  __shared__ double D[NB][NB + 1];   // D[j][c] = Linv_kk[c][j]
  __shared__ double At[NB][NB + 1];  // At[j][i] = A[i0 + i][k0 + j]
  if (*info != 0) return;
  const int i0 = k0 + nb + blockIdx.x * NB;
  for (int t = threadIdx.x; t < NB * NB; t += 256) {
    const int r = t % NB, c = t / NB;  // r walks rows (contiguous in memory)
    D[c][r] = (r < nb && c < nb) ? Linv[(long)(k0 + r) + (long)(k0 + c) * ldl] : 0.0;  // D[c][r] = Linv[r][c]
    At[c][r] = (i0 + r < N && c < nb) ? A[(long)(i0 + r) + (long)(k0 + c) * lda] : 0.0;
  }
  __syncthreads();
  const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
#pragma unroll 4
  for (int j = 0; j < NB; ++j) {
    double av[4], bv[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) av[a] = At[j][tx + 16 * a];
#pragma unroll
    for (int b = 0; b < 4; ++b) bv[b] = D[j][ty + 16 * b];  // Linv_kk[c][j] with c = ty + 16 b
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = fma(av[a], bv[b], acc[a][b]);
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int gi = i0 + tx + 16 * a, c = ty + 16 * b;
      if (gi < N && c < nb) A[(long)gi + (long)(k0 + c) * lda] = acc[a][b];
    }
}

// Trailing update: A[i][j] -= sum_c L[i][c] L[j][c], c over the panel, for 64 x 64 tiles with tile-row >= tile-col.
__global__ __launch_bounds__(256) void chol_update_kernel(double* __restrict__ A, long lda, int N, int k0, int nb,
                                                         const int* __restrict__ info) {
  __shared__ double Ls_i[TK][NB + 1];
  __shared__ double Ls_j[TK][NB + 1];
  if (*info != 0) return;
  if (blockIdx.y > blockIdx.x) return;
  const int base = k0 + nb;
  const int i0 = base + blockIdx.x * NB, j0 = base + blockIdx.y * NB;
  const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
  for (int c0 = 0; c0 < nb; c0 += TK) {
    for (int t = threadIdx.x; t < TK * NB; t += 256) {
      const int ii = t % NB, cc = t / NB;
      const int gi = i0 + ii, gj = j0 + ii, gc = k0 + c0 + cc;
      const bool cok = (c0 + cc) < nb;
      Ls_i[cc][ii] = (cok && gi < N) ? A[(long)gi + (long)gc * lda] : 0.0;
      Ls_j[cc][ii] = (cok && gj < N) ? A[(long)gj + (long)gc * lda] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int cc = 0; cc < TK; ++cc) {
      double av[4], bv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) av[a] = Ls_i[cc][tx + 16 * a];
#pragma unroll
      for (int b = 0; b < 4; ++b) bv[b] = Ls_j[cc][ty + 16 * b];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = fma(av[a], bv[b], acc[a][b]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int gi = i0 + tx + 16 * a, gj = j0 + ty + 16 * b;
      if (gi < N && gj < N && gj <= gi) A[(long)gi + (long)gj * lda] -= acc[a][b];
    }
}

// Off-diagonal blocks of L^-1, block row bi: X_{bi,bk} = -Linv_{bi,bi} * sum_{bj=bk}^{bi-1} L_{bi,bj} X_{bj,bk}.
__global__ __launch_bounds__(256) void trtri_row_kernel(const double* __restrict__ L, long lda, double* __restrict__ Linv,
                                                       long ldl, int N, int bi, const int* __restrict__ info) {
  __shared__ double As[TK][NB + 1];
  __shared__ double Bs[TK][NB + 1];
  __shared__ double Ssum[NB][NB + 1];
  if (*info != 0) return;
  const int bk = blockIdx.x;  // < bi
  const int i0 = bi * NB, j0 = bk * NB;
  const int ni = min(NB, N - i0);
  const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
  for (int k0 = j0; k0 < i0; k0 += TK) {
    for (int t = threadIdx.x; t < TK * NB; t += 256) {
      const int ii = t % NB, kk = t / NB;
      As[kk][ii] = (ii < ni) ? L[(long)(i0 + ii) + (long)(k0 + kk) * lda] : 0.0;
    }
    for (int t = threadIdx.x; t < TK * NB; t += 256) {
      const int kk = t % TK, jj = t / TK;
      // X_{bj,bk} is lower triangular within the diagonal block bk (zero above its diagonal, which was zero-filled)
      Bs[kk][jj] = Linv[(long)(k0 + kk) + (long)(j0 + jj) * ldl];
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < TK; ++kk) {
      double av[4], bv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) av[a] = As[kk][tx + 16 * a];
#pragma unroll
      for (int b = 0; b < 4; ++b) bv[b] = Bs[kk][ty + 16 * b];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = fma(av[a], bv[b], acc[a][b]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) Ssum[tx + 16 * a][ty + 16 * b] = acc[a][b];
  __syncthreads();
  // X = -Dinv_bi * Ssum ; Dinv_bi lower triangular (read straight from Linv's diagonal block, L2-resident)
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int r = tx + 16 * a, c = ty + 16 * b;
      if (r < ni) {
        double sum = 0.0;
        for (int k = 0; k <= r; ++k) sum = fma(Linv[(long)(i0 + r) + (long)(i0 + k) * ldl], Ssum[k][c], sum);
        Linv[(long)(i0 + r) + (long)(j0 + c) * ldl] = -sum;
      }
    }
}

__global__ void zero_strict_upper_kernel(double* __restrict__ A, long lda, int N) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)N * N;
  if (idx >= total) return;
  const int i = (int)(idx % N), j = (int)(idx / N);
  if (j > i) A[(long)i + (long)j * lda] = 0.0;
}

}  // namespace

void launch_tri_gemm(char op, int N, int c, const double* T, long ldt, const double* B, long ldb, double* C, long ldc,
                     hipStream_t s) {
  if (op == 'N')
    tile_gemm<1>(N, c, N, T, ldt, B, ldb, C, ldc, s);
  else
    tile_gemm<2>(N, c, N, T, ldt, B, ldb, C, ldc, s);
}

void launch_gemm_tn(int m, int n, int K, const double* A, long lda, const double* B, long ldb, double* C, long ldc,
                    hipStream_t s) {
  tile_gemm<0>(m, n, K, A, lda, B, ldb, C, ldc, s);
}

void launch_gram_batch(int E, int m, int ng, int A, int K, const double* V, long ldv, double* G, hipStream_t s) {
  const int c = m + ng + A;
  if (c <= 0 || E <= 0) return;
  GramMap gm{E, m, ng, A};
  dim3 grid((c + 31) / 32, (c + 31) / 32, E);
  hipLaunchKernelGGL(gram_batch_kernel, grid, dim3(256), 0, s, gm, K, V, ldv, G);
  MOE_HIP_CHECK(hipGetLastError());
}

size_t cholesky_work_doubles(int) { return 1; }

void launch_cholesky_and_inverse(int N, double* A, long lda, double* Linv, long ldl, double* /*work*/, int* info,
                                 hipStream_t s) {
  MOE_HIP_CHECK(hipMemsetAsync(info, 0, sizeof(int), s));
  MOE_HIP_CHECK(hipMemsetAsync(Linv, 0, sizeof(double) * (size_t)ldl * N, s));
  const int nblk = (N + NB - 1) / NB;
  for (int b = 0; b < nblk; ++b) {
    const int k0 = b * NB, nb = std::min(NB, N - k0);
    hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(64), 0, s, A, lda, Linv, ldl, k0, nb, info);
    const int below = N - k0 - nb;
    if (below > 0) {
      hipLaunchKernelGGL(chol_panel_kernel, dim3((below + NB - 1) / NB), dim3(256), 0, s, A, lda, Linv, ldl, N, k0, nb, info);
      const int tb = (below + NB - 1) / NB;
      hipLaunchKernelGGL(chol_update_kernel, dim3(tb, tb), dim3(256), 0, s, A, lda, N, k0, nb, info);
    }
  }
  {
    const long total = (long)N * N;
    hipLaunchKernelGGL(zero_strict_upper_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, A, lda, N);
  }
  for (int bi = 1; bi < nblk; ++bi)
    hipLaunchKernelGGL(trtri_row_kernel, dim3(bi), dim3(256), 0, s, A, lda, Linv, ldl, N, bi, info);
  MOE_HIP_CHECK(hipGetLastError());
}

}  // namespace moe
