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
#include <exception>
#include <algorithm>
#include <functional>
#include <cmath>
#include <cstdlib>

#include "gemm128.hpp"
#include "kernels.hpp"

namespace moe {

namespace {

constexpr int TK = 16;

// MODE 0: C = A^T B, A is K x m (element (k,i) at k + i*lda): full k range.
// MODE 1: C = T B,   T lower (element (i,k) at i + k*lda), k < i0 + TM.
// MODE 2: C = T^T B, T lower (element (k,i) at k + i*lda), k >= i0.
// MODE 3: C = A B,   A general M x K (element (i,k) at i + k*lda), B K x Ncols LOWER triangular: k >= j0.
// NEG: store -C (the off-diagonal blocks of a triangular inverse).
// KT = depth of one LDS stage: skinny outputs (few workgroups, latency-bound K loop) use deep stages so the loop has 4x
// fewer global-load / barrier round trips; big outputs keep 16 for occupancy.
template <int TM, int TN, int MODE, int KT, bool NEG = false>
struct tile_gemm_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, int M, int Ncols, int K, const double* __restrict__ A, long lda, const double* __restrict__ B, long ldb, double* __restrict__ C, long ldc) {
    constexpr int RM = TM / 16, RN = TN / 16;
    constexpr int TK = KT;
    __shared__ double As[TK][TM + 1];
    __shared__ double Bs[TK][TN + 1];
    const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;
    const int i0 = blockIdx.x * TM, j0 = blockIdx.y * TN;
    int k_lo = 0, k_hi = K;
    if (MODE == 1) k_hi = min(K, i0 + TM);
    if (MODE == 2) k_lo = (i0 / TK) * TK;
    if (MODE == 3) k_lo = (j0 / TK) * TK;
    double acc[RM][RN];
  #pragma unroll
    for (int a = 0; a < RM; ++a)
  #pragma unroll
      for (int b = 0; b < RN; ++b) acc[a][b] = 0.0;

    // The next stage's operand tiles travel global -> registers while the current stage is multiplied out of LDS: skinny
    // outputs leave a handful of workgroups walking K, and without the prefetch every stage pays a full memory round trip
    // (65 us for L^-1 (1000 x 1000) times 50 columns -- a quarter of a batch-1 q-KG evaluation's set-up).
    constexpr int NA = TK * TM / 256, NBV = TK * TN / 256;
    static_assert(TK * TM % 256 == 0 && TK * TN % 256 == 0, "operand tiles are whole multiples of the workgroup");
    double pa[NA], pb[NBV];
    auto fetch = [&](int k0) {
  #pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int t = threadIdx.x + i * 256;
        if (MODE == 1 || MODE == 3) {
          const int ii = t % TM, kk = t / TM;
          const int gi = i0 + ii, gk = k0 + kk;
          pa[i] = (gi < M && gk < K && (MODE == 3 || gk <= gi)) ? A[(long)gi + (long)gk * lda] : 0.0;
        } else {
          const int kk = t % TK, ii = t / TK;
          const int gi = i0 + ii, gk = k0 + kk;
          bool ok = gi < M && gk < K;
          if (MODE == 2) ok = ok && gk >= gi;
          pa[i] = ok ? A[(long)gk + (long)gi * lda] : 0.0;
        }
      }
  #pragma unroll
      for (int i = 0; i < NBV; ++i) {
        const int t = threadIdx.x + i * 256;
        const int kk = t % TK, jj = t / TK;
        const int gk = k0 + kk, gj = j0 + jj;
        pb[i] = (gk < K && gj < Ncols && (MODE != 3 || gk >= gj)) ? B[(long)gk + (long)gj * ldb] : 0.0;
      }
    };
    if (k_lo < k_hi) fetch(k_lo);
    for (int k0 = k_lo; k0 < k_hi; k0 += TK) {
      // stage the tiles: As[kk][ii] = Aop(i0+ii, k0+kk), Bs[kk][jj] = B(k0+kk, j0+jj)
  #pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int t = threadIdx.x + i * 256;
        if (MODE == 1 || MODE == 3)
          As[t / TM][t % TM] = pa[i];
        else
          As[t % TK][t / TK] = pa[i];
      }
  #pragma unroll
      for (int i = 0; i < NBV; ++i) {
        const int t = threadIdx.x + i * 256;
        Bs[t % TK][t / TK] = pb[i];
      }
      __syncthreads();
      if (k0 + TK < k_hi) fetch(k0 + TK);
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
        if (gi < M && gj < Ncols) C[(long)gi + (long)gj * ldc] = NEG ? -acc[a][b] : acc[a][b];
      }
  }
};
template <int TM, int TN, int MODE, int KT, bool NEG = false>
__global__ __launch_bounds__(256) void tile_gemm_kernel(int M, int Ncols, int K, const double* __restrict__ A, long lda,
                                                       const double* __restrict__ B, long ldb, double* __restrict__ C,
                                                       long ldc) {
  tile_gemm_kernel_body<TM, TN, MODE, KT, NEG>::run(MOE_VBLOCK, MOE_VGRID, nullptr, M, Ncols, K, A, lda, B, ldb, C, ldc);
}

// The same GEMM on the matrix pipe for big outputs: 64 x 64 tile per workgroup, 4 wavefronts, each owning a 32 x 32
// quadrant as 2 x 2 v_mfma_f64_16x16x4_f64 tiles.  The FP64 MFMA peak equals the FP64 vector peak on gfx950, but one MFMA
// retires 2048 flops from ONE operand pair (one f64 per lane each), so the LDS traffic per flop is 8x lower than the 4 x 4
// register-tile FMA loop above, which is LDS-issue bound at ~50 % of peak.  Operands are fed transposed (MFMA's A <- B
// tile, MFMA's B <- A tile) so that a result register's 16 consecutive lanes hold 16 consecutive ROWS of C: coalesced
// column-major stores.  The next K tile travels global -> registers while the current one is multiplied out of LDS.
using f64x4 = __attribute__((ext_vector_type(4))) double;

template <int MODE, bool NEG, int TKV = 16>
__global__ __launch_bounds__(256) void mfma_gemm_kernel(int M, int Ncols, int K, const double* __restrict__ A, long lda,
                                                       const double* __restrict__ B, long ldb, double* __restrict__ C,
                                                       long ldc, int xmul, int pair_rows, long sA = 0, long sB = 0,
                                                       long sC = 0, int m_total = 0, int m_step = 0, int tri_scale = 1) {
  constexpr int TM = 64, TN = 64, TK = TKV, LD = 65;
  constexpr int NF = TM * TK / 256;  // elements of each operand tile per thread
  __shared__ double As[TK][LD];
  __shared__ double Bs[TK][LD];
  // Column tiles are the FAST grid index: the workgroups that share a row tile of A (the big, streamed operand) are
  // dispatched together.  Triangular operands make a row tile's K range proportional to its index, and every workgroup of
  // the grid is resident from the start, so nothing rebalances a CU that drew long rows (measured: a third of the matrix
  // peak).  With pair_rows a workgroup does row tile p AND its mirror R - 1 - p, one after the other: every workgroup then
  // walks the same number of K steps.  Without it row indices are scattered (multiplier co-prime with the row count).
  // Split K (MODE 0 only, gridDim.z slices): slice z multiplies rows [z ks, (z + 1) ks) of both operands into the z-th of gridDim.z
  // partial results stacked behind C (ldc * Ncols doubles apart); launch_gemm_tn_splitk adds them in slice order.  For a skinny
  // output with a long K (S_W = W^T T of the d-KG tail: 32 x 20 000 over K = 8000) the unsplit grid is one workgroup per CU walking
  // 500 stages with one stage of loads in flight -- latency-bound at a third of what HBM delivers.
  // Batch (MODE != 0, gridDim.z problems of the same shape at element strides sA / sB / sC -- the nodes of one level of the
  // triangular inversion): problem z has min(M, m_total - z m_step) rows (the last node of a level may be cut off by the matrix
  // edge); for the triangular MODE 1 its K shrinks with it.
  if (MODE != 0 && gridDim.z > 1) {
    A += (long)blockIdx.z * sA;
    B += (long)blockIdx.z * sB;
    C += (long)blockIdx.z * sC;
    if (m_step > 0) {
      const int mz = min(M, m_total - (int)blockIdx.z * m_step);
      if (MODE == 1) K = min(K, mz);
      M = mz;
    }
  }
  if (MODE == 0 && gridDim.z > 1) {
    // (m_total = number of problems of the launch, 0 / 1: one; gridDim.z = problems x slices: z = slice * problems + problem.  The
    //  partial results of slice s, all problems, sit s * problems * ldc * Ncols behind C; problem e's at e * sC inside that)
    const int nprob = max(m_total, 1), slices = (int)gridDim.z / nprob;
    const int e = (int)blockIdx.z % nprob, sl = (int)blockIdx.z / nprob;
    const int ks = ((K + slices - 1) / slices + TK - 1) / TK * TK;
    const int kb = sl * ks;
    A += (long)e * sA + kb;
    B += (long)e * sB + kb;
    K = max(0, min(ks, K - kb));
    C += (long)sl * nprob * ldc * Ncols + (long)e * sC;
  }
  const int R = (M + TM - 1) / TM;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wi = (wave & 1) * 32, wj = (wave >> 1) * 32;  // this wavefront's quadrant
  const int lk = lane >> 4, lx = lane & 15;
  // pair_rows == 2 (MODE 3, r3): the triangular operand is B, whose K range shrinks with the COLUMN tile -- a workgroup does column
  // tile p and its mirror Ct - 1 - p (grid.x = ceil(Ct / 2)): the same balance for the L21 X11 product of the triangular inversion
  // (34 -> 45 TFLOP/s at the top level, where it is a third of the inversion's time)
  const bool pair_cols = MODE == 3 && pair_rows == 2;
  const int Ct = (Ncols + TN - 1) / TN;
  for (int half = 0; half < (pair_rows ? 2 : 1); ++half) {
    int bx, jt = blockIdx.x;
    if (pair_cols) {
      bx = (int)(((long)blockIdx.y * xmul) % gridDim.y);
      jt = half == 0 ? (int)blockIdx.x : Ct - 1 - (int)blockIdx.x;
      if (half == 1 && jt == (int)blockIdx.x) break;
    } else if (pair_rows) {
      bx = half == 0 ? (int)blockIdx.y : R - 1 - (int)blockIdx.y;
      if (half == 1 && bx == (int)blockIdx.y) break;  // odd tile count: the middle tile has no mirror
    } else {
      bx = (int)(((long)blockIdx.y * xmul) % gridDim.y);
    }
    const int j0 = jt * TN;
    const int i0 = bx * TM;
    int k_lo = 0, k_hi = K;
    if (MODE == 1) k_hi = min(K, i0 + TM);
    if (MODE == 2) k_lo = (int)(((long)i0 * tri_scale) / TK) * TK;  // (tri_scale: row i of A^T is column i * tri_scale of the triangle)
    if (MODE == 3) k_lo = (j0 / TK) * TK;
    f64x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};

    // software pipeline: tile k0 + TK travels global -> registers while tile k0 is multiplied out of LDS
    double ra[NF], rb[NF];
    auto fetch = [&](int k0) {
#pragma unroll
      for (int it = 0; it < NF; ++it) {
        const int t = threadIdx.x + 256 * it;
        if (MODE == 1 || MODE == 3) {
          const int ii = t % TM, kk = t / TM;
          const int gi = i0 + ii, gk = k0 + kk;
          ra[it] = (gi < M && gk < K && (MODE == 3 || gk <= gi)) ? A[(long)gi + (long)gk * lda] : 0.0;
        } else {
          const int kk = t % TK, ii = t / TK;
          const int gi = i0 + ii, gk = k0 + kk;
          bool ok = gi < M && gk < K;
          if (MODE == 2) ok = ok && gk >= gi * tri_scale;
          ra[it] = ok ? A[(long)gk + (long)gi * lda] : 0.0;
        }
        const int kk = t % TK, jj = t / TK;
        const int gk = k0 + kk, gj = j0 + jj;
        rb[it] = (gk < K && gj < Ncols && (MODE != 3 || gk >= gj)) ? B[(long)gk + (long)gj * ldb] : 0.0;
      }
    };
    auto stash = [&]() {
#pragma unroll
      for (int it = 0; it < NF; ++it) {
        const int t = threadIdx.x + 256 * it;
        if (MODE == 1 || MODE == 3)
          As[t / TM][t % TM] = ra[it];
        else
          As[t % TK][t / TK] = ra[it];
        Bs[t % TK][t / TK] = rb[it];
      }
    };
    if (k_lo < k_hi) fetch(k_lo);
    for (int k0 = k_lo; k0 < k_hi; k0 += TK) {
      stash();
      __syncthreads();
      if (k0 + TK < k_hi) fetch(k0 + TK);
#pragma unroll
      for (int k4 = 0; k4 < TK; k4 += 4) {
        double fa[2], fb[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) fa[a] = As[k4 + lk][wi + 16 * a + lx];  // -> MFMA's B operand: rows of C
#pragma unroll
        for (int b = 0; b < 2; ++b) fb[b] = Bs[k4 + lk][wj + 16 * b + lx];  // -> MFMA's A operand: columns of C
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[b], fa[a], acc[a][b], 0, 0, 0);
      }
      __syncthreads();
    }
    // D[x][y] = C[row = y][col = x]: lane holds y = lane & 15 (row of C), x = (lane >> 4) + 4 r (column of C)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gi = i0 + wi + 16 * a + lx, gj = j0 + wj + 16 * b + lk + 4 * r;
          if (gi < M && gj < Ncols) C[(long)gi + (long)gj * ldc] = NEG ? -acc[a][b][r] : acc[a][b][r];
        }
  }
}

template <int MODE, bool NEG = false>
void tile_gemm(int M, int Ncols, int K, const double* A, long lda, const double* B, long ldb, double* C, long ldc,
               hipStream_t s, int tri_scale = 1) {
  if (M <= 0 || Ncols <= 0) return;
  // Skinny outputs: smaller tiles give the chip more workgroups to place.
  const long blocks64 = (long)((M + 63) / 64) * ((Ncols + 63) / 64);
  static const long min_blocks = [] {
    const char* v = std::getenv("MOE_GEMM_MFMA_MIN_BLOCKS");
    return (v && *v) ? std::atol(v) : 48L;
  }();
  if (blocks64 >= min_blocks) {
    dim3 grid((M + 63) / 64, (Ncols + 63) / 64);
    // mfma_gemm_kernel: x = column tile, y = row tile -- or, for the triangular modes with enough rows, a PAIR of row
    // tiles (p, R - 1 - p) whose K ranges add up to the same total for every workgroup
    static const int pair_env = [] {
      const char* v = std::getenv("MOE_GEMM_PAIR_ROWS");
      return (v && *v) ? std::atoi(v) : 1;
    }();
    const int pair_rows = (pair_env != 0 && (MODE == 1 || MODE == 2) && grid.x >= 16 && (long)grid.x * grid.y >= 512) ? 1 : 0;
    const dim3 mgrid(grid.y, pair_rows ? (grid.x + 1) / 2 : grid.x);
    static const bool use_mfma = [] {
      const char* v = std::getenv("MOE_GEMM_MFMA");
      return !(v && *v == '0');
    }();
    if (use_mfma) {
      int xmul = 1;
      for (int cand : {37, 41, 43, 47, 53, 59})
        if ((int)grid.x % cand != 0) {
          xmul = cand;
          break;
        }
      static const int tk = [] {
        const char* v = std::getenv("MOE_GEMM_TK");
        return (v && *v) ? std::atoi(v) : 16;
      }();
      if (tk == 32)
        MOE_LAUNCH((mfma_gemm_kernel<MODE, NEG, 32>), mgrid, dim3(256), 0, s, M, Ncols, K, A, lda, B, ldb, C, ldc, xmul, pair_rows,
                           0L, 0L, 0L, 0, 0, tri_scale);
      else
        MOE_LAUNCH((mfma_gemm_kernel<MODE, NEG, 16>), mgrid, dim3(256), 0, s, M, Ncols, K, A, lda, B, ldb, C, ldc, xmul, pair_rows,
                           0L, 0L, 0L, 0, 0, tri_scale);
    }
    else
      launch_kernel_ens<tile_gemm_kernel_body<64, 64, MODE, 16, NEG>, 256>(tile_gemm_kernel<64, 64, MODE, 16, NEG>, grid, dim3(256), 0, s, M, Ncols, K, A,
                                                                           lda, B, ldb, C, ldc);
  } else {
    dim3 grid((M + 31) / 32, (Ncols + 15) / 16);
    launch_kernel_ens<tile_gemm_kernel_body<32, 16, MODE, 128, NEG>, 256>(tile_gemm_kernel<32, 16, MODE, 128, NEG>, grid, dim3(256), 0, s, M, Ncols, K, A,
                                                                          lda, B, ldb, C, ldc);
  }
  MOE_HIP_CHECK(hipGetLastError());
}

// gemm128.hpp's kernel: `batch` problems (GemmArgs strides), the LDS opt-in set on every call (it is per device).
template <bool AKC, int AMASK, bool BKC, int BMASK, bool NEG>
void launch_gemm128(const g128::GemmArgs& g, int batch, hipStream_t s) {
  if (g.M <= 0 || g.N <= 0 || batch <= 0) return;
  auto kern = g128::gemm128_kernel<AKC, AMASK, BKC, BMASK, NEG>;
  MOE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)g128::kSmemBytes));
  const int R = (g.M + g128::TM - 1) / g128::TM, Ct = (g.N + g128::TM - 1) / g128::TM;
  g128::GemmArgs ga = g;
  ga.batch = batch;
  MOE_LAUNCH(kern, dim3((unsigned)(R * Ct * batch)), dim3(256), g128::kSmemBytes, s, ga);
  MOE_HIP_CHECK(hipGetLastError());
}
// Which kernel a big product of the build takes: a cost estimate for either, from measured rates (MI355X, r4).  The 128-tile
// kernel retires a K step of 16 per tile in 2.1 us per CU with two workgroups resident (81 % of the matrix peak), 2.65 us with one;
// its launch lasts as long as its busiest CU, so few tiles (< 2 per workgroup slot) or one long tile among short ones (the
// triangular levels) cost it what the 64-tile kernels -- 44 - 46 TFLOP/s whatever the shape, four times as many workgroups, row /
// column pairing -- do not.  A function of the shape alone.  MOE_GEMM128 = 0 / 2: never / always (A/B runs, tests).
struct Gemm128Estimate {
  double total_units = 0.0, longest = 0.0, flops = 0.0;  // K steps of 16 over all tiles, of the longest tile; 2 x multiply-adds
  void add_tile(double k_len, double count = 1.0) {
    const double units = std::ceil(k_len / 16.0);
    total_units += units * count;
    longest = std::max(longest, units);
    flops += 2.0 * 128.0 * 128.0 * k_len * count;
  }
  double us128() const { return 1.25 * std::max(total_units * 2.1 / 256.0, longest * 2.65); }
  double us64() const { return flops / 44.0e6 + 10.0; }
};
inline int gemm128_mode() {
  const char* v = std::getenv("MOE_GEMM128");  // (read per call: the tests force both kernels at small sizes)
  return (v && *v) ? std::atoi(v) : 1;
}
inline bool use_gemm128(const Gemm128Estimate& e) {
  const int mode = gemm128_mode();
  if (mode == 0) return false;
  if (mode == 2) return true;
  return e.us128() < e.us64();
}
// the rank-kw update of a trailing block of `trailing` rows: equal tiles, whole rounds of 2 x 256 workgroups
inline bool use_syrk128(int trailing, int kw) {
  const int mode = gemm128_mode();
  if (mode == 0) return false;
  if (mode == 2) return true;
  const long T = (trailing + 127) / 128, tiles = T * (T + 1) / 2;
  const double per_round = kw / 16.0 * 4.2 * 1.02, alone = kw / 16.0 * 2.75;
  const double us128 = tiles <= 256 ? alone : std::ceil(tiles / 512.0) * per_round;
  const double us64 = (double)trailing * trailing * kw / 46.0e6 + 10.0;
  return us128 < us64;
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

struct gram_batch_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const GramMap& gm, int K, const double* __restrict__ V, long ldv, double* __restrict__ G, int slices) {
    constexpr int T = 32;
    constexpr int TK = 64;  // deep stages: the grid is tiny (c <= a few hundred), the K loop is latency-bound
    __shared__ double As[TK][T + 1];
    __shared__ double Bs[TK][T + 1];
    if (blockIdx.y > blockIdx.x) return;
    const int c = gm.m + gm.ng + gm.A;
    // blockIdx.z = evaluation * slices + K slice: with `slices` > 1 the output is a partial Gram per slice (summed in slice
    // order by gram_sum_kernel), which gives the few output tiles enough workgroups to hide the K loop's latency
    const int e = blockIdx.z / slices, sl = blockIdx.z % slices;
    const int kper = ((K + slices - 1) / slices + TK - 1) / TK * TK;
    const int k_begin = sl * kper, k_end = min(K, k_begin + kper);
    const int i0 = blockIdx.x * T, j0 = blockIdx.y * T;
    const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;
    double acc[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
    // (the next stage's columns travel global -> registers while the current stage is multiplied out of LDS: see tile_gemm_kernel)
    constexpr int NE = TK * T / 256;
    double pa[NE], pb[NE];
    auto fetch = [&](int k0) {
  #pragma unroll
      for (int i = 0; i < NE; ++i) {
        const int t = threadIdx.x + i * 256;
        const int kk = t % TK, ii = t / TK;
        const int gk = k0 + kk;
        const int gi = i0 + ii, gj = j0 + ii;
        pa[i] = (gi < c && gk < k_end) ? V[(long)gk + gm.col(e, gi) * ldv] : 0.0;
        pb[i] = (gj < c && gk < k_end) ? V[(long)gk + gm.col(e, gj) * ldv] : 0.0;
      }
    };
    if (k_begin < k_end) fetch(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += TK) {
  #pragma unroll
      for (int i = 0; i < NE; ++i) {
        const int t = threadIdx.x + i * 256;
        As[t % TK][t / TK] = pa[i];
        Bs[t % TK][t / TK] = pb[i];
      }
      __syncthreads();
      if (k0 + TK < k_end) fetch(k0 + TK);
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
    double* Ge = G + (long)blockIdx.z * c * c;
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
};
__global__ __launch_bounds__(256) void gram_batch_kernel(GramMap gm, int K, const double* __restrict__ V, long ldv,
                                                        double* __restrict__ G, int slices) {
  gram_batch_kernel_body::run(MOE_VBLOCK, MOE_VGRID, nullptr, gm, K, V, ldv, G, slices);
}

// Batched cross products X_e[r x m] = S_e^T W_e (r4, the KG state): S_e = the evaluation's `ng` gradient columns and `A` extra
// columns of the state matrix E (column map as in GramMap, rows l = 0 .. ng + A - 1 <-> GramMap column m + l), W_e = K^-1 K*_e
// (columns e m .. e m + m - 1 of W).  This is how the reference itself forms these blocks -- dK*^T (K^-1 K*), gpp_math.cpp:1277-1290 --
// and it means L^-1 is never applied to the gradient / extra columns (474 columns per evaluation at C5 against 32 of K*).
// 32 x 32 output tile per workgroup, blockIdx.z = evaluation * slices + K slice (partials summed in slice order).
struct gram_cross_batch_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const GramMap& gm, int K, const double* __restrict__ S, long lds, const double* __restrict__ W, long ldw, double* __restrict__ G, int slices) {
    constexpr int T = 32;
    constexpr int TK = 64;
    __shared__ double As[TK][T + 1];
    __shared__ double Bs[TK][T + 1];
    const int r = gm.ng + gm.A;
    const int e = blockIdx.z / slices, sl = blockIdx.z % slices;
    const int kper = ((K + slices - 1) / slices + TK - 1) / TK * TK;
    const int k_begin = sl * kper, k_end = min(K, k_begin + kper);
    const int i0 = blockIdx.x * T, j0 = blockIdx.y * T;
    const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;
    double acc[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
    constexpr int NE = TK * T / 256;
    double pa[NE], pb[NE];
    auto fetch = [&](int k0) {
  #pragma unroll
      for (int i = 0; i < NE; ++i) {
        const int t = threadIdx.x + i * 256;
        const int kk = t % TK, ii = t / TK;
        const int gk = k0 + kk;
        const int gi = i0 + ii, gj = j0 + ii;
        pa[i] = (gi < r && gk < k_end) ? S[(long)gk + gm.col(e, gm.m + gi) * lds] : 0.0;
        pb[i] = (gj < gm.m && gk < k_end) ? W[(long)gk + ((long)e * gm.m + gj) * ldw] : 0.0;
      }
    };
    if (k_begin < k_end) fetch(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += TK) {
  #pragma unroll
      for (int i = 0; i < NE; ++i) {
        const int t = threadIdx.x + i * 256;
        As[t % TK][t / TK] = pa[i];
        Bs[t % TK][t / TK] = pb[i];
      }
      __syncthreads();
      if (k0 + TK < k_end) fetch(k0 + TK);
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
    double* Ge = G + (long)blockIdx.z * r * gm.m;
  #pragma unroll
    for (int a = 0; a < 2; ++a)
  #pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int gi = i0 + tx + 16 * a, gj = j0 + ty + 16 * b;
        if (gi < r && gj < gm.m) Ge[(long)gi + (long)gj * r] = acc[a][b];
      }
  }
};
__global__ __launch_bounds__(256) void gram_cross_batch_kernel(GramMap gm, int K, const double* __restrict__ S, long lds,
                                                              const double* __restrict__ W, long ldw, double* __restrict__ G,
                                                              int slices) {
  gram_cross_batch_kernel_body::run(MOE_VBLOCK, MOE_VGRID, nullptr, gm, K, S, lds, W, ldw, G, slices);
}

// ------------------------------------------------------------------------------------------------------------------
// Blocked Cholesky (lower, in place) with explicit inverse factor.
// ------------------------------------------------------------------------------------------------------------------
constexpr int NB = 64;

// Factor the nb x nb diagonal block at (k0,k0); write L_kk back (strict upper zeroed) and its inverse into Linv's
// diagonal block.  One wavefront, everything fully unrolled so that lane t keeps ROW t of the block in registers (static
// indices): per elimination step one wave-uniform pivot (v_readlane), one column broadcast through LDS, and 63 - k
// independent fma -- no per-element LDS round trips.  The arithmetic (order of subtractions, division by the pivot's
// square root, the 1e-16 pivot rule of gpp_linear_algebra.cpp:118) is the reference's outer-product algorithm.  Rows /
// columns beyond nb (last, partial block) are padded with the identity.  Then lane t solves L x = e_t for column t of
// the inverse, again in registers against broadcast reads of L from LDS.
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(64) void chol_diag_kernel(double* __restrict__ A, long lda, double* __restrict__ Linv,
                                                      long ldl, int k0, int nb, int* __restrict__ info, long a_stride = 0,
                                                      long l_stride = 0) {
  __shared__ double S[NB][NB + 1];  // L, row-major
  __shared__ double X[NB][NB + 1];  // X[c][r] = (L^-1)[r][c]
  __shared__ double Ccol[NB];
  const int t = threadIdx.x;
  A += (long)blockIdx.z * a_stride;  // batch of independent matrices on grid.z (stride 0: one matrix)
  Linv += (long)blockIdx.z * l_stride;
  info += blockIdx.z;
  if (*info != 0) return;
  double a[NB];
#pragma clang loop unroll(full)
  for (int c = 0; c < NB; ++c) {
    double v = (c == t && t >= nb) ? 1.0 : 0.0;
    if (t < nb && c < nb && c <= t) v = A[(long)(k0 + t) + (long)(k0 + c) * lda];
    a[c] = v;
  }
  int bad = 0;
#pragma clang loop unroll(full)
  for (int k = 0; k < NB; ++k) {
    const double piv = readlane_f64(a[k], k);
    if (bad == 0 && !(piv > 1.0e-16)) bad = k0 + k + 1;  // gpp_linear_algebra.cpp:118 (first failing pivot; what follows it
                                                          // is garbage that is never written back)
    const double lkk = sqrt(piv);
    const double lik = (t == k) ? lkk : a[k] / lkk;
    a[k] = lik;
    Ccol[t] = lik;
    __syncthreads();
    // column k of L, broadcast from LDS in groups of 16 reads issued back to back (one LDS latency per group instead
    // of one per element); lanes t < j compute unused upper-triangle values
#pragma clang loop unroll(full)
    for (int j0 = ((k + 1) / 16) * 16; j0 < NB; j0 += 16) {
      double c[16];
#pragma clang loop unroll(full)
      for (int jj = 0; jj < 16; ++jj) c[jj] = Ccol[j0 + jj];
#pragma clang loop unroll(full)
      for (int jj = 0; jj < 16; ++jj)
        if (j0 + jj > k) a[j0 + jj] = a[j0 + jj] - lik * c[jj];
    }
    __syncthreads();
  }
  if (bad != 0) {
    if (t == 0) *info = bad;
    return;
  }
#pragma clang loop unroll(full)
  for (int c = 0; c < NB; ++c) {
    const double v = (c <= t) ? a[c] : 0.0;
    S[t][c] = v;
    if (t < nb && c < nb) A[(long)(k0 + t) + (long)(k0 + c) * lda] = v;  // strict upper written as 0
  }
  __syncthreads();
  // inverse: lane t solves L x = e_t by forward substitution (entries above t come out as exact zeros)
  double x[NB];
#pragma clang loop unroll(full)
  for (int i = 0; i < NB; ++i) {
    double sum = (i == t) ? 1.0 : 0.0;
#pragma clang loop unroll(full)
    for (int j0 = 0; j0 < i; j0 += 16) {  // row i of L in groups of 16 broadcast reads (see above)
      double r[16];
#pragma clang loop unroll(full)
      for (int jj = 0; jj < 16; ++jj) r[jj] = S[i][j0 + jj];  // j0 + jj < 64 always; entries >= i are not used
#pragma clang loop unroll(full)
      for (int jj = 0; jj < 16; ++jj)
        if (j0 + jj < i) sum -= r[jj] * x[j0 + jj];
    }
    x[i] = sum / S[i][i];
    X[t][i] = x[i];
  }
  __syncthreads();
  for (int c = 0; c < nb; ++c)
    if (t < nb) Linv[(long)(k0 + t) + (long)(k0 + c) * ldl] = X[c][t];
}

// Panel below the diagonal block: L_ik = A_ik * L_kk^-T, i.e. out[i][c] = sum_j A[i][j] * Linv_kk[c][j].
// One workgroup per 64 panel rows; both operands staged in LDS, 4 x 4 outputs per thread.
__global__ __launch_bounds__(256) void chol_panel_kernel(double* __restrict__ A, long lda, const double* __restrict__ Linv,
                                                        long ldl, int N, int k0, int nb, const int* __restrict__ info,
                                                        long a_stride = 0, long l_stride = 0) {
  __shared__ double D[NB][NB + 1];   // D[j][c] = Linv_kk[c][j]
  __shared__ double At[NB][NB + 1];  // At[j][i] = A[i0 + i][k0 + j]
  A += (long)blockIdx.z * a_stride;
  Linv += (long)blockIdx.z * l_stride;
  info += blockIdx.z;
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
                                                         const int* __restrict__ info, long a_stride = 0) {
  __shared__ double Ls_i[TK][NB + 1];
  __shared__ double Ls_j[TK][NB + 1];
  A += (long)blockIdx.z * a_stride;
  info += blockIdx.z;
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

__global__ void zero_strict_upper_kernel(double* __restrict__ A, long lda, int N) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)N * N;
  if (idx >= total) return;
  const int i = (int)(idx % N), j = (int)(idx / N);
  if (j > i) A[(long)i + (long)j * lda] = 0.0;
}

}  // namespace

namespace {
// Skinny triangular products (c <= 16 columns: K^-1 (y - mean), the block row of an append).  A tile GEMM with one or two
// column tiles leaves N / 64 workgroups walking the whole of K one 16-deep stage after the other (0.5 ms at N = 8000);
// these read T once, coalesced, with every wavefront of the chip holding loads in flight.
//
// C = T B:  one workgroup per strip of 64 rows (a row per lane), the K range dealt out to its wavefronts round-robin (each
// wavefront load is one 512-byte segment of a column), B[k, :] wave-uniform; partial sums meet in LDS in wavefront order.
template <int CB, int WAVES>
struct tri_skinny_n_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, int N, const double* __restrict__ T, long ldt, const double* __restrict__ B, long ldb, int c, double* __restrict__ C, long ldc) {
    // r4: the k range goes down in chunks of KC = 32 WAVES; per chunk a lane issues its 32 loads of T in ONE batch and the chunk's rows
    // of B are staged in LDS by the whole workgroup (coalesced), so a chunk costs one memory round trip -- the per-k version (one T load
    // and CB wave-uniform B loads per iteration, four iterations in flight) paid a round trip of 2 - 4 us every few k: 34 us for the
    // 500 x 500 factor of C2 against 10 columns.  Summation order of an entry: wave w adds its k = w, w + WAVES, ... in ascending order,
    // the waves' partial sums are added in wave order -- fixed by N alone.
    constexpr int TPL = 32;            // T loads per lane and chunk
    constexpr int KC = TPL * WAVES;    // k per chunk
    constexpr int HW = WAVES / 2;
    // dynamic LDS: red [WAVES / 2][CB][64] | Bs [KC][CB]   (64 KB at CB = 8 with 16 waves: opted in by the launcher)
    extern __shared__ __attribute__((aligned(16))) double sk_sm[];
    double(*red)[CB][64] = reinterpret_cast<double(*)[CB][64]>(sk_sm);
    double(*Bs)[CB] = reinterpret_cast<double(*)[CB]>(sk_sm + HW * CB * 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i0 = blockIdx.x * 64, i = i0 + lane, c0 = blockIdx.y * CB;
    const int kend = min(N, i0 + 64);
    const bool row_ok = i < N;
    const double* Trow = T + (row_ok ? i : 0);
    double acc[CB];
  #pragma unroll
    for (int cc = 0; cc < CB; ++cc) acc[cc] = 0.0;
    for (int k0 = 0; k0 < kend; k0 += KC) {
      double tv[TPL];
  #pragma unroll
      for (int t = 0; t < TPL; ++t) tv[t] = Trow[(long)min(k0 + w + WAVES * t, kend - 1) * ldt];  // (clamped; masked below)
      for (int t = threadIdx.x; t < KC * CB; t += WAVES * 64) {
        const int kk = t % KC, cc = t / KC;
        Bs[kk][cc] = (k0 + kk < kend && c0 + cc < c) ? B[(long)(k0 + kk) + (long)(c0 + cc) * ldb] : 0.0;
      }
      __syncthreads();
  #pragma unroll
      for (int t = 0; t < TPL; ++t) {
        const int kk = w + WAVES * t, k = k0 + kk;
        const double tm = (row_ok && k <= i && k < kend) ? tv[t] : 0.0;
  #pragma unroll
        for (int cc = 0; cc < CB; ++cc) acc[cc] = fma(tm, Bs[kk][cc], acc[cc]);
      }
      __syncthreads();
    }
    // waves HW .. WAVES - 1 hand their sums to waves 0 .. HW - 1 (w + HW -> w), then the HW partial sums are added in wave order
    if (w >= HW) {
  #pragma unroll
      for (int cc = 0; cc < CB; ++cc) red[w - HW][cc][lane] = acc[cc];
    }
    __syncthreads();
    if (w < HW) {
  #pragma unroll
      for (int cc = 0; cc < CB; ++cc) acc[cc] += red[w][cc][lane];
    }
    __syncthreads();
    if (w < HW) {
  #pragma unroll
      for (int cc = 0; cc < CB; ++cc) red[w][cc][lane] = acc[cc];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < CB * 64; t += WAVES * 64) {
      const int cc = t >> 6, l = t & 63;
      double v = 0.0;
  #pragma unroll
      for (int ww = 0; ww < HW; ++ww) v += red[ww][cc][l];
      if (i0 + l < N && c0 + cc < c) C[(long)(i0 + l) + (long)(c0 + cc) * ldc] = v;
    }
  }
};
template <int CB, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void tri_skinny_n_kernel(int N, const double* __restrict__ T, long ldt,
                                                                  const double* __restrict__ B, long ldb, int c,
                                                                  double* __restrict__ C, long ldc) {
  tri_skinny_n_kernel_body<CB, WAVES>::run(MOE_VBLOCK, MOE_VGRID, nullptr, N, T, ldt, B, ldb, c, C, ldc);
}

// C = T^T B:  one wavefront per column of T (contiguous), lanes stride down the rows from the 64-aligned row at or above
// the diagonal, fixed-order butterfly at the end.
// TJ (r6): a wavefront takes TJ consecutive columns of T (TJ | 64: they start at the same 64-aligned row) and reads every row of B once
// for all of them -- with one column per wavefront the kernel issued 1 + CB loads per CB fmas and ran at the rate of the load pipe
// (470 us for L^-T on 80 columns of 16 headline-sized GPs); an entry's summation order is that of the one-column form.
template <int CB, int TJ = 1>
struct tri_skinny_t_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, int N, const double* __restrict__ T, long ldt, const double* __restrict__ B, long ldb, int c, double* __restrict__ C, long ldc) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j0 = (blockIdx.x * 4 + w) * TJ, c0 = blockIdx.y * CB;
    if (j0 >= N) return;
    const double* col[TJ];
  #pragma unroll
    for (int t = 0; t < TJ; ++t) col[t] = T + (long)min(j0 + t, N - 1) * ldt;
    double acc[TJ][CB];
  #pragma unroll
    for (int t = 0; t < TJ; ++t)
  #pragma unroll
      for (int cc = 0; cc < CB; ++cc) acc[t][cc] = 0.0;
    const double* Bc[CB];
  #pragma unroll
    for (int cc = 0; cc < CB; ++cc) Bc[cc] = B + (long)min(c0 + cc, c - 1) * ldb;
  #pragma unroll(TJ > 1 ? 2 : (CB > 8 ? 4 : 8))
    for (int i = (j0 & ~63) + lane; i < N; i += 64) {  // (unconditional loads, masked afterwards: see tri_skinny_n_kernel)
      double bv[CB];
  #pragma unroll
      for (int cc = 0; cc < CB; ++cc) bv[cc] = Bc[cc][i];
  #pragma unroll
      for (int t = 0; t < TJ; ++t) {
        const double tv = col[t][i];
        const double tm = (i >= j0 + t) ? tv : 0.0;
  #pragma unroll
        for (int cc = 0; cc < CB; ++cc) acc[t][cc] = fma(tm, bv[cc], acc[t][cc]);
      }
    }
  #pragma unroll
    for (int t = 0; t < TJ; ++t) {
  #pragma unroll
      for (int cc = 0; cc < CB; ++cc) {
        double v = acc[t][cc];
  #pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0 && c0 + cc < c && j0 + t < N) C[(long)(j0 + t) + (long)(c0 + cc) * ldc] = v;
      }
    }
  }
};
template <int CB, int TJ = 1>
__global__ __launch_bounds__(256) void tri_skinny_t_kernel(int N, const double* __restrict__ T, long ldt,
                                                          const double* __restrict__ B, long ldb, int c,
                                                          double* __restrict__ C, long ldc) {
  tri_skinny_t_kernel_body<CB, TJ>::run(MOE_VBLOCK, MOE_VGRID, nullptr, N, T, ldt, B, ldb, c, C, ldc);
}

template <int CB>
void launch_tri_skinny(char op, int N, int c, const double* T, long ldt, const double* B, long ldb, double* C, long ldc,
                       hipStream_t s) {
  const int groups = (c + CB - 1) / CB;
  if (op == 'N') {
    // (LDS: reduction slots 8 x CB x 64 + B chunk 512 x CB doubles: 64 KB at CB = 8)
    auto kern = tri_skinny_n_kernel<CB, 16>;
    const size_t shm = sizeof(double) * ((size_t)8 * CB * 64 + (size_t)512 * CB);
    if (shm > 48 * 1024)
      MOE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    launch_kernel_ens<tri_skinny_n_kernel_body<CB, 16>, 1024>(kern, dim3((N + 63) / 64, groups), dim3(1024), shm, s, N, T, ldt, B, ldb, c, C, ldc);
  } else {
    // (several columns of T per wavefront where the call has the columns of many evaluations: MOE_TRI_SKINNY_TJ=1: one, A/B runs)
    const char* tjv = std::getenv("MOE_TRI_SKINNY_TJ");  // (read per call: tests/test_gpu_sweep.py compares the two forms)
    const bool tj_on = !(tjv != nullptr && std::atoi(tjv) == 1);
    if (CB >= 8 && tj_on)
      launch_kernel_ens<tri_skinny_t_kernel_body<CB, 4>, 256>(tri_skinny_t_kernel<CB, 4>, dim3((N + 15) / 16, groups), dim3(256), 0, s, N, T, ldt, B, ldb,
                                                              c, C, ldc);
    else
      launch_kernel_ens<tri_skinny_t_kernel_body<CB>, 256>(tri_skinny_t_kernel<CB>, dim3((N + 3) / 4, groups), dim3(256), 0, s, N, T, ldt, B, ldb, c, C, ldc);
  }
  MOE_HIP_CHECK(hipGetLastError());
}
}  // namespace

namespace {
// Triangular product against a SKINNY right-hand side (r4: the KG state applies L^-1 / L^-T to the m columns of K* per evaluation, and
// the gradient tail to the m columns of TB -- 32 at C5, where the plain tiled kernels leave N / 64 workgroups walking K = 8000 each:
// 0.5 ms for 8 GFLOP, 16 TFLOP/s, a sixth of what reading T from HBM costs).  Split K on the matrix pipe: workgroup (column tile,
// row tile, slice) multiplies the 64 x 64 output tile over ONE slice [sl KS, (sl + 1) KS) of the k range its row tile needs and
// stores the partial product to work[sl]; tri_splitk_sum_kernel adds the partials of a row in ascending slice order.  KS depends on N
// alone, so an entry's summation order does not depend on how many columns (evaluations) share the call.
// MODE 1: C = T B (T lower: row tile r needs k < (r + 1) 64).  MODE 2: C = T^T B (k >= r 64).
template <int MODE, int TK = 16>
struct tri_splitk_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, int N, int c, int KS, const double* __restrict__ T, long ldt, const double* __restrict__ B, long ldb, double* __restrict__ work) {
    constexpr int TM = 64, LD = 65;
    constexpr int NF = TM * TK / 256;
    __shared__ double As[TK][LD];
    __shared__ double Bs[TK][LD];
    const int j0 = blockIdx.x * 64, i0 = blockIdx.y * TM, sl = blockIdx.z;
    int k_lo = sl * KS, k_hi = min(N, k_lo + KS);
    if (MODE == 1) k_hi = min(k_hi, min(N, i0 + TM));
    if (MODE == 2) k_lo = max(k_lo, (i0 / TK) * TK);
    if (k_lo >= k_hi) return;  // this row tile has no work in this slice (tri_splitk_sum_kernel skips it, too)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wi = (wave & 1) * 32, wj = (wave >> 1) * 32;
    const int lk = lane >> 4, lx = lane & 15;
    f64x4 acc[2][2];
  #pragma unroll
    for (int a = 0; a < 2; ++a)
  #pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
    double ra[NF], rb[NF];
    auto fetch = [&](int k0) {
  #pragma unroll
      for (int it = 0; it < NF; ++it) {
        const int t = threadIdx.x + 256 * it;
        if (MODE == 1) {
          const int ii = t % TM, kk = t / TM;
          const int gi = i0 + ii, gk = k0 + kk;
          ra[it] = (gi < N && gk < k_hi && gk <= gi) ? T[(long)gi + (long)gk * ldt] : 0.0;
        } else {
          const int kk = t % TK, ii = t / TK;
          const int gi = i0 + ii, gk = k0 + kk;
          ra[it] = (gi < N && gk < k_hi && gk >= gi) ? T[(long)gk + (long)gi * ldt] : 0.0;
        }
        const int kk = t % TK, jj = t / TK;
        const int gk = k0 + kk, gj = j0 + jj;
        rb[it] = (gk < k_hi && gj < c) ? B[(long)gk + (long)gj * ldb] : 0.0;
      }
    };
    fetch(k_lo);
    for (int k0 = k_lo; k0 < k_hi; k0 += TK) {
  #pragma unroll
      for (int it = 0; it < NF; ++it) {
        const int t = threadIdx.x + 256 * it;
        if (MODE == 1)
          As[t / TM][t % TM] = ra[it];
        else
          As[t % TK][t / TK] = ra[it];
        Bs[t % TK][t / TK] = rb[it];
      }
      __syncthreads();
      if (k0 + TK < k_hi) fetch(k0 + TK);
  #pragma unroll
      for (int k4 = 0; k4 < TK; k4 += 4) {
        double fa[2], fb[2];
  #pragma unroll
        for (int a = 0; a < 2; ++a) fa[a] = As[k4 + lk][wi + 16 * a + lx];
  #pragma unroll
        for (int b = 0; b < 2; ++b) fb[b] = Bs[k4 + lk][wj + 16 * b + lx];
  #pragma unroll
        for (int a = 0; a < 2; ++a)
  #pragma unroll
          for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[b], fa[a], acc[a][b], 0, 0, 0);
      }
      __syncthreads();
    }
    double* W = work + (long)sl * N * c;
  #pragma unroll
    for (int a = 0; a < 2; ++a)
  #pragma unroll
      for (int b = 0; b < 2; ++b)
  #pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gi = i0 + wi + 16 * a + lx, gj = j0 + wj + 16 * b + lk + 4 * r;
          if (gi < N && gj < c) W[(long)gi + (long)gj * N] = acc[a][b][r];
        }
  }
};
template <int MODE, int TK = 16>
__global__ __launch_bounds__(256) void tri_splitk_kernel(int N, int c, int KS, const double* __restrict__ T, long ldt,
                                                        const double* __restrict__ B, long ldb, double* __restrict__ work) {
  tri_splitk_kernel_body<MODE, TK>::run(MOE_VBLOCK, MOE_VGRID, nullptr, N, c, KS, T, ldt, B, ldb, work);
}

template <int MODE>
struct tri_splitk_sum_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, int N, int c, int KS, int slices, const double* __restrict__ work, double* __restrict__ C, long ldc) {
    const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (i >= N) return;
    const int rt = i / 64;  // the slices row tile rt took part in (the k ranges of tri_splitk_kernel)
    int s_lo = 0, s_hi = slices;
    if (MODE == 1) s_hi = min(slices, (min(N, (rt + 1) * 64) - 1) / KS + 1);
    if (MODE == 2) s_lo = ((rt * 64) / 16 * 16) / KS;
    double v = 0.0;
    for (int sl = s_lo; sl < s_hi; ++sl) v += work[((long)sl * c + j) * N + i];
    C[(long)i + (long)j * ldc] = v;
  }
};
template <int MODE>
__global__ __launch_bounds__(256) void tri_splitk_sum_kernel(int N, int c, int KS, int slices, const double* __restrict__ work,
                                                            double* __restrict__ C, long ldc) {
  tri_splitk_sum_kernel_body<MODE>::run(MOE_VBLOCK, MOE_VGRID, nullptr, N, c, KS, slices, work, C, ldc);
}
}  // namespace

int tri_cols_slices(int N, int* ks_out) {
  const int tiles = (N + 63) / 64;
  const int ks = 64 * std::max(1, (tiles + 7) / 8);
  if (ks_out) *ks_out = ks;
  return (N + ks - 1) / ks;
}
size_t tri_cols_work_doubles(int N, int c) { return (size_t)tri_cols_slices(N, nullptr) * (size_t)N * (size_t)c; }

void launch_tri_gemm_cols(char op, int N, int c, int cols_per_problem, const double* T, long ldt, const double* B, long ldb,
                          double* C, long ldc, double* work, hipStream_t s) {
  if (N <= 0 || c <= 0) return;
  static const int mode = [] {
    const char* v = std::getenv("MOE_TRI_COLS");  // 0: the plain tiled kernels (A/B runs)
    return (v && *v) ? std::atoi(v) : 1;
  }();
  if (mode == 0 || N < 128) {  // (tiny factors: one or two row tiles, the plain tiled kernel)
    launch_tri_gemm(op, N, c, T, ldt, B, ldb, C, ldc, s);
    return;
  }
  if (cols_per_problem <= 16) {  // a handful of columns per problem: the row-strip / column-per-wavefront kernels
    // Columns per workgroup (r6: from the CALL's column count, up to 16).  A column's arithmetic does not depend on how many columns share
    // its workgroup -- the same k ranges per wavefront, the same order of the partial sums -- so the grouping is free to follow the batch:
    // an ensemble of 16 headline-sized GPs x 20 evaluations x 4 columns re-read every member's factor 20 times per product with groups
    // of 4 (313 + 470 us per pair of products, a third of an optimiser step of the C3-sized suggestion).  MOE_TRI_SKINNY_CB = 4 / 8 / 16
    // forces a group size (A/B runs; tests/test_gpu_sweep.py compares the results bit for bit).
    const char* cbv = std::getenv("MOE_TRI_SKINNY_CB");
    const int forced = (cbv && *cbv) ? std::atoi(cbv) : 0;
    const int cb = (forced == 4 || forced == 8 || forced == 16) ? forced : (c <= 4 ? 4 : 8);  // (16: slower for either kernel, profiles/r06_af_*, r06_ah_*)
    if (cb == 4)
      launch_tri_skinny<4>(op, N, c, T, ldt, B, ldb, C, ldc, s);
    else if (cb == 8)
      launch_tri_skinny<8>(op, N, c, T, ldt, B, ldb, C, ldc, s);
    else
      launch_tri_skinny<16>(op, N, c, T, ldt, B, ldb, C, ldc, s);
    return;
  }
  int KS = 0;
  const int slices = tri_cols_slices(N, &KS);
  const dim3 grid((c + 63) / 64, (N + 63) / 64, slices);
  const dim3 sgrid((N + 255) / 256, c);
  static const int tk = [] {
    const char* v = std::getenv("MOE_TRI_SPLITK_TK");
    return (v && *v) ? std::atoi(v) : 16;
  }();
  if (op == 'N') {
    if (tk == 32)
      launch_kernel_ens<tri_splitk_kernel_body<1, 32>, 256>(tri_splitk_kernel<1, 32>, grid, dim3(256), 0, s, N, c, KS, T, ldt, B, ldb, work);
    else
      launch_kernel_ens<tri_splitk_kernel_body<1, 16>, 256>(tri_splitk_kernel<1, 16>, grid, dim3(256), 0, s, N, c, KS, T, ldt, B, ldb, work);
    launch_kernel_ens<tri_splitk_sum_kernel_body<1>, 256>(tri_splitk_sum_kernel<1>, sgrid, dim3(256), 0, s, N, c, KS, slices, (const double*)work, C, ldc);
  } else {
    if (tk == 32)
      launch_kernel_ens<tri_splitk_kernel_body<2, 32>, 256>(tri_splitk_kernel<2, 32>, grid, dim3(256), 0, s, N, c, KS, T, ldt, B, ldb, work);
    else
      launch_kernel_ens<tri_splitk_kernel_body<2, 16>, 256>(tri_splitk_kernel<2, 16>, grid, dim3(256), 0, s, N, c, KS, T, ldt, B, ldb, work);
    launch_kernel_ens<tri_splitk_sum_kernel_body<2>, 256>(tri_splitk_sum_kernel<2>, sgrid, dim3(256), 0, s, N, c, KS, slices, (const double*)work, C, ldc);
  }
  MOE_HIP_CHECK(hipGetLastError());
}

void launch_tri_gemm(char op, int N, int c, const double* T, long ldt, const double* B, long ldb, double* C, long ldc,
                     hipStream_t s) {
  if (op == 'N')
    tile_gemm<1>(N, c, N, T, ldt, B, ldb, C, ldc, s);
  else
    tile_gemm<2>(N, c, N, T, ldt, B, ldb, C, ldc, s);
}

namespace {
// out[r] = sum_k T[k + r ldt]^2 over k >= r: the squared column norms of a lower-triangular matrix (diag of T^T T).  One
// wavefront per column, fixed summation order.
__global__ __launch_bounds__(256) void tri_colnorm2_kernel(int N, const double* __restrict__ T, long ldt, double* __restrict__ out) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= N) return;
  const double* col = T + (long)r * ldt;
  double v = 0.0;
  for (int k = r + lane; k < N; k += 64) v = fma(col[k], col[k], v);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  if (lane == 0) out[r] = v;
}
}  // namespace

void launch_tri_gram_strided(int N, int c, int stride, const double* T, long ldt, double* G, long ldg, double* diag, hipStream_t s) {
  // rows of S^T = every stride-th column of T: A(i, k) = T[k + (i stride) ldt] -- a transposed triangular operand with leading
  // dimension stride * ldt whose row i starts at k = i stride
  tile_gemm<2>(c, c, N, T, (long)stride * ldt, T, (long)stride * ldt, G, ldg, s, stride);
  if (diag != nullptr) {
    MOE_LAUNCH(tri_colnorm2_kernel, dim3((N + 3) / 4), dim3(256), 0, s, N, T, ldt, diag);
    MOE_HIP_CHECK(hipGetLastError());
  }
}

void launch_tri_gemm_skinny(char op, int N, int c, const double* T, long ldt, const double* B, long ldb, double* C,
                            long ldc, hipStream_t s) {
  if (N <= 0 || c <= 0) return;
  static const bool skinny = [] {
    const char* v = std::getenv("MOE_TRI_SKINNY");
    return !(v && *v == '0');
  }();
  if (!skinny || c > 16 || N < 128) {
    launch_tri_gemm(op, N, c, T, ldt, B, ldb, C, ldc, s);
    return;
  }
  if (c == 1)
    launch_tri_skinny<1>(op, N, c, T, ldt, B, ldb, C, ldc, s);
  else if (c <= 4)
    launch_tri_skinny<4>(op, N, c, T, ldt, B, ldb, C, ldc, s);
  else
    launch_tri_skinny<8>(op, N, c, T, ldt, B, ldb, C, ldc, s);
}

namespace {
// out[c] = sum_k A[k + c * lda] x[k]: one workgroup per column, coalesced reads, fixed-order reduction.  (A^T x as a tile
// GEMM with ONE output column leaves a handful of workgroups walking K serially: 0.84 ms at 8000 x 474.)
struct gemv_t_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, int K, const double* __restrict__ A, long lda, const double* __restrict__ x, double* __restrict__ out) {
    __shared__ double red[4];
    const double* col = A + (long)blockIdx.x * lda;
    double acc0 = 0.0, acc1 = 0.0;
    int k = threadIdx.x;
    for (; k + 256 < K; k += 512) {
      acc0 = fma(col[k], x[k], acc0);
      acc1 = fma(col[k + 256], x[k + 256], acc1);
    }
    if (k < K) acc0 = fma(col[k], x[k], acc0);
    double v = acc0 + acc1;
  #pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
};
__global__ __launch_bounds__(256) void gemv_t_kernel(int K, const double* __restrict__ A, long lda,
                                                    const double* __restrict__ x, double* __restrict__ out) {
  gemv_t_kernel_body::run(MOE_VBLOCK, MOE_VGRID, nullptr, K, A, lda, x, out);
}
}  // namespace

namespace {
__global__ __launch_bounds__(256) void sum_slices_kernel(const double* __restrict__ part, int slices, long count,
                                                        double* __restrict__ out) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= count) return;
  double v = 0.0;
  for (int z = 0; z < slices; ++z) v += part[(long)z * count + idx];  // slice order: fixed
  out[idx] = v;
}
}  // namespace

// C (m x n, ldc == m) = A^T B with K cut into `slices` (mfma_gemm_kernel's split-K); work holds slices * m * n doubles.
void launch_gemm_tn_splitk(int m, int n, int K, const double* A, long lda, const double* B, long ldb, double* C, double* work,
                           int slices, hipStream_t s, int batch, long sA, long sB) {
  if (m <= 0 || n <= 0 || batch <= 0) return;
  const dim3 mgrid((n + 63) / 64, (m + 63) / 64, slices * batch);
  int xmul = 1;
  for (int cand : {37, 41, 43, 47, 53, 59})
    if ((int)mgrid.y % cand != 0) {
      xmul = cand;
      break;
    }
  // problem e's m x n result is dense (ldc == m) at e * m * n, in C and inside every slice of work
  MOE_LAUNCH((mfma_gemm_kernel<0, false, 16>), mgrid, dim3(256), 0, s, m, n, K, A, lda, B, ldb, slices > 1 ? work : C, (long)m,
                     xmul, 0, sA, sB, (long)m * n, batch, 0, 1);
  if (slices > 1) {
    const long count = (long)m * n * batch;
    MOE_LAUNCH(sum_slices_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, (const double*)work, slices, count, C);
  }
  MOE_HIP_CHECK(hipGetLastError());
}

void launch_sum_slices(const double* work, int slices, long count, double* C, hipStream_t s) {
  MOE_LAUNCH(sum_slices_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, work, slices, count, C);
  MOE_HIP_CHECK(hipGetLastError());
}

void launch_gemm_tn(int m, int n, int K, const double* A, long lda, const double* B, long ldb, double* C, long ldc,
                    hipStream_t s) {
  if (n == 1 && m > 0) {
    launch_kernel_ens<gemv_t_kernel_body, 256>(gemv_t_kernel, dim3(m), dim3(256), 0, s, K, A, lda, B, C);
    MOE_HIP_CHECK(hipGetLastError());
    return;
  }
  tile_gemm<0>(m, n, K, A, lda, B, ldb, C, ldc, s);
}

namespace {
// G[e] = sum over the K slices of the partial Grams, in slice order.
struct gram_sum_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const double* __restrict__ part, int slices, long cc, double* __restrict__ G) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int e = blockIdx.y;
    if (idx >= cc) return;
    const double* p = part + (long)e * slices * cc + idx;
    double v = 0.0;
    for (int sl = 0; sl < slices; ++sl) v += p[(long)sl * cc];
    G[(long)e * cc + idx] = v;
  }
};
__global__ __launch_bounds__(256) void gram_sum_kernel(const double* __restrict__ part, int slices, long cc, double* __restrict__ G) {
  gram_sum_kernel_body::run(MOE_VBLOCK, MOE_VGRID, nullptr, part, slices, cc, G);
}
}  // namespace

int gram_batch_slices(int /*E*/, int c, int K) {
  // The K split fixes the summation order of every Gram entry, so it must not depend on how many evaluations share the call
  // (r3: a restart evaluated alone, in a batch of 8 or in a batch of 64 has to give the same bits -- with the batch size in
  // this formula a batch of 64 took 6 slices where smaller ones took 15 and the results moved by an ulp): sized for ONE
  // evaluation; bigger batches simply bring more workgroups.
  const long tiles = (long)((c + 31) / 32) * ((c + 31) / 32 + 1) / 2;  // lower-triangular output tiles of one evaluation
  const long want = 1024;                                                  // ~4 workgroups per CU
  long s = (want + tiles - 1) / tiles;
  s = std::min<long>(s, std::max(1, K / 64));                              // at least one stage of 64 per slice
  return (int)std::max<long>(1, std::min<long>(s, 16));
}

// K slices of gram_cross_batch_kernel: sized for ONE evaluation (the summation order must not depend on the batch size)
int gram_cross_slices(int m, int ng, int A, int K) {
  const long tiles = (long)((ng + A + 31) / 32) * ((m + 31) / 32);
  if (tiles <= 0) return 1;
  long s = (512 + tiles - 1) / tiles;
  s = std::min<long>(s, std::max(1, K / 64));
  return (int)std::max<long>(1, std::min<long>(s, 16));
}

int launch_gram_cross_batch(int E, int m, int ng, int A, int K, const double* S, long lds, const double* W, long ldw, double* G,
                            double* work, hipStream_t s, bool defer_sum) {
  const int r = ng + A;
  if (r <= 0 || m <= 0 || E <= 0) return 1;
  GramMap gm{E, m, ng, A};
  const int slices = gram_cross_slices(m, ng, A, K);
  dim3 grid((r + 31) / 32, (m + 31) / 32, E * slices);
  if (slices == 1) {
    launch_kernel_ens<gram_cross_batch_kernel_body, 256>(gram_cross_batch_kernel, grid, dim3(256), 0, s, gm, K, S, lds, W, ldw, G, 1);
  } else {
    const long cc = (long)r * m;
    launch_kernel_ens<gram_cross_batch_kernel_body, 256>(gram_cross_batch_kernel, grid, dim3(256), 0, s, gm, K, S, lds, W, ldw, work, slices);
    if (!defer_sum)
      launch_kernel_ens<gram_sum_kernel_body, 256>(gram_sum_kernel, dim3((unsigned)((cc + 255) / 256), E), dim3(256), 0, s, (const double*)work, slices, cc, G);
  }
  MOE_HIP_CHECK(hipGetLastError());
  return slices;
}

int launch_gram_batch(int E, int m, int ng, int A, int K, const double* V, long ldv, double* G, double* work, hipStream_t s,
                      bool defer_sum) {
  const int c = m + ng + A;
  if (c <= 0 || E <= 0) return 1;
  GramMap gm{E, m, ng, A};
  const int slices = (work != nullptr) ? gram_batch_slices(E, c, K) : 1;
  dim3 grid((c + 31) / 32, (c + 31) / 32, E * slices);
  if (slices == 1) {
    launch_kernel_ens<gram_batch_kernel_body, 256>(gram_batch_kernel, grid, dim3(256), 0, s, gm, K, V, ldv, G, 1);
  } else {
    const long cc = (long)c * c;
    launch_kernel_ens<gram_batch_kernel_body, 256>(gram_batch_kernel, grid, dim3(256), 0, s, gm, K, V, ldv, work, slices);
    if (!defer_sum)
      launch_kernel_ens<gram_sum_kernel_body, 256>(gram_sum_kernel, dim3((unsigned)((cc + 255) / 256), E), dim3(256), 0, s, (const double*)work, slices, cc, G);
  }
  MOE_HIP_CHECK(hipGetLastError());
  return slices;
}


// ---------------------------------------------------------------------------------------------------------------------
// Two-level blocked factorisation for large N (C5: N = 8000; its stretch point N = 26 000).  The one-level algorithm above
// pays, per 64 columns, a single-wavefront diagonal kernel whose fully unrolled 64 x 64 body does not fit the instruction
// cache (125 launches of ~150 us at N = 8000) and a rank-64 trailing update that re-reads and re-writes the whole trailing
// matrix (HBM-bound: ~43 GB at N = 8000).  Here columns are grouped into OUTER blocks of kOuter (512):
//   * inside an outer block the 64-column steps only update the columns of that block (a tall, narrow region);
//   * the diagonal 64 x 64 factor + inverse is a 256-thread LDS kernel with rolled loops (chol_diag_lds_kernel);
//   * once an outer block is done, ONE rank-512 update of the trailing matrix runs on the matrix pipe
//     (syrk_mfma_kernel, v_mfma_f64_16x16x4_f64): the trailing matrix makes N / 512 round trips through HBM instead of N / 64,
//     and 8 x more flops per byte is what lets the MFMA tiles run compute-bound.
// Same pivot rule (1e-16, gpp_linear_algebra.cpp:118) and the same error report as the one-level path.
// ---------------------------------------------------------------------------------------------------------------------
#ifndef MOE_CHOL_OUTER
#define MOE_CHOL_OUTER 512
#endif
constexpr int kOuter = MOE_CHOL_OUTER;

// Diagonal 64 x 64 block: factor AND invert, 256 threads, everything in LDS, in 16-column sub-steps:
//   (1) the 16 x 16 diagonal sub-block is factored and inverted by 16 lanes holding one row each in registers (16 unrolled
//       steps: v_readlane pivot, division by its square root, the 1e-16 pivot rule -- the reference's outer-product order);
//   (2) the rows below it get  A_panel L16^-T  through that inverse;   (3) the remaining columns get their rank-16 update,
//       folded in one column at a time (fma in k order: the unblocked algorithm's rounding);
// then L^-1 by recursive halving (16 -> 32 -> 64) with small LDS GEMMs:  inv [[L11, 0], [L21, L22]] = [[X11, 0], [-X22 L21 X11, X22]].
// ~10 us per block instead of ~110 us for a scalar 64-step loop with three barriers per column (measured), or the 46 - 250 us of
// the register-resident single-wavefront kernel above, whose fully unrolled body does not fit the instruction cache.
constexpr int SB = 16;

// L (lower triangle incl. diagonal) and X = L^-1 (lower triangle) share ONE 64 x 65 LDS array: X[i][j], j <= i, lives in the
// strict upper part at S[j][i + 1]  (50 KB of LDS for the kernel instead of 100).
#define MOE_XS(i, j) S[(j)][(i) + 1]

#if defined(MOE_DIAG_PROF)  // tools/diagbench.hip: clock stamps of thread 0 -- [0..4] phase ends, [8..11] time inside the four parts of a sub-step
__device__ unsigned long long moe_diag_prof[16];
#define MOE_DIAG_T(i) \
  if (threadIdx.x == 0) moe_diag_prof[i] = __builtin_amdgcn_s_memtime();
#define MOE_DIAG_A(i)                                                     \
  if (threadIdx.x == 0) {                                                 \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();         \
    if ((i) > 0) moe_diag_prof[8 + (i)] += now_ - diag_last_;             \
    diag_last_ = now_;                                                    \
  }
#else
#define MOE_DIAG_T(i)
#define MOE_DIAG_A(i)
#endif

// The diagonal-block work as device functions (shared by chol_diag_lds_kernel and the look-ahead of chol_step_kernel).
// diag_load: the nb x nb block at (k0, k0) into S (lower triangle; identity beyond nb).
__device__ __forceinline__ void diag_load(const double* __restrict__ A, long lda, int k0, int nb, double (*S)[NB + 1]) {
  const int t = threadIdx.x;
  // (unconditional loads from clamped addresses, all issued before the first use: one memory round trip instead of one per
  //  guarded element -- 18 k of the kernel's 128 k cycles were this loop)
  {
    constexpr int PER = (NB * (NB + 1) + 255) / 256;
    double v[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int idx = min(t + q * 256, NB * (NB + 1) - 1);
      const int i = idx % NB, j = idx / NB;  // j runs to NB: column 64 belongs to the packed inverse
      v[q] = A[(long)(k0 + min(i, nb - 1)) + (long)(k0 + min(j, nb - 1)) * lda];
    }
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int idx = t + q * 256;
      if (idx < NB * (NB + 1)) {
        const int i = idx % NB, j = idx / NB;
        double w = (i == j && i >= nb) ? 1.0 : 0.0;  // rows / columns beyond nb (last, partial block): identity
        if (i < nb && j < nb && j <= i) w = v[q];
        S[i][j] = w;
      }
    }
  }
}

// diag_factor (r3 form): S <- its Cholesky factor (lower) and the inverses of the four 16 x 16 diagonal sub-blocks packed above
// the diagonal; *s_bad = first failing pivot (global index + 1) or 0.  Ends with a barrier.  In 16-column sub-steps:
//   (1) the first wavefront, LANE = ROW of the 64 x 64 block, holds its 16 entries of the column block in registers and runs the
//       reference's unblocked outer-product elimination over them (v_readlane pivot, division by its square root, the 1e-16
//       pivot rule): the rows BELOW the 16 x 16 diagonal sub-block ride along in the same instructions, so the panel
//       A_panel L16^-T comes out of the elimination itself -- the inverse of the sub-block is not on the chain any more
//       (round 2 factored and inverted 16 rows, then formed the panel through that inverse: 4.75 us per sub-step, one
//       wavefront working, three waiting);
//   (2) the trailing 16 x 16 blocks get their rank-16 update on the matrix pipe (v_mfma_f64_16x16x4_f64, one block per
//       wavefront and round);
// then (3) the four 16 x 16 inverses, one per wavefront, column per lane, forward substitution against broadcast reads.
__device__ __forceinline__ void diag_factor(double (*S)[NB + 1], double (*W)[2 * SB + 1], int* s_bad_p, int k0) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int& s_bad = *s_bad_p;
  double* Rd = &W[32][0];  // 1 / L[k][k], k < 64 (rows 32 .. 33 of the scratch; rows 0 .. 31 hold diag_invert's product block)
  volatile __attribute__((address_space(3))) double* Lc =
      (volatile __attribute__((address_space(3))) double*)&W[36][0];  // the current pivot column, 64 entries (rows 36 .. 37)
#if defined(MOE_DIAG_PROF)
  unsigned long long diag_last_ = 0;
#endif
#pragma unroll 1
  for (int s0 = 0; s0 < NB; s0 += SB) {
    MOE_DIAG_A(0);
    if (wave == 0) {
      const int ri = lane;
      double a[SB];
#pragma unroll
      for (int c = 0; c < SB; ++c) a[c] = S[ri][s0 + c];  // (rows above s0 and entries above the diagonal: loaded, never used)
      int bad = 0;
      // rank-1 updates are applied ONE PIVOT LATE, except to the next pivot's own column: a[j] -= L[i][k] L[s0 + j][k] needs the
      // second factor from lane s0 + j.  For j = k + 1 it is fetched with v_readlane and applied at once (it feeds the next pivot:
      // the critical path); for j >= k + 2 every lane writes its L[i][k] to an LDS column, the factors come back as broadcast
      // reads (off the vector ALU: with two v_readlane per factor and the scalar-register spills they caused, the loop was ~90 VALU
      // instructions per pivot on ONE wavefront, issue-bound), and the fmas that consume them sit in the NEXT pivot's iteration,
      // in the shadow of its square-root chain -- applied in place they either wait for the LDS round trip or, sunk by the
      // scheduler to their first use, pile up to k dependent fmas in front of pivot k.
      double p_lik = 0.0, p_ljk[SB];
#pragma unroll
      for (int j = 0; j < SB; ++j) p_ljk[j] = 0.0;
#pragma unroll
      for (int k = 0; k < SB; ++k) {
        const double piv = readlane_f64(a[k], s0 + k);
        if (bad == 0 && !(piv > 1.0e-16)) bad = k0 + s0 + k + 1;  // gpp_linear_algebra.cpp:118
        const double ps = piv;  // (a failed pivot is reported above; what follows it -- NaN, inf -- is never written back)
        // sqrt and reciprocal sqrt TOGETHER (the coupled iteration of fastmath.hpp's sqrt_pos): g -> sqrt(ps), h -> 1 / (2 sqrt(ps)),
        // one step on both from the v_rsq_f64 seed (2^-22 -> 2^-44), then the Heron correction makes lkk the correctly rounded
        // square root.  r = 2 h is good to 2^-44, which is all the divisions below need: each is followed by its residual
        // correction against lkk.
        const double y = __builtin_amdgcn_rsq(ps);
        // (the previous pivot's deferred updates: independent of the chain above / below)
#pragma unroll
        for (int j = k + 1; j < SB; ++j) a[j] = a[j] - p_lik * p_ljk[j];
        double g = ps * y, h = 0.5 * y;
        const double e = fma(-h, g, 0.5);
        g = fma(g, e, g);
        h = fma(h, e, h);
        const double lkk = fma(fma(-g, g, ps), h, g);
        const double r = h + h;
        if (lane == 0) Rd[s0 + k] = r;
        double q = a[k] * r;
        q = fma(fma(-q, lkk, a[k]), r, q);  // residual correction of a / lkk
        const double lik = (ri == s0 + k) ? lkk : q;
        a[k] = lik;
        Lc[ri] = lik;
        if (k + 1 < SB) {
          const double ljk = readlane_f64(q, s0 + k + 1);  // (lane s0 + k + 1 is below the diagonal: its lik IS its q)
          a[k + 1 < SB ? k + 1 : k] = a[k + 1 < SB ? k + 1 : k] - lik * ljk;
        }
        p_lik = lik;
#pragma unroll
        for (int j = 0; j < SB; ++j) p_ljk[j] = (j >= k + 2) ? Lc[s0 + (j >= k + 2 ? j : 0)] : 0.0;  // (this pivot's column, in order)
      }
      if (ri >= s0) {
#pragma unroll
        for (int c = 0; c < SB; ++c)
          if (ri >= s0 + SB || c <= ri - s0) S[ri][s0 + c] = a[c];  // (the diagonal sub-block: lower triangle only)
      }
      if (lane == 0 && bad != 0 && s_bad == 0) s_bad = bad;
    }
    __syncthreads();
    MOE_DIAG_A(1);
    // (2) S[i][j] -= sum_c P[i][c] P[j][c] over the 16 x 16 blocks below / right of the sub-block, lower block triangle
    const int nbk = (NB - s0 - SB) / SB;
    int bidx = 0;
    for (int ib = 0; ib < nbk; ++ib)
      for (int jb = 0; jb <= ib; ++jb, ++bidx) {
        if ((bidx & 3) != wave) continue;
        const int r0i = s0 + SB + ib * SB, r0j = s0 + SB + jb * SB;
        f64x4 acc = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < SB; kk += 4) {
          const double av = S[r0i + (lane & 15)][s0 + kk + (lane >> 4)];
          const double bv = S[r0j + (lane & 15)][s0 + kk + (lane >> 4)];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = 4 * r + (lane >> 4), n = lane & 15;
          if (ib != jb || n <= m) S[r0i + m][r0j + n] -= acc[r];
        }
      }
    __syncthreads();
    MOE_DIAG_A(2);
  }
  // (3) the inverses of the four 16 x 16 diagonal sub-blocks, one per wavefront: lane & 15 = column i of the inverse, forward
  // substitution L16 x = e_i in column order (once x_r is final every later row folds it in: independent fmas, the same
  // left-to-right sums as a row-by-row solve); entries above i are exact zeros
  {
    const int b0 = wave * SB, i = lane & (SB - 1);
    double sum[SB], x[SB];
#pragma unroll
    for (int r = 0; r < SB; ++r) sum[r] = (r == i) ? 1.0 : 0.0;
#pragma unroll
    for (int r = 0; r < SB; ++r) {
      const double lrr = S[b0 + r][b0 + r], rd = Rd[b0 + r];
      double q = sum[r] * rd;
      q = fma(fma(-q, lrr, sum[r]), rd, q);
      x[r] = q;
#pragma unroll
      for (int r2 = r + 1; r2 < SB; ++r2) sum[r2] = fma(-S[b0 + r2][b0 + r], q, sum[r2]);
      __builtin_amdgcn_sched_barrier(0);  // (keeps the scheduler from hoisting all 120 broadcast reads to the top: 240 VGPRs)
    }
    if (lane < SB) {
#pragma unroll
      for (int c = 0; c < SB; ++c)
        if (c >= i) MOE_XS(b0 + c, b0 + i) = x[c];  // x[c] = (L16^-1)[c][i]
    }
  }
  __syncthreads();
  MOE_DIAG_A(3);
}

// diag_invert: the off-diagonal blocks of the 64 x 64 inverse by recursive halving on the matrix pipe,
//     inv [[L11, 0], [L21, L22]] = [[X11, 0], [-X22 (L21 X11), X22]],
// 16 x 16 nodes (0,16,32) and (32,48,64) on wavefronts 0 and 1 -- the product L21 X11 stays in the MFMA result registers, whose
// layout IS the B-operand layout of the second product -- then the 32 x 32 node (0,32,64) on all four wavefronts (its product
// block goes through the scratch W).  The packed layout of MOE_XS; triangular operands are masked on the fly.
__device__ __forceinline__ void diag_invert(double (*S)[NB + 1], double (*W)[2 * SB + 1]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l16 = lane & 15, l4 = lane >> 4;
  if (wave < 2) {
    const int LO = wave * 32, MID = LO + SB;
    f64x4 w = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < SB; kk += 4) {  // W = L21 X11
      const int k = kk + l4;
      const double av = S[MID + l16][LO + k];                          // L21[m = l16][k]
      const double xv = MOE_XS(LO + max(k, l16), LO + l16);            // X11[k][n = l16], lower triangular
      w = __builtin_amdgcn_mfma_f64_16x16x4f64(av, (k >= l16) ? xv : 0.0, w, 0, 0, 0);
    }
    f64x4 x = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 4; ++r) {  // X21 = -X22 W: register r of W is the B operand of k-step 4 r
      const int k = 4 * r + l4;
      const double xv = MOE_XS(MID + max(l16, k), MID + k);           // X22[m = l16][k], lower triangular
      x = __builtin_amdgcn_mfma_f64_16x16x4f64((l16 >= k) ? xv : 0.0, w[r], x, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) MOE_XS(MID + 4 * r + l4, LO + l16) = -x[r];
  }
  __syncthreads();
  {
    const int mb = wave >> 1, nb = wave & 1;  // 16 x 16 output block of the 32 x 32 node
    f64x4 w = f64x4{0.0, 0.0, 0.0, 0.0};
    for (int kb = nb; kb < 2; ++kb)  // W[mb][nb] = sum_kb L21[mb][kb] X11[kb][nb]   (X11 lower: kb >= nb)
#pragma unroll
      for (int kk = 0; kk < SB; kk += 4) {
        const int k = kb * SB + kk + l4, n = nb * SB + l16;
        const double av = S[32 + mb * SB + l16][k];
        const double xv = MOE_XS(max(k, n), n);
        w = __builtin_amdgcn_mfma_f64_16x16x4f64(av, (k >= n) ? xv : 0.0, w, 0, 0, 0);
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) W[mb * SB + 4 * r + l4][nb * SB + l16] = w[r];
    __syncthreads();
    f64x4 x = f64x4{0.0, 0.0, 0.0, 0.0};
    for (int kb = 0; kb <= mb; ++kb)  // X21[mb][nb] = -sum_kb X22[mb][kb] W[kb][nb]   (X22 lower: kb <= mb)
#pragma unroll
      for (int kk = 0; kk < SB; kk += 4) {
        const int m = mb * SB + l16, k = kb * SB + kk + l4;
        const double xv = MOE_XS(32 + max(m, k), 32 + k);
        x = __builtin_amdgcn_mfma_f64_16x16x4f64((m >= k) ? xv : 0.0, W[k][nb * SB + l16], x, 0, 0, 0);
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) MOE_XS(32 + mb * SB + 4 * r + l4, nb * SB + l16) = -x[r];
  }
  __syncthreads();
}

// diag_store: L (strict upper written as 0) into A's block, L^-1 into Linv's block.
__device__ __forceinline__ void diag_store(double* __restrict__ A, long lda, double* __restrict__ Linv, long ldl, int k0, int nb,
                                           double (*S)[NB + 1]) {
  const int t = threadIdx.x;
  for (int idx = t; idx < NB * NB; idx += 256) {
    const int i = idx % NB, j = idx / NB;
    if (i < nb && j < nb) {
      A[(long)(k0 + i) + (long)(k0 + j) * lda] = (j <= i) ? S[i][j] : 0.0;  // strict upper written as 0
      Linv[(long)(k0 + i) + (long)(k0 + j) * ldl] = (j <= i) ? MOE_XS(i, j) : 0.0;
    }
  }
}

__global__ __launch_bounds__(256) void chol_diag_lds_kernel(double* __restrict__ A, long lda, double* __restrict__ Linv,
                                                           long ldl, int k0, int nb, int* __restrict__ info) {
  __shared__ double S[NB][NB + 1];      // S[i][j] = L[i][j] for j <= i;  (L^-1)[i][j] at S[j][i + 1]
  __shared__ double W[NB][2 * SB + 1];  // scratch: panel / product blocks (up to 32 columns)
  __shared__ int s_bad;
  if (*info != 0) return;
  const int t = threadIdx.x;
  MOE_DIAG_T(0);
  diag_load(A, lda, k0, nb, S);
  if (t == 0) s_bad = 0;
  __syncthreads();
  MOE_DIAG_T(1);
  diag_factor(S, W, &s_bad, k0);
  MOE_DIAG_T(2);
  if (s_bad != 0) {
    if (t == 0) *info = s_bad;
    return;
  }
  diag_invert(S, W);
  MOE_DIAG_T(3);
  diag_store(A, lda, Linv, ldl, k0, nb, S);
  MOE_DIAG_T(4);
}

// ---------------------------------------------------------------------------------------------------------------------
// One 64-column step of the two-level factorisation in ONE launch (r3; before: diagonal kernel -> panel kernel -> update kernel,
// three dependent launches of 50 + 17 + 23 us, 125 times at N = 8000), with the next diagonal block factored AHEAD inside it.
// Grid (cb + 1, tb): y = row tile bi (i0 = k0 + nb + 64 bi), x = 0: the panel solve of that row tile, x = 1 + j: the update of
// tile (bi, j), j < cb the column tiles this outer block still has to factor.  No workgroup waits for another one:
//   * the UNSOLVED column block of this step is read from a scratch copy `Ccur` (N x 64, rows by absolute index), never from A;
//     P_i = C_i L_kk^-T is what workgroup (0, bi) writes into A (the final L panel) -- nobody reads that part of A meanwhile;
//   * an update workgroup recomputes the two panels it needs, P_i and P_j, from Ccur and the inverted diagonal block (two extra
//     64^3 products on the matrix pipe, ~1 us each) instead of waiting for their owners, then A_ij -= P_i P_j^T;
//   * tiles of the NEXT column block (j = 0) are written to the other scratch buffer `Cnext` -- they are the next step's Ccur;
//   * LOOK-AHEAD: workgroup (1, 0) holds the updated tile (k+1, k+1) and factors and inverts it on the spot (diag_factor /
//     diag_invert), so the diagonal-block chain -- 47 us a link, the critical path of the whole factorisation -- no longer
//     waits for the panel and update launches of its own step: the next launch finds its diagonal block done.
// All 64 x 64 x 64 products run on the matrix pipe (v_mfma_f64_16x16x4_f64, a 32 x 32 quadrant per wavefront).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void mfma_64(const double (*As)[NB + 1], const double (*Bs)[NB + 1], f64x4 (&acc)[2][2], int wi, int wj,
                                        int lk, int lx) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
  for (int k4 = 0; k4 < NB; k4 += 4) {
    double fa[2], fb[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) fa[a] = As[k4 + lk][wi + 16 * a + lx];
#pragma unroll
    for (int b = 0; b < 2; ++b) fb[b] = Bs[k4 + lk][wj + 16 * b + lx];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[b], fa[a], acc[a][b], 0, 0, 0);
  }
}

// the unsolved column block k0 (rows below its diagonal block) into the scratch buffer: the start of an outer block
__global__ __launch_bounds__(256) void chol_colcopy_kernel(const double* __restrict__ A, long lda, int N, int k0, int nb,
                                                          double* __restrict__ C, long ldc) {
  // grid (row chunks, nb columns): one element per thread (a thread walking its row's 64 columns was 64 dependent strided
  // round trips: 20 us for 4 MB)
  const long r = (long)k0 + nb + (long)blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  if (r < N) C[r + (long)c * ldc] = A[r + (long)(k0 + c) * lda];
}

__global__ __launch_bounds__(256) void chol_step_kernel(double* __restrict__ A, long lda, double* __restrict__ Linv, long ldl, int N,
                                                       int k0, int nb, int cb, int* __restrict__ info,
                                                       const double* __restrict__ Ccur, double* __restrict__ Cnext, long ldc,
                                                       int lookahead) {
  // two 64 x 65 LDS buffers (66.5 KB: two workgroups per CU): B0 = Ds, then Pt_i, then the diagonal-block layout S;
  // B1 = C_i^T, then C_j^T, then Pt_j, then the diagonal block's scratch W
  extern __shared__ __attribute__((aligned(16))) double step_smem[];
  double (*B0)[NB + 1] = reinterpret_cast<double (*)[NB + 1]>(step_smem);
  double (*B1)[NB + 1] = reinterpret_cast<double (*)[NB + 1]>(step_smem + NB * (NB + 1));
  __shared__ int s_bad;
  if (*info != 0) return;
  const int bi = blockIdx.y, jc = blockIdx.x;
  const int j = jc - 1;
  if (j > bi) return;  // (upper triangle)
  const int base = k0 + nb;
  const int i0 = base + bi * NB;
  const int t = threadIdx.x;
  const int wave = t >> 6, lane = t & 63;
  const int wi = (wave & 1) * 32, wj = (wave >> 1) * 32;
  const int lk = lane >> 4, lx = lane & 15;
  const bool need_j = j >= 0 && j != bi;
  const int j0 = base + max(j, 0) * NB;
  const bool ahead = lookahead && bi == 0 && j == 0;  // tile (k+1, k+1): the next diagonal block
  constexpr int PER = NB * NB / 256;  // elements of a 64 x 64 tile per thread
  // Global loads are issued in two batches, each from clamped addresses into registers and only then staged: one memory round
  // trip per batch instead of one per staged element (guarded loads inside a staging loop cost a round trip each: 16 in a row).
  // The second batch (the other panel's rows, the tile to update) is requested before the first product and lands behind it.
  double rD[PER], rI[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int q = t + u * 256, r = q % NB, c = q / NB;  // r walks rows (contiguous in memory)
    const int rc = min(r, nb - 1), cc = min(c, nb - 1);
    rD[u] = Linv[(long)(k0 + rc) + (long)(k0 + cc) * ldl];
    rI[u] = Ccur[(long)min(i0 + r, N - 1) + (long)cc * ldc];
  }
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int q = t + u * 256, r = q % NB, c = q / NB;
    B0[c][r] = (r < nb && c < nb) ? rD[u] : 0.0;          // Ds[c][r] = Linv_kk[r][c]
    B1[c][r] = (i0 + r < N && c < nb) ? rI[u] : 0.0;      // C_i^T
  }
  double rJ[PER], rC[PER];
  if (need_j) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int q = t + u * 256, r = q % NB, c = q / NB;
      rJ[u] = Ccur[(long)min(j0 + r, N - 1) + (long)min(c, nb - 1) * ldc];
    }
  }
  if (jc > 0) {  // the tile this workgroup updates, in the result layout of the products below
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gi = i0 + wi + 16 * a + lx, gj = j0 + wj + 16 * b + lk + 4 * r;
          rC[(a * 2 + b) * 4 + r] = A[(long)min(gi, N - 1) + (long)min(gj, N - 1) * lda];
        }
  }
  __syncthreads();
  f64x4 acc[2][2], accj[2][2];
  mfma_64(B1, B0, acc, wi, wj, lk, lx);  // P_i[r][c] = sum_j C_i[r][j] Linv[c][j]
  if (jc == 0) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rr = wi + 16 * a + lx, cc = wj + 16 * b + lk + 4 * r;
          if (i0 + rr < N && cc < nb) A[(long)(i0 + rr) + (long)(k0 + cc) * lda] = acc[a][b][r];  // the final L panel
        }
    return;
  }
  if (need_j) {
    __syncthreads();  // everybody is done reading C_i^T
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int q = t + u * 256, r = q % NB, c = q / NB;
      B1[c][r] = (j0 + r < N && c < nb) ? rJ[u] : 0.0;  // C_j^T
    }
    __syncthreads();
    mfma_64(B1, B0, accj, wi, wj, lk, lx);
  }
  __syncthreads();  // everybody is done with Ds and C^T
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rr = wi + 16 * a + lx, cc = wj + 16 * b + lk + 4 * r;
        B0[cc][rr] = acc[a][b][r];               // Pt_i[k][r]
        if (need_j) B1[cc][rr] = accj[a][b][r];  // Pt_j[k][c]
      }
  __syncthreads();
  mfma_64(B0, need_j ? B1 : B0, acc, wi, wj, lk, lx);  // U[r][c] = sum_k P_i[r][k] P_j[c][k]
  if (!ahead) {
    // the next column block (j == 0) goes to the scratch buffer the next step reads, everything else is updated in place
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gi = i0 + wi + 16 * a + lx, cj = wj + 16 * b + lk + 4 * r, gj = j0 + cj;
          if (gi < N && gj < N && gj <= gi) {
            const double v = rC[(a * 2 + b) * 4 + r] - acc[a][b][r];
            if (j == 0)
              Cnext[(long)gi + (long)cj * ldc] = v;
            else
              A[(long)gi + (long)gj * lda] = v;
          }
        }
    return;
  }
  // the updated tile (k+1, k+1) goes straight into the diagonal-block layout: S = B0, W = B1
  const int nb1 = min(NB, N - i0);
  double (*S)[NB + 1] = B0;
  double (*W)[2 * SB + 1] = reinterpret_cast<double (*)[2 * SB + 1]>(step_smem + NB * (NB + 1));
  __syncthreads();  // everybody is done reading Pt_i
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ii = wi + 16 * a + lx, jj = wj + 16 * b + lk + 4 * r;
        double w = (ii == jj && ii >= nb1) ? 1.0 : 0.0;  // rows / columns beyond nb1 (last, partial block): identity
        if (ii < nb1 && jj < nb1 && jj <= ii) w = rC[(a * 2 + b) * 4 + r] - acc[a][b][r];
        S[ii][jj] = w;
      }
  if (t < NB) S[t][NB] = 0.0;  // (column 64 belongs to the packed inverse)
  if (t == 0) s_bad = 0;
  __syncthreads();
  diag_factor(S, W, &s_bad, i0);
  if (s_bad != 0) {
    if (t == 0) *info = s_bad;
    return;
  }
  diag_invert(S, W);
  diag_store(A, lda, Linv, ldl, i0, nb1, S);
}
#undef MOE_XS

// A[i][j] -= sum_{k in [kp0, kp0 + kw)} A[i][k] A[j][k] for the trailing rows / columns i, j >= base, lower triangle only: the
// rank-kw update of a right-looking factorisation on the matrix pipe.  64 x 64 tile per workgroup, each wavefront a 32 x 32
// quadrant as 2 x 2 MFMA tiles; both operands are row panels of the same matrix (row index fastest in memory: coalesced);
// the next K tile travels global -> registers while the current one is multiplied out of LDS.
__global__ __launch_bounds__(256) void syrk_mfma_kernel(double* __restrict__ A, long lda, int N, int base, int kp0, int kw,
                                                       const int* __restrict__ info, int cj0 = 0) {
  constexpr int TM = 64, TKS = 16, LD = 65;
  constexpr int NF = TM * TKS / 256;
  __shared__ double As[TKS][LD];
  __shared__ double Bs[TKS][LD];
  if (*info != 0) return;
  // triangular grid folded into a rectangle: workgroup (x, y) of a T x ceil((T + 1) / 2) ... kept simple: skip the upper half
  const int bi = blockIdx.y, bj = blockIdx.x + cj0;  // (cj0: first column tile of this launch -- the update is issued in two parts)
  if (bj > bi) return;
  const int i0 = base + bi * TM, j0 = base + bj * TM;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wi = (wave & 1) * 32, wj = (wave >> 1) * 32;
  const int lk = lane >> 4, lx = lane & 15;
  f64x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
  double ra[NF], rb[NF];
  auto fetch = [&](int k) {
#pragma unroll
    for (int it = 0; it < NF; ++it) {
      const int t = threadIdx.x + 256 * it;
      const int ii = t % TM, kk = t / TM;
      const bool kok = k + kk < kw;
      const long col = (long)(kp0 + k + kk) * lda;
      ra[it] = (kok && i0 + ii < N) ? A[(long)(i0 + ii) + col] : 0.0;
      rb[it] = (kok && j0 + ii < N) ? A[(long)(j0 + ii) + col] : 0.0;
    }
  };
  fetch(0);
  for (int k = 0; k < kw; k += TKS) {
#pragma unroll
    for (int it = 0; it < NF; ++it) {
      const int t = threadIdx.x + 256 * it;
      As[t / TM][t % TM] = ra[it];
      Bs[t / TM][t % TM] = rb[it];
    }
    __syncthreads();
    if (k + TKS < kw) fetch(k + TKS);
#pragma unroll
    for (int k4 = 0; k4 < TKS; k4 += 4) {
      double fa[2], fb[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) fa[a] = As[k4 + lk][wi + 16 * a + lx];
#pragma unroll
      for (int b = 0; b < 2; ++b) fb[b] = Bs[k4 + lk][wj + 16 * b + lx];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[b], fa[a], acc[a][b], 0, 0, 0);
    }
    __syncthreads();
  }
  // D[x][y] = C[row = y][col = x]: lane holds y = lane & 15 (row), x = (lane >> 4) + 4 r (column).  The 16 elements are LOADED
  // first (clamped addresses), then updated and stored: written as 16 read-modify-writes the loads cannot move above the stores
  // to the same array, and each pays its own memory round trip (r3: a third of this kernel's time at K = 512)
  double cv[16];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gi = i0 + wi + 16 * a + lx, gj = j0 + wj + 16 * b + lk + 4 * r;
        cv[(a * 2 + b) * 4 + r] = A[(long)min(gi, N - 1) + (long)min(gj, N - 1) * lda];
      }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gi = i0 + wi + 16 * a + lx, gj = j0 + wj + 16 * b + lk + 4 * r;
        if (gi < N && gj < N && gj <= gi) A[(long)gi + (long)gj * lda] = cv[(a * 2 + b) * 4 + r] - acc[a][b][r];
      }
}

namespace {
// Off-diagonal blocks of L^-1 (the 64 x 64 diagonal blocks are already inverted), LEVEL BY LEVEL: at block size B = 64, 128, ...
// node k pairs the inverted diagonal blocks [2kB, (2k+1)B) and [(2k+1)B, (2k+2)B):
//     inv [[L11, 0], [L21, L22]] = [[X11, 0], [-X22 (L21 X11), X22]],
// two triangular GEMMs per node -- and ALL nodes of a level go down in one batched launch of each (grid.z = node).  The
// recursion this replaces issued them node by node: 248 launches at N = 8000, most of them 64 .. 256-wide products that are
// pure launch latency (~2.5 of the inverse's 7 ms).  r3.
// r5: levels B_lo <= B < B_hi only; phase 1 = only the first product of each (L21 X11 -> work), 2 = only the second (X21 = -X22 work),
// 0 = both -- the early-inverse schedule of launch_cholesky_and_inverse runs a level's two products at different times.
void trtri_levels(const double* L, long lda, double* Linv, long ldl, int N, double* work, hipStream_t s, long B_lo = NB,
                  long B_hi = (long)1 << 40, int phase = 0) {
  for (long B = B_lo; B < N && B < B_hi; B *= 2) {
    const int nn = (int)((N - B + 2 * B - 1) / (2 * B));  // nodes with a non-empty lower block: (2k + 1) B < N
    const int rows_max = (int)std::min<long>(B, N - B);
    const long ldw = rows_max;
    const int rt = (rows_max + 63) / 64, ct = (int)((B + 63) / 64);
    int xmul = 1;
    for (int cand : {37, 41, 43, 47, 53, 59})
      if (rt % cand != 0) {
        xmul = cand;
        break;
      }
    // cost estimates of the level's two products (the nodes' tiles together: they are dispatched longest first across the batch)
    Gemm128Estimate e1, e2;
    for (int z = 0; z < nn; ++z) {
      const long mz = std::min<long>(rows_max, (N - B) - (long)z * 2 * B);
      const double rtiles = (double)((mz + 127) / 128);
      for (long j0 = 0; j0 < B; j0 += 128) e1.add_tile((double)(B - j0), rtiles);                       // L21 X11: k >= j0
      for (long i0 = 0; i0 < mz; i0 += 128) e2.add_tile((double)std::min<long>(mz, i0 + 128), (double)((B + 127) / 128));  // X22 W: k < i0 + 128
    }
    Gemm128Estimate both = e1;
    both.total_units += e2.total_units;
    both.flops += e2.flops;
    both.longest = std::max(e1.longest, e2.longest) * 2.0;  // (two launches: their busiest CUs add up)
    if (B >= 128 && use_gemm128(both)) {
      // r4: both products through the 128-tile kernel (gemm128.hpp), the level's nodes as its batch
      g128::GemmArgs g1{};
      g1.A = g128::Operand{L + B, lda, rows_max, (int)B, 1};            // L21: rows contiguous
      g1.B = g128::Operand{Linv, ldl, (int)B, (int)B, 1};               // X11: lower triangular, K contiguous
      g1.C = work;
      g1.ldc = ldw;
      g1.M = rows_max;
      g1.N = (int)B;
      g1.K = (int)B;
      g1.sA = 2 * B * (1 + lda);
      g1.sB = 2 * B * (1 + ldl);
      g1.sC = ldw * B;
      g1.m_total = (int)(N - B);
      g1.m_step = (int)(2 * B);
      if (phase != 2) launch_gemm128<false, 0, true, 2, false>(g1, nn, s);
      g128::GemmArgs g2{};
      g2.A = g128::Operand{Linv + B + B * ldl, ldl, rows_max, rows_max, 1};  // X22: lower triangular, rows contiguous
      g2.B = g128::Operand{work, ldw, (int)B, rows_max, 1};                  // L21 X11: K contiguous
      g2.C = Linv + B;
      g2.ldc = ldl;
      g2.M = rows_max;
      g2.N = (int)B;
      g2.K = rows_max;
      g2.sA = 2 * B * (1 + ldl);
      g2.sB = ldw * B;
      g2.sC = 2 * B * (1 + ldl);
      g2.m_total = (int)(N - B);
      g2.m_step = (int)(2 * B);
      if (phase != 1) launch_gemm128<false, 1, true, 0, true>(g2, nn, s);
      continue;
    }
    // work_k (rows x B) = L21 X11   (X11 lower triangular: MODE 3)
    const int pairc = (ct >= 16) ? 2 : 0;  // (column pairing: see mfma_gemm_kernel)
    if (phase != 2)
    MOE_LAUNCH((mfma_gemm_kernel<3, false, 16>), dim3(pairc ? (ct + 1) / 2 : ct, rt, nn), dim3(256), 0, s, rows_max, (int)B,
                       (int)B, L + B, lda, (const double*)Linv, ldl, work, ldw, xmul, pairc, 2 * B * (1 + lda), 2 * B * (1 + ldl),
                       ldw * B, (int)(N - B), (int)(2 * B));
    // X21 = -X22 work_k   (X22 lower triangular: MODE 1, negated).  A single big node (the top levels) pairs row tile p with its
    // mirror so that every workgroup walks the same number of K steps (see mfma_gemm_kernel).
    const int pair = (nn == 1 && rt >= 16) ? 1 : 0;
    if (phase != 1)
    MOE_LAUNCH((mfma_gemm_kernel<1, true, 16>), dim3(ct, pair ? (rt + 1) / 2 : rt, nn), dim3(256), 0, s, rows_max, (int)B,
                       rows_max, (const double*)(Linv + B + B * ldl), ldl, (const double*)work, ldw, Linv + B, ldl, xmul, pair,
                       2 * B * (1 + ldl), ldw * B, 2 * B * (1 + ldl), (int)(N - B), (int)(2 * B));
  }
}
}  // namespace

namespace {
size_t trtri_level_doubles(long N, long B_hi) {  // the largest level below B_hi of trtri_levels: nodes x (rows x B)
  size_t need = 1;
  for (long B = NB; B < N && B < B_hi; B *= 2) {
    const long nn = (N - B + 2 * B - 1) / (2 * B), rows_max = std::min<long>(B, N - B);
    need = std::max(need, (size_t)(nn * rows_max * B));
  }
  return need;
}
}  // namespace

// Early-inverse split (r5): H = the top level's block size (the largest 64 * 2^k below N) when the schedule applies -- H on an outer-block
// boundary, and a trailing part worth a level of its own -- else 0.
long early_inverse_split(int N) {
  if (N < 2048) return 0;
  long H = NB;
  while (H * 2 < N) H *= 2;
  if (H % kOuter != 0 || N - H < 4 * NB) return 0;
  return H;
}

size_t cholesky_work_doubles(int N) {
  // (never less than the step kernel's column-block buffers, which the inversion's workspace lends the factorisation: no
  //  stream-ordered allocation inside a build -- r5)
  size_t need = std::max(trtri_level_doubles(N, (long)1 << 40), chol_scratch_doubles(N));
  // the early-inverse schedule keeps four regions alive at once: the step kernel's column-block buffers | the levels below the top of
  // the leading part | the same of the trailing part | the top level's L21 X11
  const long H = early_inverse_split(N);
  if (H > 0)
    need = std::max(need, chol_scratch_doubles(N) + trtri_level_doubles(H, H) + trtri_level_doubles(N - H, H) + (size_t)(N - H) * H);
  return need;
}

// The factorisation part of the two-level algorithm (L in place, the 64 x 64 diagonal blocks of L^-1 in Linv's diagonal blocks).
// Default schedule (r3): per 64-column step ONE launch of chol_step_kernel -- panel solve, update of the outer block's remaining
// columns, and the NEXT diagonal block factored ahead by the workgroup that owns it; a stand-alone diagonal kernel (and a copy of
// the column block into the step kernel's scratch buffer) only at the start of each outer block, whose first tiles are final only
// after the rank-512 update.  MOE_CHOL_FUSED_STEP=0: the round-2 schedule, three dependent launches per step (diagonal, panel,
// update) -- kept for A/B runs and as the second implementation the tests compare against.  (Round 2 also tried a look-ahead over
// two HIP streams with events: slower than in-order -- three cross-queue dependences and two more launches per step cost more
// than the overlap; the in-kernel look-ahead replaces it.  A first fused kernel with one workgroup per ROW tile and release /
// acquire flags between workgroups was no faster than three launches at N = 8000 -- each workgroup walked eight dependent
// load-multiply-store rounds -- and 2.7x slower at N = 26 000, where its 406 workgroups of 100 KB LDS are not co-resident.)
// `scratch`: chol_scratch_doubles(N) doubles for the step kernel's two column-block buffers.
size_t chol_scratch_doubles(int N) { return (size_t)2 * (((size_t)N + 15) / 16 * 16) * NB; }

void cholesky_factor_two_level(int N, double* A, long lda, double* Linv, long ldl, int* info, hipStream_t s, double* scratch,
                               long hook_rows = 0, const std::function<void()>* hook = nullptr) {
  const char* fs_env = std::getenv("MOE_CHOL_FUSED_STEP");  // (read per call: the tests compare the two schedules)
  // beyond ~16 k rows the step kernel's recomputed panels cost more than the look-ahead buys (N = 26 000: 330 vs 299 ms -- the
  // rank-512 updates dominate there and the diagonal chain hides behind nothing anyway): the three-launch schedule
  const bool fused = (fs_env && *fs_env) ? (*fs_env != '0') : (N <= 16384);
  constexpr size_t kStepSmem = sizeof(double) * 2 * NB * (NB + 1);
  double* cbuf = nullptr;
  const long ldc = ((long)N + 15) / 16 * 16;
  if (fused) {
    // (on every factorisation: the opt-in is per device where the runtime enforces it, and one process may build GPs on several
    //  devices -- moe_kg_batch_multi, bench.py's in-process fallback)
    MOE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(chol_step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)kStepSmem));
    // (r5: always the caller's -- cholesky_work_doubles covers it.  Until then a missing scratch came from hipMallocAsync, and with
    //  recycled streams and device blocks (DevicePool) one test order made the N = 262 build of tools/fuzz_parity.py report a singular
    //  second diagonal block, reproducibly, until the pool was trimmed or poisoned: the stream-ordered allocation was the only
    //  ingredient whose removal cured it.)
    cbuf = scratch;
    if (cbuf == nullptr) throw Error(MOE_ERR_RUNTIME, "cholesky_factor_two_level: scratch missing (chol_scratch_doubles(N) doubles)");
  }
  // (Measured and not kept in this round, both bit-identical to the one-stream order: the far columns of the rank-512 update on a
  //  second, low-priority stream next to the inner steps -- 19.45 -> 19.2 ms at N = 8000, within the noise: the bulk update's
  //  workgroups take the CUs the diagonal chain needs; and the next outer block's first diagonal kernel + column copy on a second
  //  stream next to the far columns of the update -- 15.58 vs 15.77 ms: the 320-VGPR diagonal workgroup finds no CU to start on
  //  until the update drains.  profiles/r03_c2_chol_overlap.txt, r03_k_chol_time.txt.)
  auto diag_and_copy = [&](int k0, hipStream_t st, int buf) {  // the start of an outer block
    const int nb = std::min(NB, N - k0), below = N - k0 - nb;
    MOE_LAUNCH(chol_diag_lds_kernel, dim3(1), dim3(256), 0, st, A, lda, Linv, ldl, k0, nb, info);
    if (fused && below > 0)
      MOE_LAUNCH(chol_colcopy_kernel, dim3((below + 255) / 256, nb), dim3(256), 0, st, (const double*)A, lda, N, k0, nb,
                         cbuf + (size_t)buf * ldc * NB, ldc);
  };
  int cur = 0;
  for (int ko = 0; ko < N; ko += kOuter) {
    const int wo = std::min(kOuter, N - ko);
    bool ahead_done = false;  // the diagonal block and the scratch copy of this step's column block come from the previous step
    for (int k0 = ko; k0 < ko + wo; k0 += NB) {
      const int nb = std::min(NB, N - k0);
      const int below = N - k0 - nb;
      if (!ahead_done) diag_and_copy(k0, s, cur);
      ahead_done = false;
      if (below <= 0) continue;
      const int tb = (below + NB - 1) / NB;
      const int left = ko + wo - (k0 + nb);  // columns of this outer block still to be factored
      const int cb = (left + NB - 1) / NB;
      if (!fused) {
        MOE_LAUNCH(chol_panel_kernel, dim3(tb), dim3(256), 0, s, A, lda, Linv, ldl, N, k0, nb, info);
        if (left > 0) MOE_LAUNCH(chol_update_kernel, dim3(tb, cb), dim3(256), 0, s, A, lda, N, k0, nb, info);
        continue;
      }
      MOE_LAUNCH(chol_step_kernel, dim3(cb + 1, tb), dim3(256), kStepSmem, s, A, lda, Linv, ldl, N, k0, nb, cb, info,
                         (const double*)(cbuf + (size_t)cur * ldc * NB), cbuf + (size_t)(1 - cur) * ldc * NB, ldc, left > 0 ? 1 : 0);
      if (left > 0) {
        cur = 1 - cur;
        ahead_done = true;
      }
    }
    // (columns below hook_rows are final from here on -- this outer block's rank-512 update touches the trailing matrix only)
    if (hook != nullptr && ko + wo == hook_rows) (*hook)();
    const int trailing = N - (ko + wo);
    if (trailing > 0) {
      const int t128 = (trailing + 127) / 128;
      if (use_syrk128(trailing, wo)) {
        MOE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(g128::syrk128_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)g128::kSmemBytes));
        MOE_LAUNCH(g128::syrk128_kernel, dim3(t128 * (t128 + 1) / 2), dim3(256), g128::kSmemBytes, s, A, lda, N, ko + wo, ko,
                           wo, (const int*)info);
      } else {
        const int tt = (trailing + 63) / 64;
        MOE_LAUNCH(syrk_mfma_kernel, dim3(tt, tt), dim3(256), 0, s, A, lda, N, ko + wo, ko, wo, (const int*)info, 0);
      }
    }
  }
  MOE_HIP_CHECK(hipGetLastError());
}

void launch_cholesky_and_inverse(int N, double* A, long lda, double* Linv, long ldl, double* work, int* info,
                                 hipStream_t s, bool upper_is_zero, hipStream_t side) {
  MOE_HIP_CHECK(hipMemsetAsync(info, 0, sizeof(int), s));
  MOE_HIP_CHECK(hipMemsetAsync(Linv, 0, sizeof(double) * (size_t)ldl * N, s));
  const int nblk = (N + NB - 1) / NB;
  // r5, early inverse: once the leading H columns are final -- the top level's block size, on an outer-block boundary -- the inverse of
  // the leading H x H triangle (every level below the top) and the top level's first product L21 X11 run on `side` while `s` factors
  // the trailing part, whose 64-column steps leave most of the chip idle; afterwards the trailing part's own levels and the top level's
  // second product.  The same launches on the same data as the level-wise schedule (a level's batch is split by halves), so the
  // result agrees with it to rounding (a level's kernel is picked from its batch).  MOE_CHOL_EARLY_INVERSE=0: off.
  long H = 0;
  bool early_top = true;
  {
    const char* ei = std::getenv("MOE_CHOL_EARLY_INVERSE");
    if (side != nullptr && side != s && work != nullptr && !(ei && *ei == '0')) H = early_inverse_split(N);
    early_top = !(ei && *ei == '1');  // (MOE_CHOL_EARLY_INVERSE=1: the leading half's levels only; A/B)
  }
  double* work_lead = work;
  double* work_trail = work;
  double* work_top = work;
  // (the two events of the early-inverse schedule: destroyed on every way out; when an exception unwinds through here the side stream
  //  may hold queued work nobody waits for any more -- it is drained before the buffers it touches can be released)
  struct SideGuard {
    hipEvent_t cols = nullptr, side_ev = nullptr;
    hipStream_t side_stream = nullptr;
    int exceptions = std::uncaught_exceptions();
    ~SideGuard() {
      if (side_stream != nullptr && std::uncaught_exceptions() > exceptions) (void)hipStreamSynchronize(side_stream);
      if (cols) (void)hipEventDestroy(cols);
      if (side_ev) (void)hipEventDestroy(side_ev);
    }
  } guard;
  hipEvent_t& ev_cols = guard.cols;
  hipEvent_t& ev_side = guard.side_ev;
  std::function<void()> hook;
  if (H > 0) {
    work_lead = work + chol_scratch_doubles(N);
    work_trail = work_lead + trtri_level_doubles(H, H);
    work_top = work_trail + trtri_level_doubles(N - H, H);
    MOE_HIP_CHECK(hipEventCreateWithFlags(&ev_cols, hipEventDisableTiming));
    MOE_HIP_CHECK(hipEventCreateWithFlags(&ev_side, hipEventDisableTiming));
    guard.side_stream = side;
    hook = [&] {
      MOE_HIP_CHECK(hipEventRecord(ev_cols, s));
      MOE_HIP_CHECK(hipStreamWaitEvent(side, ev_cols, 0));
      trtri_levels(A, lda, Linv, ldl, (int)H, work_lead, side);
      if (early_top) trtri_levels(A, lda, Linv, ldl, N, work_top, side, H, 2 * H, 1);
      MOE_HIP_CHECK(hipEventRecord(ev_side, side));
    };
  }
  const char* tl_env = std::getenv("MOE_CHOL_TWO_LEVEL_MIN");  // (read per call: tests force the two-level path at small N)
  // r3: 256 (was 2048) -- with one launch per 64-column step and the 25 us diagonal block the two-level schedule also wins at
  // BO-sized training sets: factor + inverse at N = 1000 0.9 instead of 1.7 ms (one-level: three launches per step and a
  // 46 us single-wavefront diagonal kernel)
  const int two_level_min = (tl_env && *tl_env) ? std::atoi(tl_env) : 256;
  if (N >= two_level_min) {
    // (the inversion's workspace is idle during the factorisation: it lends the step kernel its column-block buffers)
    cholesky_factor_two_level(N, A, lda, Linv, ldl, info, s,
                              (work != nullptr && cholesky_work_doubles(N) >= chol_scratch_doubles(N)) ? work : nullptr, H,
                              H > 0 ? &hook : nullptr);
  } else {
    H = 0;
    for (int b = 0; b < nblk; ++b) {
      const int k0 = b * NB, nb = std::min(NB, N - k0);
      MOE_LAUNCH(chol_diag_kernel, dim3(1), dim3(64), 0, s, A, lda, Linv, ldl, k0, nb, info);
      const int below = N - k0 - nb;
      if (below > 0) {
        MOE_LAUNCH(chol_panel_kernel, dim3((below + NB - 1) / NB), dim3(256), 0, s, A, lda, Linv, ldl, N, k0, nb, info);
        const int tb = (below + NB - 1) / NB;
        MOE_LAUNCH(chol_update_kernel, dim3(tb, tb), dim3(256), 0, s, A, lda, N, k0, nb, info);
      }
    }
  }
  if (!upper_is_zero) {
    const long total = (long)N * N;
    MOE_LAUNCH(zero_strict_upper_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, A, lda, N);
  }
  // Off-diagonal blocks of L^-1 by recursive halving over the block range (diagonal blocks are already inverted):
  //   inv [[L11, 0], [L21, L22]] = [[X11, 0], [-X22 (L21 X11), X22]]
  // -- two triangular GEMMs per node, big and wide near the root, instead of one dependent block row after another.
  if (nblk > 1) {
    if (work == nullptr) throw Error(MOE_ERR_RUNTIME, "launch_cholesky_and_inverse: workspace missing");
    if (H > 0) {
      trtri_levels(A + H + H * lda, lda, Linv + H + H * ldl, ldl, (int)(N - H), work_trail, s);
      MOE_HIP_CHECK(hipStreamWaitEvent(s, ev_side, 0));
      trtri_levels(A, lda, Linv, ldl, N, work_top, s, H, 2 * H, early_top ? 2 : 0);
    } else {
      trtri_levels(A, lda, Linv, ldl, N, work, s);
    }
  }
  MOE_HIP_CHECK(hipGetLastError());
}

namespace {
__global__ __launch_bounds__(256) void sub_inplace_kernel(double* __restrict__ C, const double* __restrict__ G, int count) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < count) C[i] -= G[i];
}
// The new block row: Lrow[i + j * ldl] = V[j + i * N0];  Xrow[i + j * ldi] = -sum_{l <= i} X22[i + l * kk] W[j + l * N0]
// (column j per thread: V / W reads coalesced, kk contiguous doubles written per column).
__global__ __launch_bounds__(256) void append_rows_kernel(const double* __restrict__ V, const double* __restrict__ W,
                                                         const double* __restrict__ X22, int N0, int kk,
                                                         double* __restrict__ Lrow, long ldl, double* __restrict__ Xrow,
                                                         long ldi) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= N0) return;
  for (int i = 0; i < kk; ++i) {
    Lrow[i + (long)j * ldl] = V[j + (long)i * N0];
    double acc = 0.0;
    for (int l = 0; l <= i; ++l) acc = fma(X22[i + (long)l * kk], W[j + (long)l * N0], acc);
    Xrow[i + (long)j * ldi] = -acc;
  }
}
}  // namespace

size_t cholesky_append_work_doubles(int N0, int kk) {
  return (size_t)2 * N0 * kk + (size_t)2 * kk * kk + cholesky_work_doubles(kk);
}

void launch_cholesky_append(int N0, int kk, double* L, long ldl, double* Linv, long ldi, const double* B, double* C,
                            double* work, int* info, hipStream_t s) {
  double* V = work;                    // N0 x kk  (ld N0)   L11^-1 K12 = L21^T
  double* W = V + (size_t)N0 * kk;     // N0 x kk  (ld N0)   L11^-T V   = (L21 X11)^T
  double* G = W + (size_t)N0 * kk;     // kk x kk            V^T V
  double* X22 = G + (size_t)kk * kk;   // kk x kk            L22^-1
  double* cw = X22 + (size_t)kk * kk;  // scratch of the small factorisation
  // strict upper part of the new columns
  MOE_HIP_CHECK(hipMemset2DAsync(L + (size_t)N0 * ldl, sizeof(double) * ldl, 0, sizeof(double) * N0, kk, s));
  MOE_HIP_CHECK(hipMemset2DAsync(Linv + (size_t)N0 * ldi, sizeof(double) * ldi, 0, sizeof(double) * N0, kk, s));
  // V = L11^-1 K12;  Schur complement S = K22 - V^T V;  L22 = chol(S), X22 = L22^-1
  launch_tri_gemm_skinny('N', N0, kk, Linv, ldi, B, N0, V, N0, s);
  launch_gemm_tn(kk, kk, N0, V, N0, V, N0, G, kk, s);
  MOE_LAUNCH(sub_inplace_kernel, dim3((kk * kk + 255) / 256), dim3(256), 0, s, C, (const double*)G, kk * kk);
  launch_cholesky_and_inverse(kk, C, kk, X22, kk, cw, info, s);
  MOE_HIP_CHECK(hipMemcpy2DAsync(L + N0 + (size_t)N0 * ldl, sizeof(double) * ldl, C, sizeof(double) * kk,
                                 sizeof(double) * kk, kk, hipMemcpyDeviceToDevice, s));
  MOE_HIP_CHECK(hipMemcpy2DAsync(Linv + N0 + (size_t)N0 * ldi, sizeof(double) * ldi, X22, sizeof(double) * kk,
                                 sizeof(double) * kk, kk, hipMemcpyDeviceToDevice, s));
  // L21 = V^T;  X21 = -X22 (L21 X11) = -X22 W^T
  launch_tri_gemm_skinny('T', N0, kk, Linv, ldi, V, N0, W, N0, s);
  MOE_LAUNCH(append_rows_kernel, dim3((N0 + 255) / 256), dim3(256), 0, s, (const double*)V, (const double*)W,
                     (const double*)X22, N0, kk, L + N0, ldl, Linv + N0, ldi);
  MOE_HIP_CHECK(hipGetLastError());
}

namespace {
// Border of the log-likelihood factorisation: row N of matrix b = (y - mean)^T, corner = 1e100 (see launch_ll_batch).
__global__ __launch_bounds__(256) void ll_border_kernel(double* __restrict__ A, long lda, long a_stride, int N,
                                                        const double* __restrict__ yc) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  double* Ab = A + (long)blockIdx.y * a_stride;
  if (j < N) Ab[(long)N + (long)j * lda] = yc[j];
  if (j == N) Ab[(long)N + (long)N * lda] = 1.0e100;
}
// out[b] = (sum_i log L_ii, sum_j L[N][j]^2) of matrix b: one workgroup per matrix, fixed-order reduction.
__global__ __launch_bounds__(256) void ll_terms_batch_kernel(const double* __restrict__ A, long lda, long a_stride, int N,
                                                             double* __restrict__ out) {
  __shared__ double red[2][256];
  const double* Ab = A + (long)blockIdx.x * a_stride;
  double a = 0.0, q = 0.0;
  for (int i = threadIdx.x; i < N; i += 256) {
    a += log(Ab[(long)i + (long)i * lda]);
    const double v = Ab[(long)N + (long)i * lda];
    q = fma(v, v, q);
  }
  red[0][threadIdx.x] = a;
  red[1][threadIdx.x] = q;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
      red[0][threadIdx.x] += red[0][threadIdx.x + w];
      red[1][threadIdx.x] += red[1][threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = red[0][0];
    out[2 * blockIdx.x + 1] = red[1][0];
  }
}
}  // namespace

void launch_cholesky_batch(int N, double* A, long lda, long a_stride, double* Linv, long ldl, long l_stride, int* info,
                           int batch, hipStream_t s, double* scratch) {
  MOE_HIP_CHECK(hipMemsetAsync(info, 0, sizeof(int) * batch, s));
  {
    // large matrices (the log likelihood at C5's N = 8000): the two-level factorisation with its look-ahead schedule, one matrix
    // after the other -- a single factorisation fills the chip there, and the one-level batch kernels below are 125 steps of a
    // 150 us register-resident diagonal kernel
    // (a single BO-sized matrix too: 0.65 instead of 1.38 ms at N = 1000; batches keep the one-level kernels, whose every
    //  launch covers the whole batch -- 0.05 ms per set in calls of 64)
    const char* tl_env = std::getenv("MOE_CHOL_TWO_LEVEL_MIN");
    const int two_level_min = (tl_env && *tl_env) ? std::atoi(tl_env) : (batch == 1 ? 256 : 2048);
    if (N >= two_level_min) {
      for (int b = 0; b < batch; ++b)
        cholesky_factor_two_level(N, A + (size_t)b * a_stride, lda, Linv + (size_t)b * l_stride, ldl, info + b, s, scratch);
      return;
    }
  }
  const int nblk = (N + NB - 1) / NB;
  for (int b = 0; b < nblk; ++b) {
    const int k0 = b * NB, nb = std::min(NB, N - k0);
    MOE_LAUNCH(chol_diag_kernel, dim3(1, 1, batch), dim3(64), 0, s, A, lda, Linv, ldl, k0, nb, info, a_stride, l_stride);
    const int below = N - k0 - nb;
    if (below > 0) {
      const int tb = (below + NB - 1) / NB;
      MOE_LAUNCH(chol_panel_kernel, dim3(tb, 1, batch), dim3(256), 0, s, A, lda, (const double*)Linv, ldl, N, k0, nb,
                         (const int*)info, a_stride, l_stride);
      MOE_LAUNCH(chol_update_kernel, dim3(tb, tb, batch), dim3(256), 0, s, A, lda, N, k0, nb, (const int*)info, a_stride);
    }
  }
  MOE_HIP_CHECK(hipGetLastError());
}

void launch_ll_border(double* A, long lda, long a_stride, int N, const double* yc, int batch, hipStream_t s) {
  MOE_LAUNCH(ll_border_kernel, dim3((N + 1 + 255) / 256, batch), dim3(256), 0, s, A, lda, a_stride, N, yc);
  MOE_HIP_CHECK(hipGetLastError());
}

void launch_ll_terms_batch(const double* A, long lda, long a_stride, int N, double* out, int batch, hipStream_t s) {
  MOE_LAUNCH(ll_terms_batch_kernel, dim3(batch), dim3(256), 0, s, A, lda, a_stride, N, out);
  MOE_HIP_CHECK(hipGetLastError());
}


}  // namespace moe
