// cornell_moe_amd/csrc/kernels_cov.hip -- covariance-matrix assembly kernels for gfx950 (CDNA4, wave64).
//
// cov_build_kernel: K(A, B) with optional derivative-observation blocks (BuildMixCovarianceMatrix,
// gpp_math.cpp:309-335; BuildCovarianceMatrixWithNoiseVariance :426-455).  Output-write bound: one thread owns one
// OUTPUT ROW (so a wavefront's 64 stores of one column are 512 contiguous bytes), the B points of the tile are staged
// once in LDS and read as wave-uniform broadcasts, the A point stays in registers for the whole tile.
// Algorithmic bytes per launch: 8 * [nA*d + nB*d + nA(1+gA) * nB(1+gB)].
#include <algorithm>
#include <cstdlib>

#include "device_cov.hpp"

namespace moe {

namespace {

constexpr int kCovRows = 256;  // output rows per workgroup (4 wavefronts)
constexpr int kCovCols = 16;   // B points per workgroup

template <int DP, bool DERIVS>
struct cov_build_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const CovParams& cp, const double* __restrict__ A, int nA, const DerivList& dA, const double* __restrict__ B, int nB, const DerivList& dB, const double* __restrict__ diag_noise, double* __restrict__ out, long ld, long col0, int lower_only, int cols_per_wg, int split_at, long split_shift) {
    // (split_at / split_shift, r5, value-only builds: B points from index split_at on write their columns split_shift further right --
    //  two point sets with separate column ranges in ONE launch, launch_cov_build_pair)
    __shared__ double Bs[kCovCols][DP];
    const int gA = DERIVS ? dA.g : 0, gB = DERIVS ? dB.g : 0;
    const int rows = nA * (1 + gA);
    const int j0 = blockIdx.x * cols_per_wg;  // column tiles on grid.x (up to 2^31-1 tiles: N x M builds have M >> 65535*16)
    const int nj = min(cols_per_wg, nB - j0);
    // lower_only (K(X, X) for the factorisation, r4): tiles above the diagonal are not visited and entries above it not stored -- the
    // strict upper triangle of the destination stays what it is (zero: GpDev::rebuild clears the buffer when its shape changes)
    if (lower_only && (long)blockIdx.y * kCovRows + kCovRows - 1 < (long)j0 * (1 + gB)) return;
    for (int t = threadIdx.x; t < nj * DP; t += blockDim.x) Bs[t / DP][t % DP] = B[(long)(j0 + t / DP) * DP + (t % DP)];
    __syncthreads();
    const int r = blockIdx.y * kCovRows + threadIdx.x;
    if (r >= rows) return;
    const int i = DERIVS ? r / (1 + gA) : r;
    const int a = DERIVS ? r % (1 + gA) : 0;
    double xi[DP];
  #pragma unroll
    for (int k = 0; k < DP; ++k) xi[k] = A[(long)i * DP + k];
    for (int jj = 0; jj < nj; ++jj) {
      double diff[DP];
      double r2 = 0.0;
  #pragma unroll
      for (int k = 0; k < DP; ++k) {
        diff[k] = xi[k] - Bs[jj][k];
        r2 = fma(diff[k] * diff[k], cp.inv_l2[k], r2);
      }
      const Radial rd = radial_scalars(cp.type, cp.alpha, r2);
      if (!DERIVS) {
        double v = rd.base;
        const long col = col0 + j0 + jj;
        if (diag_noise != nullptr && (long)r == col - col0) v += diag_noise[0];
        if (!lower_only || (long)r >= col - col0) out[(long)r + (col + (j0 + jj >= split_at ? split_shift : 0)) * ld] = v;
      } else {
        for (int b = 0; b < 1 + gB; ++b) {
          double v = cov_entry<DP>(cp, rd, diff, a, b, dA, dB);
          const long colrel = (long)(j0 + jj) * (1 + gB) + b;
          if (diag_noise != nullptr && (long)r == colrel) v += diag_noise[a];
          if (!lower_only || (long)r >= colrel) out[(long)r + (col0 + colrel) * ld] = v;
        }
      }
    }
  }
};
template <int DP, bool DERIVS>
__global__ __launch_bounds__(kCovRows) void cov_build_kernel(CovParams cp, const double* __restrict__ A, int nA,
                                                            DerivList dA, const double* __restrict__ B, int nB,
                                                            DerivList dB, const double* __restrict__ diag_noise,
                                                            double* __restrict__ out, long ld, long col0, int lower_only,
                                                            int cols_per_wg, int split_at = 0x7fffffff, long split_shift = 0) {
  cov_build_kernel_body<DP, DERIVS>::run(MOE_VBLOCK, MOE_VGRID, nullptr, cp, A, nA, dA, B, nB, dB, diag_noise, out, ld, col0, lower_only, cols_per_wg, split_at, split_shift);
}

// Derivative observations on the A side (K(X, X) of a d-KG GP, K*, the N x M gradient-tail matrix): one thread per A POINT
// instead of per output row.  With g observed derivatives the 1 + g rows of a point used to be 1 + g threads that each
// recomputed the pair's distance, square root and exponential (4 x the arithmetic at C5's g = 3, 13 x at g = 12) and the kernel
// sat at 0.24 of HBM peak, FP64-issue bound.  Here the radial scalars of a pair are formed once and its (1 + gA) x (1 + gB) entries
// come from the same cov_entry as before (bit-identical values); a wavefront's 64 (1 + gA) rows of one output column are
// contiguous in memory, so they are transposed through a per-wave LDS stage and stored as full 512-byte runs.
template <int DP>
struct cov_build_points_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const CovParams& cp, const double* __restrict__ A, int nA, const DerivList& dA, const double* __restrict__ B, int nB, const DerivList& dB, const double* __restrict__ diag_noise, double* __restrict__ out, long ld, long col0, int lower_only, int cols_per_wg) {
    __shared__ double Bs[kCovCols][DP];
    __shared__ double stage_all[4][64 * (1 + kMaxDerivs)];
    const int gA = dA.g, gB = dB.g, a1 = 1 + gA;
    const int j0 = blockIdx.x * cols_per_wg;
    const int nj = min(cols_per_wg, nB - j0);
    if (lower_only && ((long)blockIdx.y * 256 + 256) * a1 - 1 < (long)j0 * (1 + gB)) return;  // (see cov_build_kernel)
    for (int t = threadIdx.x; t < nj * DP; t += blockDim.x) Bs[t / DP][t % DP] = B[(long)(j0 + t / DP) * DP + (t % DP)];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    volatile __attribute__((address_space(3))) double* stage =
        (volatile __attribute__((address_space(3))) double*)&stage_all[wave][0];
    const int p0 = blockIdx.y * 256 + wave * 64;  // first A point of this wavefront
    if (p0 >= nA) return;
    const int i = min(p0 + lane, nA - 1);  // (lanes beyond nA recompute the last point; their rows are not stored)
    const long row0 = (long)p0 * a1;       // first output row of this wavefront
    const long rows = (long)nA * a1;
    double xi[DP];
  #pragma unroll
    for (int k = 0; k < DP; ++k) xi[k] = A[(long)i * DP + k];
    for (int jj = 0; jj < nj; ++jj) {
      double diff[DP];
      double r2 = 0.0;
  #pragma unroll
      for (int k = 0; k < DP; ++k) {
        diff[k] = xi[k] - Bs[jj][k];
        r2 = fma(diff[k] * diff[k], cp.inv_l2[k], r2);
      }
      const Radial rd = radial_scalars(cp.type, cp.alpha, r2);
      for (int b = 0; b <= gB; ++b) {
        const long colrel = (long)(j0 + jj) * (1 + gB) + b;
        if (lower_only && row0 + 64 * a1 - 1 < colrel) continue;  // this wavefront's rows all lie above the diagonal in this column
        for (int a = 0; a <= gA; ++a) {
          double v = cov_entry<DP>(cp, rd, diff, a, b, dA, dB);
          if (diag_noise != nullptr && (long)i * a1 + a == colrel) v += diag_noise[a];
          stage[lane * a1 + a] = v;
        }
        // (LDS operations of one wavefront complete in order: the reads below see the writes above)
        double* dst = out + (col0 + colrel) * ld + row0;
        for (int t = 0; t <= gA; ++t) {
          const int rr = lane + 64 * t;
          const double v = stage[rr];
          if (row0 + rr < rows && (!lower_only || row0 + rr >= colrel)) dst[rr] = v;
        }
      }
    }
  }
};
template <int DP>
__global__ __launch_bounds__(256) void cov_build_points_kernel(CovParams cp, const double* __restrict__ A, int nA, DerivList dA,
                                                              const double* __restrict__ B, int nB, DerivList dB,
                                                              const double* __restrict__ diag_noise, double* __restrict__ out,
                                                              long ld, long col0, int lower_only, int cols_per_wg) {
  cov_build_points_kernel_body<DP>::run(MOE_VBLOCK, MOE_VGRID, nullptr, cp, A, nA, dA, B, nB, dB, diag_noise, out, ld, col0, lower_only, cols_per_wg);
}

// Value-only blocks (no derivative observations on either side -- K(X, X) of a q-KG GP, K*, the N x M gradient-tail matrix):
// the same mapping, with the arithmetic per entry cut from ~55 to ~40 FP64 instructions, because at ~3.4 TB/s of stores this
// kernel was FP64-issue co-limited, not store-limited (tools/covbench.hip):
//   * coordinates are centred on cp.center and divided by the length scale ONCE (the A point when it is loaded, the B points
//     when they are staged) -- sqrt(5) folded in for Matern -- so an entry needs DP subtractions + DP fmas instead of DP
//     subtractions + DP multiplications + DP fmas (centred first: scaled coordinates keep the precision of the differences);
//   * exp through the 64-entry table of fastmath.hpp (10 FP64 + 3 integer instructions, <= 1.5 ulp) instead of the degree-11
//     polynomial (17), sqrt without the clamp (the accumulation starts from 1e-300).
template <int DP>
struct cov_build_value_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const CovParams& cp, const double* __restrict__ A, int nA, const double* __restrict__ B, int nB, const double* __restrict__ diag_noise, double* __restrict__ out, long ld, long col0) {
    __shared__ double Bs[kCovCols][DP];
    __shared__ double etab[64];
    const bool matern = cp.type == MOE_COV_MATERN_NU_2P5;
    const double s5 = matern ? 2.236067977499789696409173668731276235 : 1.0;
    const int j0 = blockIdx.x * kCovCols;
    const int nj = min(kCovCols, nB - j0);
    if (threadIdx.x < 64) etab[threadIdx.x] = kExp2Tab64[threadIdx.x];
    for (int t = threadIdx.x; t < nj * DP; t += blockDim.x) {
      const int k = t % DP;
      Bs[t / DP][k] = (B[(long)(j0 + t / DP) * DP + k] - cp.center[k]) * (cp.inv_l[k] * s5);
    }
    __syncthreads();
    const int r = blockIdx.y * kCovRows + threadIdx.x;
    if (r >= nA) return;
    double xi[DP];
  #pragma unroll
    for (int k = 0; k < DP; ++k) xi[k] = (A[(long)r * DP + k] - cp.center[k]) * (cp.inv_l[k] * s5);
    const double noise = diag_noise != nullptr ? diag_noise[0] : 0.0;
    // two columns per iteration: two independent distance / sqrt / exp chains in flight per thread
    for (int jj = 0; jj < nj; jj += 2) {
      const int j1 = min(jj + 1, nj - 1);
      double ra = 1.0e-300, rb = 1.0e-300;
  #pragma unroll
      for (int k = 0; k < DP; ++k) {
        const double da = xi[k] - Bs[jj][k], db = xi[k] - Bs[j1][k];
        ra = fma(da, da, ra);
        rb = fma(db, db, rb);
      }
      double va, vb;
      if (matern) {
        const double aa = sqrt_pos(ra), ab = sqrt_pos(rb);  // = sqrt(5) r
        va = (cp.alpha * exp_nonpos_tab(-aa, etab)) * fma(aa, fma(aa, 1.0 / 3.0, 1.0), 1.0);
        vb = (cp.alpha * exp_nonpos_tab(-ab, etab)) * fma(ab, fma(ab, 1.0 / 3.0, 1.0), 1.0);
      } else {
        va = cp.alpha * exp_nonpos_tab(fmax(-0.5 * ra, -1000.0), etab);
        vb = cp.alpha * exp_nonpos_tab(fmax(-0.5 * rb, -1000.0), etab);
      }
      const long col = col0 + j0 + jj;
      if (diag_noise != nullptr && (long)r == col - col0) va += noise;
      if (diag_noise != nullptr && (long)r == col + 1 - col0) vb += noise;
      // (streaming stores: the matrix is written once and read by a later kernel, nothing of it is reused from L2 here)
      __builtin_nontemporal_store(va, &out[(long)r + col * ld]);
      if (jj + 1 < nj) __builtin_nontemporal_store(vb, &out[(long)r + (col + 1) * ld]);
    }
  }
};
template <int DP>
__global__ __launch_bounds__(kCovRows) void cov_build_value_kernel(CovParams cp, const double* __restrict__ A, int nA,
                                                                  const double* __restrict__ B, int nB,
                                                                  const double* __restrict__ diag_noise,
                                                                  double* __restrict__ out, long ld, long col0) {
  cov_build_value_kernel_body<DP>::run(MOE_VBLOCK, MOE_VGRID, nullptr, cp, A, nA, B, nB, diag_noise, out, ld, col0);
}

// d cov(P_i, X_j)[m, n] / d P_{i,dd}: one thread per training row (j, n); P staged in LDS.
template <int DP, bool DERIVS>
struct grad_kstar_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const CovParams& cp, const double* __restrict__ X, int n, const DerivList& dX, const double* __restrict__ P, int nP, const DerivList& dP, double* __restrict__ out, long ld, long col0, int pts_per_block) {
    extern __shared__ double Ps[];  // [nP][DP]
    for (int t = threadIdx.x; t < nP * DP; t += blockDim.x) Ps[t] = P[t];
    __syncthreads();
    // grid.y cuts the points into runs of pts_per_block: with a thread per training row alone, 8000 rows are 32 workgroups that
    // each walk every point and column serially (336 us for two C5 evaluations' 768 columns, 0.15 TB/s)
    const int i_lo = blockIdx.y * pts_per_block, i_hi = min(nP, i_lo + pts_per_block);
    const int g = DERIVS ? dX.g : 0, gt = DERIVS ? dP.g : 0;
    const int rows = n * (1 + g);
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const int j = DERIVS ? r / (1 + g) : r;
    const int nb = DERIVS ? r % (1 + g) : 0;
    double xj[DP];
  #pragma unroll
    for (int k = 0; k < DP; ++k) xj[k] = X[(long)j * DP + k];
    for (int i = i_lo; i < i_hi; ++i) {
      double diff[DP];  // p1 - p2 = P_i - X_j
      double r2 = 0.0;
  #pragma unroll
      for (int k = 0; k < DP; ++k) {
        diff[k] = Ps[i * DP + k] - xj[k];
        r2 = fma(diff[k] * diff[k], cp.inv_l2[k], r2);
      }
      const Radial rd = radial_scalars(cp.type, cp.alpha, r2);
      if (!DERIVS) {
  #pragma unroll
        for (int dd = 0; dd < DP; ++dd) {
          if (dd < cp.dim) out[(long)r + (col0 + (long)i * cp.dim + dd) * ld] = (-diff[dd] * cp.inv_l2[dd]) * rd.first;
        }
      } else {
        for (int m = 0; m < 1 + gt; ++m)
          for (int dd = 0; dd < cp.dim; ++dd)
            out[(long)r + (col0 + ((long)i * (1 + gt) + m) * cp.dim + dd) * ld] =
                grad_cov_entry<DP>(cp, rd, diff, m, nb, dd, dP, dX);
      }
    }
  }
};
template <int DP, bool DERIVS>
__global__ __launch_bounds__(256) void grad_kstar_kernel(CovParams cp, const double* __restrict__ X, int n, DerivList dX,
                                                        const double* __restrict__ P, int nP, DerivList dP,
                                                        double* __restrict__ out, long ld, long col0, int pts_per_block) {
  grad_kstar_kernel_body<DP, DERIVS>::run(MOE_VBLOCK, MOE_VGRID, nullptr, cp, X, n, dX, P, nP, dP, out, ld, col0, pts_per_block);
}


// Posterior mean (and its spatial gradient) of the function value at nP query points in ONE launch:
//   mu_p = mean + sum_rows K(x_p, X)[0, row] KinvY[row],   d mu_p / d x_p,dd likewise with the gradient blocks
// (ComputeMeanOfPoints / ComputeGradMeanOfPoints for value rows, gpp_math.cpp:662-757).  One workgroup per query point,
// threads stride the training points, fixed-order block reduction.  This is the latency path of the boundary
// (compute_posterior_mean is called once per candidate point by the reference's Python loops): one kernel and one
// copy each way instead of the general state set-up.
template <int DP, bool GRAD>
__global__ __launch_bounds__(256) void mean_kernel(CovParams cp, const double* __restrict__ X, int n, DerivList dX,
                                                  const double* __restrict__ KinvY, const double* __restrict__ P,
                                                  double mean, double* __restrict__ out) {
  __shared__ double red[4][1 + DP];
  const int p = blockIdx.x;
  const int g1 = 1 + dX.g;
  DerivList none;
  none.g = 0;
  double xp[DP];
#pragma unroll
  for (int k = 0; k < DP; ++k) xp[k] = P[(long)p * DP + k];
  double acc = 0.0, accg[DP];
#pragma unroll
  for (int k = 0; k < DP; ++k) accg[k] = 0.0;
  for (int j = threadIdx.x; j < n; j += 256) {
    double diff[DP];
    double r2 = 0.0;
#pragma unroll
    for (int k = 0; k < DP; ++k) {
      diff[k] = xp[k] - X[(long)j * DP + k];
      r2 = fma(diff[k] * diff[k], cp.inv_l2[k], r2);
    }
    const Radial rd = radial_scalars(cp.type, cp.alpha, r2);
    for (int b = 0; b < g1; ++b) {
      const double w = KinvY[(long)j * g1 + b];
      acc = fma(cov_entry<DP>(cp, rd, diff, 0, b, none, dX), w, acc);
      if (GRAD) {
#pragma unroll
        for (int dd = 0; dd < DP; ++dd)
          if (dd < cp.dim) accg[dd] = fma(grad_cov_entry<DP>(cp, rd, diff, 0, b, dd, none, dX), w, accg[dd]);
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double v = acc;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  if (lane == 0) red[wave][0] = v;
  if (GRAD) {
#pragma unroll
    for (int dd = 0; dd < DP; ++dd) {
      double u = accg[dd];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) u += __shfl_xor(u, off, 64);
      if (lane == 0) red[wave][1 + dd] = u;
    }
  }
  __syncthreads();
  constexpr int W = GRAD ? 1 + DP : 1;
  if ((int)threadIdx.x < W) {
    const int c = threadIdx.x;
    const double tot = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    out[(long)p * W + c] = (c == 0) ? mean + tot : tot;
  }
}

bool value_fast_path() {  // MOE_COV_FAST=0: the general kernel for value-only blocks too (A/B runs, tests)
  const char* v = std::getenv("MOE_COV_FAST");
  return !(v && *v == '0');
}

template <int DP>
void cov_build_dp(const CovParams& cp, const double* A, int nA, const DerivList& dA, const double* B, int nB,
                  const DerivList& dB, const double* diag_noise, double* out, long ld, long col0, hipStream_t s, bool streaming,
                  bool lower_only) {
  const int lo = lower_only ? 1 : 0;
  const bool derivs = dA.g > 0 || dB.g > 0;
  const int rows = nA * (1 + dA.g);
  // column tile of a workgroup: 16 B points; the triangular build visits half the tiles and would leave the chip with ~2 workgroups
  // per CU at N = 8000 -- 2 points per workgroup there (r4, TB/s of stores by the symmetric byte count, N = 8000 / 26 000: 16 columns 2.4 / 2.7, 8: 3.2 / 3.6, 4: 3.9 / 4.5, 2: 4.1 / 4.8, 1: 3.8 / 5.0)
  static const int kxx_cols = [] {
    const char* v = std::getenv("MOE_KXX_COLS");
    return (v && *v) ? std::max(1, std::min(kCovCols, std::atoi(v))) : 2;
  }();
  const int cpw = lower_only ? kxx_cols : kCovCols;
  dim3 grid((nB + cpw - 1) / cpw, (rows + kCovRows - 1) / kCovRows);
  if (grid.x == 0 || grid.y == 0) return;
  if (dA.g > 0 && value_fast_path()) {  // thread per point, rows transposed through LDS (MOE_COV_FAST=0: the row-per-thread kernel)
    dim3 pgrid(grid.x, (nA + 255) / 256);
    launch_kernel_ens<cov_build_points_kernel_body<DP>, 256>(cov_build_points_kernel<DP>, pgrid, dim3(256), 0, s, cp, A, nA, dA, B, nB, dB, diag_noise, out, ld, col0, lo, cpw);
  } else if (derivs)
    launch_kernel_ens<cov_build_kernel_body<DP, true>, kCovRows>(cov_build_kernel<DP, true>, grid, dim3(kCovRows), 0, s, cp, A, nA, dA, B, nB, dB,
                                                                 diag_noise, out, ld, col0, lo, cpw, 0x7fffffff, 0L);
  else if (streaming && value_fast_path() && !lower_only)
    launch_kernel_ens<cov_build_value_kernel_body<DP>, kCovRows>(cov_build_value_kernel<DP>, grid, dim3(kCovRows), 0, s, cp, A, nA, B, nB, diag_noise, out, ld, col0);
  else
    launch_kernel_ens<cov_build_kernel_body<DP, false>, kCovRows>(cov_build_kernel<DP, false>, grid, dim3(kCovRows), 0, s, cp, A, nA, dA, B, nB,
                                                                  dB, diag_noise, out, ld, col0, lo, cpw, 0x7fffffff, 0L);
}

template <int DP>
void grad_kstar_dp(const CovParams& cp, const double* X, int n, const DerivList& dX, const double* P, int nP,
                   const DerivList& dP, double* out, long ld, long col0, hipStream_t s) {
  const bool derivs = dX.g > 0 || dP.g > 0;
  const int rows = n * (1 + dX.g);
  if (rows == 0 || nP == 0) return;
  constexpr int kChunk = 256;  // points staged in LDS per launch (batched states pass E * nd points)
  for (int p0 = 0; p0 < nP; p0 += kChunk) {
    const int np = std::min(kChunk, nP - p0);
    const int row_blocks = (rows + 255) / 256;
    const int slices = std::min(np, std::max(1, 1024 / row_blocks));  // ~4 workgroups per CU
    const int ppb = (np + slices - 1) / slices;
    dim3 grid(row_blocks, (np + ppb - 1) / ppb);
    const size_t shm = sizeof(double) * (size_t)np * DP;
    const long c0 = col0 + (long)p0 * (1 + dP.g) * cp.dim;
    if (derivs)
      launch_kernel_ens<grad_kstar_kernel_body<DP, true>, 256>(grad_kstar_kernel<DP, true>, grid, dim3(256), shm, s, cp, X, n, dX,
                                                               P + (long)p0 * DP, np, dP, out, ld, c0, ppb);
    else
      launch_kernel_ens<grad_kstar_kernel_body<DP, false>, 256>(grad_kstar_kernel<DP, false>, grid, dim3(256), shm, s, cp, X, n, dX,
                                                                P + (long)p0 * DP, np, dP, out, ld, c0, ppb);
  }
}

__global__ void debug_math_kernel(const double* __restrict__ x, int n, double* __restrict__ e, double* __restrict__ r) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  e[i] = exp_nonpos(-x[i]);
  r[i] = sqrt_nonneg(x[i]);
}

}  // namespace

void launch_debug_math(const double* x, int n, double* e, double* r, hipStream_t s) {
  MOE_LAUNCH(debug_math_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, n, e, r);
  MOE_HIP_CHECK(hipGetLastError());
}

void launch_cov_build(const CovParams& cp, const double* A, int nA, const DerivList& dA, const double* B, int nB,
                      const DerivList& dB, const double* diag_noise, double* out, long ld, long col0, hipStream_t s,
                      bool streaming, bool lower_only) {
  switch (cp.dp) {
    case 4: cov_build_dp<4>(cp, A, nA, dA, B, nB, dB, diag_noise, out, ld, col0, s, streaming, lower_only); break;
    case 8: cov_build_dp<8>(cp, A, nA, dA, B, nB, dB, diag_noise, out, ld, col0, s, streaming, lower_only); break;
    case 12: cov_build_dp<12>(cp, A, nA, dA, B, nB, dB, diag_noise, out, ld, col0, s, streaming, lower_only); break;
    case 16: cov_build_dp<16>(cp, A, nA, dA, B, nB, dB, diag_noise, out, ld, col0, s, streaming, lower_only); break;
    case 24: cov_build_dp<24>(cp, A, nA, dA, B, nB, dB, diag_noise, out, ld, col0, s, streaming, lower_only); break;
    case 32: cov_build_dp<32>(cp, A, nA, dA, B, nB, dB, diag_noise, out, ld, col0, s, streaming, lower_only); break;
    default: throw Error(MOE_ERR_RUNTIME, "unsupported padded dimension");
  }
  MOE_HIP_CHECK(hipGetLastError());
}

namespace {
template <int DP>
void cov_build_pair_dp(const CovParams& cp, const double* A, int nA, const double* B, int nB1, long col1, int nB2, long col2,
                       double* out, long ld, hipStream_t s) {
  DerivList none;
  none.g = 0;
  for (int i = 0; i < kMaxDerivs; ++i) none.idx[i] = 0;
  const int nB = nB1 + nB2;
  dim3 grid((nB + kCovCols - 1) / kCovCols, (nA + kCovRows - 1) / kCovRows);
  if (grid.x == 0 || grid.y == 0) return;
  // point b >= nB1 belongs at column col2 + (b - nB1) = col1 + b + (col2 - col1 - nB1)
  launch_kernel_ens<cov_build_kernel_body<DP, false>, kCovRows>(cov_build_kernel<DP, false>, grid, dim3(kCovRows), 0, s, cp, A, nA, none, B, nB, none,
                                                                (const double*)nullptr, out, ld, col1, 0, kCovCols, nB1,
                                                                col2 - col1 - (long)nB1);
}
}  // namespace

// K(A, [B1 | B2]) without derivative observations on either side, B1's nB1 columns from col1 on and B2's nB2 (the points right behind
// B1 in memory) from col2 on: the two builds of a q-KG state -- K* and K(X, discretised set) -- in ONE launch (r5).  Entry for entry
// what two launch_cov_build calls write.
void launch_cov_build_pair(const CovParams& cp, const double* A, int nA, const double* B, int nB1, long col1, int nB2, long col2,
                           double* out, long ld, hipStream_t s) {
  switch (cp.dp) {
    case 4: cov_build_pair_dp<4>(cp, A, nA, B, nB1, col1, nB2, col2, out, ld, s); break;
    case 8: cov_build_pair_dp<8>(cp, A, nA, B, nB1, col1, nB2, col2, out, ld, s); break;
    case 12: cov_build_pair_dp<12>(cp, A, nA, B, nB1, col1, nB2, col2, out, ld, s); break;
    case 16: cov_build_pair_dp<16>(cp, A, nA, B, nB1, col1, nB2, col2, out, ld, s); break;
    case 24: cov_build_pair_dp<24>(cp, A, nA, B, nB1, col1, nB2, col2, out, ld, s); break;
    case 32: cov_build_pair_dp<32>(cp, A, nA, B, nB1, col1, nB2, col2, out, ld, s); break;
    default: throw Error(MOE_ERR_RUNTIME, "unsupported padded dimension");
  }
  MOE_HIP_CHECK(hipGetLastError());
}

void launch_mean(const CovParams& cp, const double* X, int n, const DerivList& dX, const double* KinvY, const double* P,
                 int nP, double mean, bool want_grad, double* out, hipStream_t s) {
  if (nP <= 0) return;
#define MOE_MEAN_CASE(DPV)                                                                                                \
  case DPV:                                                                                                               \
    if (want_grad)                                                                                                        \
      MOE_LAUNCH((mean_kernel<DPV, true>), dim3(nP), dim3(256), 0, s, cp, X, n, dX, KinvY, P, mean, out);        \
    else                                                                                                                  \
      MOE_LAUNCH((mean_kernel<DPV, false>), dim3(nP), dim3(256), 0, s, cp, X, n, dX, KinvY, P, mean, out);       \
    break;
  switch (cp.dp) {
    MOE_MEAN_CASE(4)
    MOE_MEAN_CASE(8)
    MOE_MEAN_CASE(12)
    MOE_MEAN_CASE(16)
    MOE_MEAN_CASE(24)
    MOE_MEAN_CASE(32)
    default: throw Error(MOE_ERR_RUNTIME, "unsupported padded dimension");
  }
#undef MOE_MEAN_CASE
  MOE_HIP_CHECK(hipGetLastError());
}

void launch_grad_kstar(const CovParams& cp, const double* X, int n, const DerivList& dX, const double* P, int nP,
                       const DerivList& dP, double* out, long ld, long col0, hipStream_t s) {
  switch (cp.dp) {
    case 4: grad_kstar_dp<4>(cp, X, n, dX, P, nP, dP, out, ld, col0, s); break;
    case 8: grad_kstar_dp<8>(cp, X, n, dX, P, nP, dP, out, ld, col0, s); break;
    case 12: grad_kstar_dp<12>(cp, X, n, dX, P, nP, dP, out, ld, col0, s); break;
    case 16: grad_kstar_dp<16>(cp, X, n, dX, P, nP, dP, out, ld, col0, s); break;
    case 24: grad_kstar_dp<24>(cp, X, n, dX, P, nP, dP, out, ld, col0, s); break;
    case 32: grad_kstar_dp<32>(cp, X, n, dX, P, nP, dP, out, ld, col0, s); break;
    default: throw Error(MOE_ERR_RUNTIME, "unsupported padded dimension");
  }
  MOE_HIP_CHECK(hipGetLastError());
}

}  // namespace moe
