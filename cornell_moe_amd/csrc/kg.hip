// cornell_moe_amd/csrc/kg.hip -- q-KG / d-KG Monte-Carlo evaluation (value + gradient) on gfx950: host orchestration of
// one BATCH of independent evaluations (the multistart axis) and the kernels of the gradient tail.
//
// Pipeline for a batch of E evaluations (every stage is one launch, or one launch per kind, for the whole batch):
//   1. state set-up (gp.hip compute_state_batch): K*(X, Xu_e), dK*/dXq_e, K(X, discretised set_e) -> L^-1 / K^-1 applies
//      (tile GEMMs against the explicit inverse factor) -> Gram matrices; ONE device->host sync brings the c x c Grams back
//      and the m x m algebra of PointsToSampleState / KnowledgeGradientState (gpp_math.cpp:600-653,
//      gpp_knowledge_gradient_optimization.cpp:292-317) runs on the host (host_math.hip);
//   2. MC kernel (kg_mc.hpp): persistent wavefronts, one MC sample at a time per wave, all E evaluations in one launch;
//   3. gradient tail (.cpp:199-225 restated so that nothing of size M x (q d) is ever formed):
//        T   = K(X, x*_i) for every sample                      (covariance build, N x M per evaluation, HBM write-bound)
//        c_i = L^-1 ( K(Xu, x*_i) - W^T T_i )                   (kg_sw_kernel: one wave per sample)
//        TB  = sum_i T_i beta_i^T                               (kg_tb_kernel + fixed-order chunk reduction)
//      and three small reductions  ZC = sum_i z_i c_i^T (m x m),  DIR = sum_i beta_i,(k,b) dK(Xu_k, x*_i)[b,0]/dXu_k,
//      GTB = (K^-1 dK*/dXq)^T TB,  from which the host assembles
//        grad KG[k,dd] = ( [winner = k] M grad mu_k  -  sum_b (DIR - GTB)[(k,b),dd]  +  < L^-1 dL/dXq_k,dd , ZC > ) / M.
//      This is  sum_i z_i^T d(c_i)/dXq_k  of the reference (gpp_math.cpp:1601-1651) with the sum over samples pulled inside.
// All reductions use fixed orders, so results are bitwise reproducible for a given shard layout.
#include "kg.hpp"

#include <algorithm>
#include <array>
#include <atomic>
#include <exception>
#include <mutex>
#include <thread>
#include <memory>
#include <chrono>
#include <cmath>
#include <cstdlib>

#include "device_cov.hpp"
#include "fastmath.hpp"
#include "gemm128.hpp"
#include "kg_mc.hpp"
#include "kg_state.hpp"

namespace moe {

// how many ensemble members share the launches being recorded on this thread (mcmc.hip; kg_launch sizes its MC grid for its share)
static thread_local int t_ens_members_hint = 1;
int ensemble_members_hint() { return t_ens_members_hint; }
void set_ensemble_members_hint(int members) { t_ens_members_hint = members > 1 ? members : 1; }


namespace {

__device__ __forceinline__ double wave_sum64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Sum over a 256-thread workgroup in a fixed order; result valid in thread 0.
__device__ __forceinline__ double block_sum_256(double v, double* red) {
  const double w = wave_sum64(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

struct TabParams {
  int perm[kMaxDimPadded];
  double inv_lp[kMaxDimPadded];
  double center[kMaxDimPadded];  // training-set mean per table row (unscaled)
};

// XsTab[e][tile][r][lane] = coordinate perm[r] of point (tile*64 + lane) of evaluation e, minus the training-set mean of that
// coordinate, divided by its length scale (centred first: the difference of two nearby numbers is exact, so the scaled
// coordinates carry the precision of the distances however far from the origin the domain sits):
// the n training points, then the u union points of that evaluation, zero beyond.
// (r5: the workgroups of evaluation 0 also clear the call's counter block -- pass counters, sample tickets -- which a memset did before)
struct build_xs_tab_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const double* __restrict__ X, int n, const double* __restrict__ XuAll, int u, int dp, int ntiles, const TabParams& tp, double* __restrict__ tab, long tab_stride, int pair_rows, unsigned long long* __restrict__ ctr, long n_ctr) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    if (e == 0)
      for (long i = idx; i < n_ctr; i += (long)gridDim.x * blockDim.x) ctr[i] = 0ull;
    if (idx >= ntiles * dp * 64) return;
    const int l = idx & 63, r = (idx >> 6) % dp, t = (idx >> 6) / dp;
    const int j = t * 64 + l;
    const int k = tp.perm[r];
    double v = tp.center[r];  // padded points sit at the centre (zero weight; a far-away pad would only stress the exp)
    if (j < n)
      v = X[(long)j * dp + k];
    else if (j < n + u)
      v = XuAll[((long)e * u + (j - n)) * dp + k];
    // (pair_rows: the rows of a point in pairs, [tile][dp / 2][64][2] -- one 16-byte load per lane and pair, kg_mc.hpp WideEval)
    const long out = pair_rows ? (((long)t * (dp / 2) + (r >> 1)) * 64 + l) * 2 + (r & 1) : idx;
    tab[(long)e * tab_stride + out] = (v - tp.center[r]) * tp.inv_lp[r];
  }
};
__global__ __launch_bounds__(256) void build_xs_tab_kernel(const double* __restrict__ X, int n, const double* __restrict__ XuAll, int u, int dp,
                                    int ntiles, TabParams tp, double* __restrict__ tab, long tab_stride, int pair_rows,
                                    unsigned long long* __restrict__ ctr, long n_ctr) {
  build_xs_tab_kernel_body::run(MOE_VBLOCK, MOE_VGRID, nullptr, X, n, XuAll, u, dp, ntiles, tp, tab, tab_stride, pair_rows, ctr, n_ctr);
}

// ---------------------------------------------------------------------------------------------------------------------
// Gradient tail kernels
// ---------------------------------------------------------------------------------------------------------------------
struct KgTailParams {
  CovParams cp;
  DerivList derivs;  // the GP's derivative observations (carried by the union points too)
  int u, q, m, g, N, E, num_local, first_sample, ngrad, chunks, sw_chunk;
  int chunk_len;  // samples per TB partial (kTbChunk, or kFusedChunk on the T-free q-KG path): chunks = ceil(num_local / chunk_len)
  const double* T;           // [N x E*num_local], ld N
  const double* SW;          // optional precomputed W^T T, [m x E*num_local] col-major (large m: tile GEMM); else null
  const double* W;           // evaluation e at W + e * w_stride, [N x m], ld N
  long w_stride;
  const double* Gm;          // dK*/dXq: evaluation e at Gm + e * g_stride, [N x ngrad], ld N
  long g_stride;
  const double* Linv;        // the GP's explicit inverse factor (ld ldL) and 2 N E m doubles of workspace: K^-1 TB (launch_gtb)
  long ldL;
  double* work;
  double* tri_work;          // tri_cols_work_doubles(N, E m) doubles (split-K partials)
  const double* blob;
  KgRec rec;
  int rec_bp;                // offset of best_posterior inside a record
  const double* best_point;  // [E][num_local][dp]
  const double* best_value;  // [E][num_local]
  const double* beta;        // [E][num_local][m]
  const double* normals;     // [ceil(M/2)][m]
  double* C;                 // [E][num_local][m]   c_i = L^-1 cov_n(Xu, x*_i)
  double* TBpart;            // [E][chunks][m][N]
  double* out;               // [E][out_stride]: kg_sum | ZC (m*m, col-major) | DIR (ngrad) | GTB (ngrad)
  int out_stride;
};

constexpr int kTbChunk = 256;  // samples per workgroup of kg_tb_kernel
constexpr int kTbChunkWide = 2048;  // m > 64: samples per partial of the matrix-pipe TB product (kg_tb128_kernel)
// ... and of kg_fused_point_kernel: 128, so that ONE q-KG evaluation (n = 1000: 4 row blocks x 79 chunks) puts more than one
// workgroup on every CU -- 40.7 -> ~20 us of a batch-1 evaluation's tail.  Fixed per path, not per batch size: an evaluation's
// bits do not depend on what it is batched with.
constexpr int kFusedChunk = 128;
constexpr int kDirSlices = 32;  // sample ranges of kg_dir_kernel

// c_i = L^-1 ( K(Xu, x*_i)[:, 0] - W^T T_i ) for every sample; one wavefront per sample at a time, lane r owns component r.
// S_W = W^T T_i comes either precomputed (P.SW: tile GEMM, large m) or is formed here by lanes striding the N rows with
// W_e staged ONCE per workgroup in LDS ([c][row], conflict-free) and reused for the kSwChunk samples of the workgroup --
// re-reading W from L2 for every sample (N m 8 bytes each) made this the slowest kernel of the tail.
constexpr int kSwChunk = 64;  // samples per workgroup (16 per wavefront); 16 when the batch is too small to fill the chip

// SLOTS = 2: components lane and lane + 64 (m up to 128; S_W always precomputed there).
template <int DP, int MU, int SLOTS = 1>
struct kg_sw_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const KgTailParams& P) {
    extern __shared__ __attribute__((aligned(16))) double Ws[];  // [m][N] when P.SW == nullptr
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = blockIdx.y;
    const int m = P.m, g1 = 1 + P.g, N = P.N;
    const double* rec = P.blob + (long)e * P.rec.stride;
    const double* Lsm = rec + P.rec.L;
    if (P.SW == nullptr) {
      const double* We = P.W + (long)e * P.w_stride;
      for (int t = threadIdx.x; t < N * m; t += 256) Ws[t] = We[t];  // W_e is [N x m] col-major == [c][row]
      __syncthreads();
    }
    const int i1 = min(P.num_local, (int)(blockIdx.x + 1) * P.sw_chunk);
    for (int i = blockIdx.x * P.sw_chunk + wave; i < i1; i += 4) {
      const long w = (long)e * P.num_local + i;
      double mine[SLOTS];
  #pragma unroll
      for (int sl = 0; sl < SLOTS; ++sl) mine[sl] = 0.0;
      if (P.SW != nullptr) {
  #pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl)
          if (lane + 64 * sl < m) mine[sl] = P.SW[w * m + lane + 64 * sl];
      } else {
        const double* Tc = P.T + w * N;
        double acc[MU];
  #pragma unroll
        for (int c = 0; c < MU; ++c) acc[c] = 0.0;
  #pragma unroll 4
        for (int row = lane; row < N; row += 64) {
          const double t = Tc[row];
  #pragma unroll
          for (int c = 0; c < MU; ++c)
            if (c < m) acc[c] = fma(Ws[c * N + row], t, acc[c]);
        }
  #pragma unroll
        for (int c = 0; c < MU; ++c) {
          const double v = wave_sum64(acc[c]);
          if (lane == c) mine[0] = v;
        }
      }
      double R[SLOTS], cv[SLOTS];
  #pragma unroll
      for (int sl = 0; sl < SLOTS; ++sl) {
        R[sl] = 0.0;
        cv[sl] = 0.0;
        const int comp = lane + 64 * sl;
        if (comp < m) {
          const int r = comp / g1, b = comp - r * g1;
          const double* Xu = rec + P.rec.XuP + (long)r * DP;
          const double* xs = P.best_point + w * DP;
          double diff[DP];
          double r2 = 0.0;
  #pragma unroll
          for (int k = 0; k < DP; ++k) {
            diff[k] = Xu[k] - xs[k];
            r2 = fma(diff[k] * diff[k], P.cp.inv_l2[k], r2);
          }
          const Radial rd = radial_scalars(P.cp.type, P.cp.alpha, r2);
          DerivList none;
          none.g = 0;
          R[sl] = cov_entry<DP>(P.cp, rd, diff, b, 0, P.derivs, none) - mine[sl];
        }
      }
      for (int r = 0; r < m; ++r) {
        double part = 0.0;
  #pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl)
          if (lane + 64 * sl < r) part = fma(Lsm[r + (long)(lane + 64 * sl) * m], cv[sl], part);
        const double tot = wave_sum64(part);
  #pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl)
          if (lane + 64 * sl == r) cv[sl] = (R[sl] - tot) / Lsm[r + (long)r * m];
      }
  #pragma unroll
      for (int sl = 0; sl < SLOTS; ++sl)
        if (lane + 64 * sl < m) P.C[w * m + lane + 64 * sl] = cv[sl];
    }
  }
};
template <int DP, int MU, int SLOTS = 1>
__global__ __launch_bounds__(256) void kg_sw_kernel(KgTailParams P) {
  kg_sw_kernel_body<DP, MU, SLOTS>::run(MOE_VBLOCK, MOE_VGRID, nullptr, P);
}

// TBpart[e][chunk][c][row] = sum over the chunk's samples of T[row, i] beta_i[c]   (thread = row; T read coalesced, the
// beta row is wave-uniform).
template <int MU>
struct kg_tb_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const KgTailParams& P, int c_lo) {    // columns [c_lo, c_lo + MU) of beta / TB
    const int row = blockIdx.x * 256 + threadIdx.x;
    const int chunk = blockIdx.y, e = blockIdx.z;
    const int m = P.m;
    const int i0 = chunk * P.chunk_len, i1 = min(P.num_local, i0 + P.chunk_len);
    double acc[MU];
  #pragma unroll
    for (int c = 0; c < MU; ++c) acc[c] = 0.0;
    const bool ok = row < P.N;
    const double* __restrict__ Tcol = P.T + (ok ? row : 0);
    const double* __restrict__ beta = P.beta;
  #pragma unroll 4
    for (int i = i0; i < i1; ++i) {
      const long w = (long)e * P.num_local + i;
      const double t = Tcol[w * P.N];
      const double* __restrict__ b = beta + w * m + c_lo;
      // all MU entries unconditionally (uniform, contiguous: wide scalar loads, no branch per entry); the entries beyond m
      // belong to the next sample / the pad behind the buffer and only feed accumulators that are never stored
  #pragma unroll
      for (int c = 0; c < MU; ++c) acc[c] = fma(t, b[c], acc[c]);
    }
    if (ok) {
      double* dst = P.TBpart + ((long)e * P.chunks + chunk) * m * P.N;
  #pragma unroll
      for (int c = 0; c < MU; ++c)
        if (c_lo + c < m) dst[(long)(c_lo + c) * P.N + row] = acc[c];
    }
  }
};
template <int MU>
__global__ __launch_bounds__(256) void kg_tb_kernel(KgTailParams P, int c_lo = 0) {
  kg_tb_kernel_body<MU>::run(MOE_VBLOCK, MOE_VGRID, nullptr, P, c_lo);
}

// The same partial sums for m > 64 (r4; the stretch point's m = 104) as a product on the matrix pipe: TBpart[e][chunk] (N x m) =
// T_e[:, chunk] (N x len) beta_e[chunk, :] (len x m) through gemm128.hpp's tile core -- one pass over T instead of one per 64 columns
// of beta, chunks of kTbChunkWide samples (ten partials instead of 79 at M = 20 000).  Grid (row tiles of 128, chunks, E).
struct kg_tb128_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const KgTailParams& P) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int chunk = blockIdx.y, e = blockIdx.z, m = P.m;
    const int i0 = chunk * P.chunk_len, len = min(P.num_local - i0, P.chunk_len);
    const long w0 = (long)e * P.num_local + i0;
    g128::Operand A{P.T + w0 * P.N, (long)P.N, P.N, len, 1};     // T[row + sample N]: rows contiguous
    g128::Operand B{P.beta + w0 * m, (long)m, m, len, 1};        // beta[sample m + c]: the output column contiguous
    g128::f64x4 acc[4][4];
    const int r0 = blockIdx.x * g128::TM;
    g128::tile_product<false, false>(A, B, r0, 0, 0, len, smem, acc);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wi = (wave & 1) * 64, wj = (wave >> 1) * 64;
    const int lk = lane >> 4, lx = lane & 15;
    double* dst = P.TBpart + ((long)e * P.chunks + chunk) * m * P.N;
  #pragma unroll
    for (int a = 0; a < 4; ++a)
  #pragma unroll
      for (int b = 0; b < 4; ++b)
  #pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = r0 + wi + 16 * a + lx, c = wj + 16 * b + lk + 4 * r;
          if (row < P.N && c < m) dst[(long)c * P.N + row] = acc[a][b][r];
        }
  }
};
__global__ __launch_bounds__(256, 2) void kg_tb128_kernel(KgTailParams P) {
  kg_tb128_kernel_body::run(MOE_VBLOCK, MOE_VGRID, nullptr, P);
}

// S_W = W_e^T T_e (m x samples) for 64 < m <= 128 on the same tile core: one 128-row tile holds all m rows, so T is read ONCE (the
// 64-tile kernel reads it once per row tile: twice at m = 104).  K = N is cut into `slices`; slice sl of evaluation e goes to
// SWpart[sl][e] (dense m x num_local), summed in slice order by sum_slices_kernel.  Grid (column tiles of 128 samples, slices, E).
struct kg_sw128_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const KgTailParams& P, double* __restrict__ SWpart, int slices) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int sl = blockIdx.y, e = blockIdx.z, m = P.m;
    const int ks = ((P.N + slices - 1) / slices + g128::TK - 1) / g128::TK * g128::TK;
    const int k_lo = sl * ks, k_hi = min(P.N, k_lo + ks);
    g128::Operand A{P.W + (long)e * P.w_stride, (long)P.N, m, P.N, 1};                        // W[row + c N]: K (= row) contiguous
    g128::Operand B{P.T + (long)e * P.num_local * P.N, (long)P.N, P.num_local, P.N, 1};      // T[row + sample N]: K contiguous
    g128::f64x4 acc[4][4];
    const int j0 = blockIdx.x * g128::TM;
    g128::tile_product<true, true>(A, B, 0, j0, k_lo, k_hi, smem, acc);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wi = (wave & 1) * 64, wj = (wave >> 1) * 64;
    const int lk = lane >> 4, lx = lane & 15;
    double* dst = SWpart + ((long)sl * P.E + e) * m * P.num_local;
  #pragma unroll
    for (int a = 0; a < 4; ++a)
  #pragma unroll
      for (int b = 0; b < 4; ++b)
  #pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = wi + 16 * a + lx, smp = j0 + wj + 16 * b + lk + 4 * r;
          if (c < m && smp < P.num_local) dst[(long)smp * m + c] = acc[a][b][r];
        }
  }
};
__global__ __launch_bounds__(256, 2) void kg_sw128_kernel(KgTailParams P, double* __restrict__ SWpart, int slices) {
  kg_sw128_kernel_body::run(MOE_VBLOCK, MOE_VGRID, nullptr, P, SWpart, slices);
}

// TB[e][c][row] = sum of the chunk partials in chunk order (one pass over TBpart instead of one per gradient column).
struct kg_tbsum_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const KgTailParams& P, double* __restrict__ TBsum) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;  // over m * N
    const int e = blockIdx.y;
    const long mn = (long)P.m * P.N;
    if (idx >= mn) return;
    const double* part = P.TBpart + (long)e * P.chunks * mn + idx;
    double tb = 0.0;
  #pragma unroll 8
    for (int ch = 0; ch < P.chunks; ++ch) tb += part[(long)ch * mn];  // (independent loads, issued eight at a time)
    TBsum[(long)e * mn + idx] = tb;
  }
};
__global__ __launch_bounds__(256) void kg_tbsum_kernel(KgTailParams P, double* __restrict__ TBsum) {
  kg_tbsum_kernel_body::run(MOE_VBLOCK, MOE_VGRID, nullptr, P, TBsum);
}

// GTB[e][gc] = sum_row dK*_e[row, gc] * (K^-1 TB_e)[row, col(gc)];  gc = (k (1+g) + b) d + dd  ->  col = k (1+g) + b.
struct kg_gtb_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const KgTailParams& P, const double* __restrict__ TBsum) {
    __shared__ double red[4];
    const int gc = blockIdx.x, e = blockIdx.y;
    const int col = gc / P.cp.dim;
    const double* Gc = P.Gm + (long)e * P.g_stride + (long)gc * P.N;
    const double* tb = TBsum + ((long)e * P.m + col) * P.N;
    double acc = 0.0;
    for (int row = threadIdx.x; row < P.N; row += 256) acc = fma(Gc[row], tb[row], acc);
    const double tot = block_sum_256(acc, red);
    if (threadIdx.x == 0) P.out[(long)e * P.out_stride + 1 + P.m * P.m + P.ngrad + gc] = tot;
  }
};
__global__ __launch_bounds__(256) void kg_gtb_kernel(KgTailParams P, const double* __restrict__ TBsum) {
  kg_gtb_kernel_body::run(MOE_VBLOCK, MOE_VGRID, nullptr, P, TBsum);
}

// The summed TB lives behind the chunk partials in the same buffer (the host reserves E (chunks + 1) m N doubles).
void launch_gtb(const KgTailParams& P, hipStream_t s) {
  const long mn = (long)P.m * P.N;
  double* TBsum = P.TBpart + (long)P.E * P.chunks * mn;
  launch_kernel_ens<kg_tbsum_kernel_body, 256>(kg_tbsum_kernel, dim3((unsigned)((mn + 255) / 256), P.E), dim3(256), 0, s, P, TBsum);
  // (K^-1 dK*)^T TB = dK*^T (K^-1 TB): the two triangular products run over the m columns of TB, not over the q (1 + g) d columns of
  // dK* (r4: 32 against 384 per evaluation at C5)
  const int cm = P.E * P.m;
  double* half = P.work;
  double* KinvTB = P.work + (long)P.N * cm;
  launch_tri_gemm_cols('N', P.N, cm, P.m, P.Linv, P.ldL, TBsum, P.N, half, P.N, P.tri_work, s);
  launch_tri_gemm_cols('T', P.N, cm, P.m, P.Linv, P.ldL, half, P.N, KinvTB, P.N, P.tri_work, s);
  launch_kernel_ens<kg_gtb_kernel_body, 256>(kg_gtb_kernel, dim3(P.ngrad, P.E), dim3(256), 0, s, P, (const double*)KinvTB);
}

// ZC[e][r + j m] = sum_i z_i[r] c_i[j] and kg_sum = sum_i (best_posterior + best_value_i)   (.cpp:196).
// Without a strided sample walk (round 1: one workgroup per (r, j) reading c_i[j] every m doubles: 254 us for two
// C5 evaluations): a workgroup takes a chunk of kZcChunk samples, stages their z and c rows in LDS with coalesced loads and forms all
// m x m partial products; kg_zc_sum_kernel adds the chunk partials in chunk order and forms kg_sum exactly as kg_sum_kernel does.
constexpr int kZcChunk = 64;  // samples per chunk (32 for m > 32: the two staged blocks stay within 32 KB)
inline int zc_chunk_len(int m) { return m > 32 ? kZcChunk / 2 : kZcChunk; }  // (r4: m > 64 too -- 2 x 32 x 128 doubles: 64 KB, opted in)
struct kg_zc_part_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const KgTailParams& P, double* __restrict__ part, int chunks, int len) {
    extern __shared__ __attribute__((aligned(16))) double zc_sm[];  // z [len][m] | c [len][m]
    const int chunk = blockIdx.x, e = blockIdx.y, m = P.m;
    const int i0 = chunk * len, cnt = min(len, P.num_local - i0);
    double* zs = zc_sm;
    double* cs = zc_sm + len * m;
    for (int t = threadIdx.x; t < cnt * m; t += 256) {
      const int ii = t / m, r = t - ii * m;
      const int s = P.first_sample + i0 + ii;
      zs[t] = ((s & 1) ? -1.0 : 1.0) * P.normals[(long)(s >> 1) * m + r];
      cs[t] = P.C[((long)e * P.num_local + i0 + ii) * m + r];
    }
    __syncthreads();
    for (int o = threadIdx.x; o < m * m; o += 256) {
      const int r = o % m, j = o / m;
      double acc = 0.0;
  #pragma unroll 4
      for (int ii = 0; ii < cnt; ++ii) acc = fma(zs[ii * m + r], cs[ii * m + j], acc);
      part[((long)e * chunks + chunk) * m * m + o] = acc;
    }
  }
};
__global__ __launch_bounds__(256) void kg_zc_part_kernel(KgTailParams P, double* __restrict__ part, int chunks, int len) {
  kg_zc_part_kernel_body::run(MOE_VBLOCK, MOE_VGRID, nullptr, P, part, chunks, len);
}

// (r4: `gs` lanes share an output -- a power of two, m m gs <= 256 -- and stride the chunks, then a fixed butterfly: a lane per output
//  walked all the chunks in batches of eight loads, one memory round trip per batch: 19 us for the 157 chunks of one C3 evaluation)
struct kg_zc_sum_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const KgTailParams& P, const double* __restrict__ part, int chunks, int gs) {
    __shared__ double red[4];
    const int e = blockIdx.y, m = P.m;
    const int per_block = 256 / gs;
    const int o = blockIdx.x * per_block + (int)threadIdx.x / gs, g = (int)threadIdx.x % gs;
    double* out = P.out + (long)e * P.out_stride;
    {
      const bool ok = o < m * m;
      const double* p = part + (long)e * chunks * m * m + (ok ? o : 0);
      double v = 0.0;
  #pragma unroll 8
      for (int ch = g; ch < chunks; ch += gs) v += p[(long)ch * m * m];
      for (int off = gs >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      if (ok && g == 0) out[1 + o] = v;
    }
    if (blockIdx.x == 0) {  // kg_sum = sum_i (best_posterior + best_value_i): the summation of kg_sum_kernel, bit for bit
      const double bp = P.blob[(long)e * P.rec.stride + P.rec_bp];
      double acc = 0.0;
  #pragma unroll 8
      for (int i = threadIdx.x; i < P.num_local; i += 256) acc += bp + P.best_value[(long)e * P.num_local + i];
      const double tot = block_sum_256(acc, red);
      if (threadIdx.x == 0) out[0] = tot;
    }
  }
};
__global__ __launch_bounds__(256) void kg_zc_sum_kernel(KgTailParams P, const double* __restrict__ part, int chunks, int gs) {
  kg_zc_sum_kernel_body::run(MOE_VBLOCK, MOE_VGRID, nullptr, P, part, chunks, gs);
}

int env_int(const char* name, int dflt);
inline int zc_sum_lanes(int m) {  // lanes that share an entry of ZC in kg_zc_sum_kernel
  int gs = 1;
  while (gs < 64 && m * m * gs * 2 <= 256) gs *= 2;
  return gs;
}
// r5: for small m and few chunks the sum of the partials is taken by kg_finish_kernel itself (one launch less on the latency path)
inline bool zc_sum_in_finish(int m, int num_local) {
  const long chunks = (num_local + zc_chunk_len(m) - 1) / zc_chunk_len(m);
  return m <= 8 && chunks * m * m <= 16384 && env_int("MOE_KG_ZC_IN_FINISH", 1) != 0;
}

// host side of the two: part = E * chunks * m * m doubles of workspace
void launch_zc(const KgTailParams& P, double* part, hipStream_t s, bool sum_here = true) {
  const int len = zc_chunk_len(P.m);
  const int chunks = (P.num_local + len - 1) / len;
  const size_t shm = sizeof(double) * 2 * len * P.m;
  if (shm > 48 * 1024)
    MOE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kg_zc_part_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  launch_kernel_ens<kg_zc_part_kernel_body, 256>(kg_zc_part_kernel, dim3(chunks, P.E), dim3(256), shm, s, P, part, chunks, len);
  if (!sum_here) return;
  const int gs = zc_sum_lanes(P.m);
  const int per_block = 256 / gs;
  launch_kernel_ens<kg_zc_sum_kernel_body, 256>(kg_zc_sum_kernel, dim3((P.m * P.m + per_block - 1) / per_block, P.E), dim3(256), 0, s, P, (const double*)part, chunks, gs);
}

// DIR[e][(k (1+g) + b) d + dd] = sum_i beta_i[(k,b)] * d cov(Xu_k, x*_i)[b, 0] / d Xu_k,dd ; workgroup (k, e).
template <int DP>
struct kg_dir_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const KgTailParams& P, double* __restrict__ part, int slices) {
    __shared__ double red[4];
    const int k = blockIdx.x, e = blockIdx.y, sl = blockIdx.z;
    // the samples are cut into `slices` contiguous ranges (grid.z) so that q x E workgroups become q x E x slices -- at C5
    // eight workgroups walked 20 000 samples each while 248 CUs idled; kg_dir_sum_kernel adds the partials in slice order
    const int per = (P.num_local + slices - 1) / slices;
    const int i_lo = sl * per, i_hi = min(P.num_local, i_lo + per);
    const int m = P.m, g1 = 1 + P.g, d = P.cp.dim;
    const double* Xu = P.blob + (long)e * P.rec.stride + P.rec.XuP + (long)k * DP;
    DerivList none;
    none.g = 0;
    double xk[DP];
  #pragma unroll
    for (int kk = 0; kk < DP; ++kk) xk[kk] = Xu[kk];
    for (int b = 0; b < g1; ++b) {
      double acc[DP];
  #pragma unroll
      for (int dd = 0; dd < DP; ++dd) acc[dd] = 0.0;
      for (int i = i_lo + threadIdx.x; i < i_hi; i += 256) {
        const long w = (long)e * P.num_local + i;
        const double* xs = P.best_point + w * DP;
        double diff[DP];
        double r2 = 0.0;
  #pragma unroll
        for (int kk = 0; kk < DP; ++kk) {
          diff[kk] = xk[kk] - xs[kk];
          r2 = fma(diff[kk] * diff[kk], P.cp.inv_l2[kk], r2);
        }
        const Radial rd = radial_scalars(P.cp.type, P.cp.alpha, r2);
        const double bt = P.beta[w * m + k * g1 + b];
  #pragma unroll
        for (int dd = 0; dd < DP; ++dd)
          if (dd < d) acc[dd] = fma(bt, grad_cov_entry<DP>(P.cp, rd, diff, b, 0, dd, P.derivs, none), acc[dd]);
      }
  #pragma unroll
      for (int dd = 0; dd < DP; ++dd) {
        if (dd < d) {  // d is workgroup-uniform
          const double tot = block_sum_256(acc[dd], red);
          if (threadIdx.x == 0) part[((long)e * P.ngrad + (k * g1 + b) * d + dd) * slices + sl] = tot;
        }
      }
    }
  }
};
template <int DP>
__global__ __launch_bounds__(256) void kg_dir_kernel(KgTailParams P, double* __restrict__ part, int slices) {
  kg_dir_kernel_body<DP>::run(MOE_VBLOCK, MOE_VGRID, nullptr, P, part, slices);
}

// (the slices are added up in order by kg_finish_kernel, where DIR is consumed -- r5; a kernel of its own before)
inline int dir_slices(int num_local) { return std::max(1, std::min(kDirSlices, (num_local + 255) / 256)); }
inline double* dir_partials(const KgTailParams& P) {
  // behind the summed TB in the TB buffer (the host reserves E * ngrad * kDirSlices doubles more)
  return P.TBpart + (long)P.E * (P.chunks + 1) * (long)P.m * P.N;
}

template <int DP>
void launch_dir(const KgTailParams& P, hipStream_t s) {
  const int slices = dir_slices(P.num_local);
  launch_kernel_ens<kg_dir_kernel_body<DP>, 256>(kg_dir_kernel<DP>, dim3(P.q, P.E, slices), dim3(256), 0, s, P, dir_partials(P), slices);
}

template <int DP, int MU>
void launch_sw_inst(const KgTailParams& P, hipStream_t s) {
  dim3 grid((P.num_local + P.sw_chunk - 1) / P.sw_chunk, P.E);
  const size_t shm = (P.SW == nullptr) ? sizeof(double) * (size_t)P.N * P.m : 0;
  auto kern = kg_sw_kernel<DP, MU>;
  if (shm > 48 * 1024)
    MOE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  launch_kernel_ens<kg_sw_kernel_body<DP, MU>, 256>(kern, grid, dim3(256), shm, s, P);
}

template <int DP>
void launch_sw_dp(const KgTailParams& P, hipStream_t s) {
  if (P.m <= 4)
    launch_sw_inst<DP, 4>(P, s);
  else if (P.m <= 8)
    launch_sw_inst<DP, 8>(P, s);
  else if (P.m <= 16)
    launch_sw_inst<DP, 16>(P, s);
  else if (P.m <= 32)
    launch_sw_inst<DP, 32>(P, s);
  else if (P.m <= 64)
    launch_sw_inst<DP, 64>(P, s);
  else {  // two components per lane; S_W comes precomputed (the host always forms it by GEMM for m > 8)
    if (P.SW == nullptr) throw Error(MOE_ERR_RUNTIME, "m > 64 needs the precomputed W^T T");
    dim3 grid((P.num_local + P.sw_chunk - 1) / P.sw_chunk, P.E);
    launch_kernel_ens<kg_sw_kernel_body<DP, 1, 2>, 256>(kg_sw_kernel<DP, 1, 2>, grid, dim3(256), 0, s, P);
  }
  launch_dir<DP>(P, s);
}

void launch_tail(const KgTailParams& P, hipStream_t s) {
  switch (P.cp.dp) {
    case 4: launch_sw_dp<4>(P, s); break;
    case 8: launch_sw_dp<8>(P, s); break;
    case 12: launch_sw_dp<12>(P, s); break;
    case 16: launch_sw_dp<16>(P, s); break;
    case 24: launch_sw_dp<24>(P, s); break;
    case 32: launch_sw_dp<32>(P, s); break;
    default: throw Error(MOE_ERR_RUNTIME, "unsupported padded dimension");
  }
  dim3 gtb((P.N + 255) / 256, P.chunks, P.E);
  if (P.m <= 4)
    launch_kernel_ens<kg_tb_kernel_body<4>, 256>(kg_tb_kernel<4>, gtb, dim3(256), 0, s, P, 0);
  else if (P.m <= 8)
    launch_kernel_ens<kg_tb_kernel_body<8>, 256>(kg_tb_kernel<8>, gtb, dim3(256), 0, s, P, 0);
  else if (P.m <= 16)
    launch_kernel_ens<kg_tb_kernel_body<16>, 256>(kg_tb_kernel<16>, gtb, dim3(256), 0, s, P, 0);
  else if (P.m <= 32)
    launch_kernel_ens<kg_tb_kernel_body<32>, 256>(kg_tb_kernel<32>, gtb, dim3(256), 0, s, P, 0);
  else if (P.m <= 64) {
    launch_kernel_ens<kg_tb_kernel_body<64>, 256>(kg_tb_kernel<64>, gtb, dim3(256), 0, s, P, 0);
  } else {  // (m <= kMaxMB = 128: one column tile)
    MOE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kg_tb128_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)g128::kSmemBytes));
    launch_kernel_ens<kg_tb128_kernel_body, 256, 2>(kg_tb128_kernel, dim3((P.N + g128::TM - 1) / g128::TM, P.chunks, P.E), dim3(256), g128::kSmemBytes, s, P);
  }
  launch_gtb(P, s);
  MOE_HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused tail for q-KG (no derivative observations, m = q + p <= 8): the N x M matrix T = K(X, x*_i) is never written.
// Its entries are recomputed where they are consumed -- once with a thread per SAMPLE (S_W = W^T T_i, then c_i), once with a
// thread per training POINT (TB partials) -- which costs ~2 x 46 FP64 instructions per entry against 16 B written + 24 B
// re-read per entry through HBM: at C3 the covariance build + S_W + TB kernels took 86 us per evaluation, these two take
// about half.  Operands that are uniform over the workgroup (the point tile in the first kernel, the sample chunk in the
// second) are staged in LDS pre-scaled by 1/length and read as broadcasts.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kFusedTile = 256;  // points (kernel A) / samples (kernel B, at most: KgTailParams::chunk_len) staged per LDS tile

template <int COV>
__device__ __forceinline__ double radial_base(double r2, const double* __restrict__ etab) {
  if (COV == MOE_COV_SQUARE_EXPONENTIAL) return exp_nonpos_tab(fmax(-0.5 * r2, -1000.0), etab);  // (table exp range)
  const double a = 2.236067977499789696409173668731276235 * sqrt_pos(r2);
  return exp_nonpos_tab(-a, etab) * fma(a, fma(a, 1.0 / 3.0, 1.0), 1.0);
}

// thread = sample, blockIdx.z = slice of the training points: SWpart[e][i][slice][c] = sum over the slice of W[j, c] k(X_j, x*_i)
// (the slices only exist to give the chip enough wavefronts: E * M / 64 alone is ~1 per SIMD at the headline shape)
template <int DP, int MU, int COV>
struct kg_fused_sample_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const KgTailParams& P, const double* __restrict__ X, int n, double* __restrict__ SWpart) {
    __shared__ double etab[kExpTabLen];
    __shared__ double Xt[kFusedTile][DP];
    __shared__ double Wt[kFusedTile][MU];
    const int e = blockIdx.y, m = P.m, N = P.N;
    const int slices = gridDim.z, slice = blockIdx.z;
    const int per = ((n + slices - 1) / slices + 63) / 64 * 64;
    const int j_lo = slice * per, j_hi = min(n, j_lo + per);
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool ok = i < P.num_local;
    const long w = (long)e * P.num_local + (ok ? i : 0);
    if (threadIdx.x < kExpTabLen) etab[threadIdx.x] = kExp2Tab64[threadIdx.x];
    double xs[DP], acc[MU];
  #pragma unroll
    for (int k = 0; k < DP; ++k) xs[k] = P.best_point[w * DP + k] * P.cp.inv_l[k];
  #pragma unroll
    for (int c = 0; c < MU; ++c) acc[c] = 0.0;
    const double* We = P.W + (long)e * P.w_stride;
    for (int j0 = j_lo; j0 < j_hi; j0 += kFusedTile) {
      __syncthreads();
      for (int t = threadIdx.x; t < kFusedTile * DP; t += 256) {
        const int jj = t / DP, k = t % DP;
        Xt[jj][k] = (j0 + jj < j_hi) ? X[(long)(j0 + jj) * DP + k] * P.cp.inv_l[k] : 0.0;
      }
      for (int t = threadIdx.x; t < kFusedTile * MU; t += 256) {
        const int jj = t % kFusedTile, c = t / kFusedTile;  // W_e is [N x m] col-major: consecutive jj are contiguous
        Wt[jj][c] = (j0 + jj < j_hi && c < m) ? We[(long)c * N + j0 + jj] : 0.0;  // zero weight: padded points drop out
      }
      __syncthreads();
      const int cnt = min(kFusedTile, j_hi - j0);
  #pragma unroll 2
      for (int jj = 0; jj < cnt; ++jj) {
        double r2 = 1.0e-300;
  #pragma unroll
        for (int k = 0; k < DP; ++k) {
          const double dlt = Xt[jj][k] - xs[k];
          r2 = fma(dlt, dlt, r2);
        }
        const double t = radial_base<COV>(r2, etab);
  #pragma unroll
        for (int c = 0; c < MU; ++c) acc[c] = fma(Wt[jj][c], t, acc[c]);
      }
    }
    if (!ok) return;
  #pragma unroll
    for (int c = 0; c < MU; ++c) SWpart[(w * slices + slice) * MU + c] = acc[c];
  }
};
template <int DP, int MU, int COV>
__global__ __launch_bounds__(256) void kg_fused_sample_kernel(KgTailParams P, const double* __restrict__ X, int n,
                                                             double* __restrict__ SWpart) {
  kg_fused_sample_kernel_body<DP, MU, COV>::run(MOE_VBLOCK, MOE_VGRID, nullptr, P, X, n, SWpart);
}

// thread = sample: S_W = sum of the slices (fixed order); R_r = alpha (k(Xu_r, x*_i) - S_W[r]); c_i = L^-1 R by forward
// substitution (L is m x m, column-major, workgroup-uniform)
template <int DP, int MU, int COV>
struct kg_fused_c_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const KgTailParams& P, const double* __restrict__ SWpart, int slices) {
    __shared__ double etab[kExpTabLen];
    const int e = blockIdx.y, m = P.m;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (threadIdx.x < kExpTabLen) etab[threadIdx.x] = kExp2Tab64[threadIdx.x];
    __syncthreads();
    if (i >= P.num_local) return;
    const long w = (long)e * P.num_local + i;
    double xs[DP], sw[MU], cv[MU];
  #pragma unroll
    for (int k = 0; k < DP; ++k) xs[k] = P.best_point[w * DP + k] * P.cp.inv_l[k];
  #pragma unroll
    for (int c = 0; c < MU; ++c) {
      sw[c] = 0.0;
      for (int sl = 0; sl < slices; ++sl) sw[c] += SWpart[(w * slices + sl) * MU + c];
    }
    const double* rec = P.blob + (long)e * P.rec.stride;
    const double* Lsm = rec + P.rec.L;
  #pragma unroll
    for (int r = 0; r < MU; ++r) {
      cv[r] = 0.0;
      if (r < m) {
        const double* Xu = rec + P.rec.XuP + (long)r * DP;
        double r2 = 1.0e-300;
  #pragma unroll
        for (int k = 0; k < DP; ++k) {
          const double dlt = Xu[k] * P.cp.inv_l[k] - xs[k];
          r2 = fma(dlt, dlt, r2);
        }
        double R = P.cp.alpha * (radial_base<COV>(r2, etab) - sw[r]);
  #pragma unroll
        for (int c = 0; c < MU; ++c)
          if (c < r) R -= Lsm[r + c * m] * cv[c];
        cv[r] = R / Lsm[r + r * m];
        P.C[w * m + r] = cv[r];
      }
    }
  }
};
template <int DP, int MU, int COV>
__global__ __launch_bounds__(256) void kg_fused_c_kernel(KgTailParams P, const double* __restrict__ SWpart, int slices) {
  kg_fused_c_kernel_body<DP, MU, COV>::run(MOE_VBLOCK, MOE_VGRID, nullptr, P, SWpart, slices);
}

// thread = training point: TBpart[e][chunk][c][row] = sum over the chunk's samples of K(X_row, x*_i) beta_i[c]
template <int DP, int MU, int COV>
struct kg_fused_point_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const KgTailParams& P, const double* __restrict__ X, int n) {
    __shared__ double etab[kExpTabLen];
    __shared__ double St[kFusedTile][DP];
    __shared__ double Bt[kFusedTile][MU];
    const int row = blockIdx.x * 256 + threadIdx.x;
    const int chunk = blockIdx.y, e = blockIdx.z;
    const int m = P.m;
    const int i0 = chunk * P.chunk_len, cnt = min(P.num_local - i0, P.chunk_len);  // (chunk_len <= kFusedTile)
    if (threadIdx.x < kExpTabLen) etab[threadIdx.x] = kExp2Tab64[threadIdx.x];
    for (int t = threadIdx.x; t < kFusedTile * DP; t += 256) {
      const int ii = t / DP, k = t % DP;
      St[ii][k] = (ii < cnt) ? P.best_point[((long)e * P.num_local + i0 + ii) * DP + k] * P.cp.inv_l[k] : 0.0;
    }
    for (int t = threadIdx.x; t < kFusedTile * MU; t += 256) {
      const int ii = t / MU, c = t % MU;
      Bt[ii][c] = (ii < cnt && c < m) ? P.beta[((long)e * P.num_local + i0 + ii) * m + c] : 0.0;
    }
    __syncthreads();
    const bool ok = row < n;
    double xr[DP], acc[MU];
  #pragma unroll
    for (int k = 0; k < DP; ++k) xr[k] = ok ? X[(long)row * DP + k] * P.cp.inv_l[k] : 0.0;
  #pragma unroll
    for (int c = 0; c < MU; ++c) acc[c] = 0.0;
  #pragma unroll 2
    for (int ii = 0; ii < cnt; ++ii) {
      double r2 = 1.0e-300;
  #pragma unroll
      for (int k = 0; k < DP; ++k) {
        const double dlt = xr[k] - St[ii][k];
        r2 = fma(dlt, dlt, r2);
      }
      const double t = radial_base<COV>(r2, etab);
  #pragma unroll
      for (int c = 0; c < MU; ++c) acc[c] = fma(t, Bt[ii][c], acc[c]);
    }
    if (ok) {
      double* dst = P.TBpart + ((long)e * P.chunks + chunk) * m * P.N;
  #pragma unroll
      for (int c = 0; c < MU; ++c)
        if (c < m) dst[(long)c * P.N + row] = P.cp.alpha * acc[c];
    }
  }
};
template <int DP, int MU, int COV>
__global__ __launch_bounds__(256) void kg_fused_point_kernel(KgTailParams P, const double* __restrict__ X, int n) {
  kg_fused_point_kernel_body<DP, MU, COV>::run(MOE_VBLOCK, MOE_VGRID, nullptr, P, X, n);
}

// Both contractions of T in ONE pass (r6; m <= 4, d <= 8: the headline): every entry K(X_row, x*_i) is computed once and used for
// the TB partial of its point AND the S_W partial of its sample (the two kernels above compute every entry twice).  A wavefront
// is an 8 x 8 grid -- lane = 8 a + b: point slot a, sample slot b -- that walks the chunk's 128 samples eight at a time (si; the
// sample's coordinates and beta in registers) against its 32 points eight at a time (pi; coordinates and the row of W from LDS):
// the S_W sums of a sample live across the pi loop and are added up over the eight point slots once per si; the TB sums of the
// 4 x 8 points live in registers (4 MU doubles per lane) across the whole walk and are added up over the eight sample slots once
// per wavefront.  The eight wavefronts' S_W sums meet in LDS and are added in wavefront order: SWpart[e][i][point block][c], one
// partial per 256 points, summed in block order by kg_fused_c_kernel.  Every order is fixed and a function of (n, num_local)
// alone: an evaluation's bits do not depend on its batch.
// Measured at C3 (64 evaluations per launch): 1.71 ms against 0.93 + 0.89 ms of the two kernels on the same box, not the 0.95 the
// instruction count promises (46 VALU instructions per 64 entries): with both operands varying over the lanes every entry costs
// 7.5 ds_read_b128 where the two-kernel forms read one broadcast row, and the LDS pipe (1.05 ms) and the vector ALU (0.81 ms)
// overlap badly at four wavefronts per SIMD.  Forms tried and dropped (profiles/r06_aj_*): 64 points per wavefront on four
// wavefronts (two per SIMD: 1.68 ms), two samples per lane against one point read (half the LDS traffic, 256 VGPRs: 1.99 ms),
// the wavefront's points held in registers (no LDS reads in the loop, two wavefronts per SIMD: 2.05 ms).
template <int DP, int MU, int COV>
struct kg_fused_pair_kernel_body {
  static constexpr int SB = kFusedChunk / 8;  // sample blocks of eight
  static constexpr int SROW = DP + 2;         // row stride of the staged points: eight rows 16-byte aligned on distinct banks
  static constexpr int NW = 8, PB = 4;        // wavefronts per workgroup, point blocks of eight per wavefront: 8 x 32 = 256 points
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const KgTailParams& P, const double* __restrict__ X, int n, double* __restrict__ SWpart) {
    static_assert(MU <= 4 && kFusedChunk == 128, "4 point blocks x MU sums per lane");
    __shared__ double etab[kExpTabLen];
    __shared__ __attribute__((aligned(16))) double St[kFusedChunk][SROW];
    __shared__ __attribute__((aligned(16))) double Bt[kFusedChunk][MU];
    __shared__ __attribute__((aligned(16))) double Xt[256][SROW];
    __shared__ __attribute__((aligned(16))) double Wt[256][MU];
    __shared__ double Sx[NW][kFusedChunk][MU];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int a = lane >> 3, b = lane & 7;
    const int chunk = blockIdx.y, e = blockIdx.z, m = P.m, N = P.N;
    const int i0 = chunk * P.chunk_len, cnt = min(P.num_local - i0, P.chunk_len);  // (chunk_len == kFusedChunk)
    const long w0 = (long)e * P.num_local + i0;
    const int row0 = blockIdx.x * 256;
    if (threadIdx.x < kExpTabLen) etab[threadIdx.x] = kExp2Tab64[threadIdx.x];
    for (int t = threadIdx.x; t < kFusedChunk * DP; t += 64 * NW) {
      const int ii = t / DP, k = t % DP;
      St[ii][k] = (ii < cnt) ? P.best_point[(w0 + ii) * DP + k] * P.cp.inv_l[k] : 0.0;
    }
    for (int t = threadIdx.x; t < kFusedChunk * MU; t += 64 * NW) {
      const int ii = t / MU, c = t % MU;
      Bt[ii][c] = (ii < cnt && c < m) ? P.beta[(w0 + ii) * m + c] : 0.0;  // zero beta: padded samples drop out of TB
    }
    for (int t = threadIdx.x; t < 256 * DP; t += 64 * NW) {
      const int jj = t / DP, k = t % DP;
      Xt[jj][k] = (row0 + jj < n) ? X[(long)(row0 + jj) * DP + k] * P.cp.inv_l[k] : 0.0;
    }
    {
      const double* We = P.W + (long)e * P.w_stride;
      for (int t = threadIdx.x; t < 256 * MU; t += 64 * NW) {
        const int jj = t % 256, c = t / 256;  // W_e is [N x m] col-major: consecutive jj are contiguous
        Wt[jj][c] = (row0 + jj < n && c < m) ? We[(long)c * N + row0 + jj] : 0.0;  // zero weight: padded points drop out of S_W
      }
    }
    __syncthreads();
    double tb[PB][MU];
  #pragma unroll
    for (int pi = 0; pi < PB; ++pi)
  #pragma unroll
      for (int c = 0; c < MU; ++c) tb[pi][c] = 0.0;
    const int jw = wave * (8 * PB) + a;
  #pragma unroll 1
    for (int si = 0; si < SB; ++si) {
      const int ii = si * 8 + b;
      int hold = 0;
      asm volatile("" : "+v"(hold));  // (an index the optimiser cannot see through: the point tile is READ per sample block -- kept in registers, 4 x 8 points
                                      //  cost 80 VGPRs and with them half the wavefronts: 1.71 -> 2.05 ms per 64 evaluations at C3)
      double xs[DP], bs[MU], sw[MU];
  #pragma unroll
      for (int k = 0; k < DP; ++k) xs[k] = St[ii][k];
  #pragma unroll
      for (int c = 0; c < MU; ++c) {
        bs[c] = Bt[ii][c];
        sw[c] = 0.0;
      }
  #pragma unroll
      for (int pi = 0; pi < PB; ++pi) {
        const int jj = jw + pi * 8 + hold;
        double r2 = 1.0e-300;
  #pragma unroll
        for (int k = 0; k < DP; ++k) {
          const double dlt = Xt[jj][k] - xs[k];
          r2 = fma(dlt, dlt, r2);
        }
        const double t = radial_base<COV>(r2, etab);
  #pragma unroll
        for (int c = 0; c < MU; ++c) {
          tb[pi][c] = fma(t, bs[c], tb[pi][c]);
          sw[c] = fma(Wt[jj][c], t, sw[c]);
        }
      }
      // S_W partial of sample ii over this wavefront's 32 points: the eight point slots added up (lanes b, b + 8, ...)
  #pragma unroll
      for (int c = 0; c < MU; ++c) {
        double v = sw[c];
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (a == 0) Sx[wave][ii][c] = v;
      }
    }
    // TB partial of point jw + 8 pi: the eight sample slots added up (lanes 8 a .. 8 a + 7)
    double* dst = P.TBpart + ((long)e * P.chunks + chunk) * m * N;
  #pragma unroll
    for (int pi = 0; pi < PB; ++pi)
  #pragma unroll
      for (int c = 0; c < MU; ++c) {
        double v = tb[pi][c];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        const int row = row0 + jw + pi * 8;
        if (b == 0 && c < m && row < n) dst[(long)c * N + row] = P.cp.alpha * v;
      }
    __syncthreads();
    const int slices = gridDim.x;
    for (int t = threadIdx.x; t < kFusedChunk * MU; t += 64 * NW) {
      const int ii = t / MU, c = t % MU;
      if (ii < cnt) {
        double v = Sx[0][ii][c];
  #pragma unroll
        for (int wv = 1; wv < NW; ++wv) v += Sx[wv][ii][c];
        SWpart[((w0 + ii) * slices + blockIdx.x) * MU + c] = v;
      }
    }
  }
};
template <int DP, int MU, int COV>
__global__ __launch_bounds__(512, 2) void kg_fused_pair_kernel(KgTailParams P, const double* __restrict__ X, int n, double* __restrict__ SWpart) {
  kg_fused_pair_kernel_body<DP, MU, COV>::run(MOE_VBLOCK, MOE_VGRID, nullptr, P, X, n, SWpart);
}

// whether an evaluation's T-free tail takes the one-pass kernel (a function of its shape alone), and how many S_W partials a sample then has
inline bool fused_tail_one_pass(int m, int dp) { return m <= 4 && dp <= 8 && env_int("MOE_KG_FUSED_ONE_PASS", 1) != 0; }

template <int DP, int MU, int COV>
void launch_fused_tail_cov(const KgTailParams& P, const double* X, int n, double* SWpart, int slices, hipStream_t s) {
  dim3 ga((P.num_local + 255) / 256, P.E, slices), gc((P.num_local + 255) / 256, P.E), gb((n + 255) / 256, P.chunks, P.E);
  if constexpr (MU <= 4 && DP <= 8) {
    if (fused_tail_one_pass(P.m, DP)) {  // (slices == gb.x: fused_tail_slices)
      launch_kernel_ens<kg_fused_pair_kernel_body<DP, MU, COV>, 512, 2>(kg_fused_pair_kernel<DP, MU, COV>, gb, dim3(512), 0, s, P, X, n, SWpart);
      launch_kernel_ens<kg_fused_c_kernel_body<DP, MU, COV>, 256>(kg_fused_c_kernel<DP, MU, COV>, gc, dim3(256), 0, s, P, SWpart, (int)gb.x);
      launch_dir<DP>(P, s);
      launch_gtb(P, s);
      MOE_HIP_CHECK(hipGetLastError());
      return;
    }
  }
  launch_kernel_ens<kg_fused_sample_kernel_body<DP, MU, COV>, 256>(kg_fused_sample_kernel<DP, MU, COV>, ga, dim3(256), 0, s, P, X, n, SWpart);
  launch_kernel_ens<kg_fused_c_kernel_body<DP, MU, COV>, 256>(kg_fused_c_kernel<DP, MU, COV>, gc, dim3(256), 0, s, P, SWpart, slices);
  launch_dir<DP>(P, s);
  launch_kernel_ens<kg_fused_point_kernel_body<DP, MU, COV>, 256>(kg_fused_point_kernel<DP, MU, COV>, gb, dim3(256), 0, s, P, X, n);
  launch_gtb(P, s);
  MOE_HIP_CHECK(hipGetLastError());
}

template <int DP, int MU>
void launch_fused_tail_inst(const KgTailParams& P, const double* X, int n, double* SWpart, int slices, hipStream_t s) {
  if (P.cp.type == MOE_COV_SQUARE_EXPONENTIAL)
    launch_fused_tail_cov<DP, MU, MOE_COV_SQUARE_EXPONENTIAL>(P, X, n, SWpart, slices, s);
  else
    launch_fused_tail_cov<DP, MU, MOE_COV_MATERN_NU_2P5>(P, X, n, SWpart, slices, s);
}

// number of point slices of kg_fused_sample_kernel: enough wavefronts for ~4 per SIMD when ONE evaluation runs alone.  It does not
// depend on the batch size: the slices are summed in order, so an evaluation's bits would otherwise change with its batch.
int fused_tail_slices(int /*E*/, int num_local, int n, int num_cu, int m, int dp) {
  if (fused_tail_one_pass(m, dp)) return (n + 255) / 256;  // the one-pass kernel: one S_W partial per block of 256 points
  const long waves = (long)((num_local + 255) / 256) * 4;
  const long want = (long)num_cu * 4 * 4;
  int s = (int)std::min<long>(8, std::max<long>(1, (want + waves - 1) / waves));
  return std::max(1, std::min(s, (n + 63) / 64));
}

// T-free tail (requires g == 0, m <= 8, chunk_len <= kFusedTile)
void launch_fused_tail(const KgTailParams& P, const double* X, int n, double* SWpart, int slices, hipStream_t s) {
  static_assert(kFusedChunk <= kFusedTile, "a chunk of samples is staged in one LDS tile");
  const bool m4 = P.m <= 4;
  switch (P.cp.dp) {
    case 4: m4 ? launch_fused_tail_inst<4, 4>(P, X, n, SWpart, slices, s) : launch_fused_tail_inst<4, 8>(P, X, n, SWpart, slices, s); break;
    case 8: m4 ? launch_fused_tail_inst<8, 4>(P, X, n, SWpart, slices, s) : launch_fused_tail_inst<8, 8>(P, X, n, SWpart, slices, s); break;
    case 12: m4 ? launch_fused_tail_inst<12, 4>(P, X, n, SWpart, slices, s) : launch_fused_tail_inst<12, 8>(P, X, n, SWpart, slices, s); break;
    case 16: m4 ? launch_fused_tail_inst<16, 4>(P, X, n, SWpart, slices, s) : launch_fused_tail_inst<16, 8>(P, X, n, SWpart, slices, s); break;
    case 24: m4 ? launch_fused_tail_inst<24, 4>(P, X, n, SWpart, slices, s) : launch_fused_tail_inst<24, 8>(P, X, n, SWpart, slices, s); break;
    case 32: m4 ? launch_fused_tail_inst<32, 4>(P, X, n, SWpart, slices, s) : launch_fused_tail_inst<32, 8>(P, X, n, SWpart, slices, s); break;
    default: throw Error(MOE_ERR_RUNTIME, "unsupported padded dimension");
  }
}

// value-only finish: kg_sum per evaluation (same summation as block 0 of kg_zc_sum_kernel)
struct kg_sum_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const KgTailParams& P, double* __restrict__ fin, const unsigned long long* __restrict__ counters, const int* __restrict__ flags) {
    __shared__ double red[4];
    const int e = blockIdx.x;
    const double bp = P.blob[(long)e * P.rec.stride + P.rec_bp];
    double acc = 0.0;
    for (int i = threadIdx.x; i < P.num_local; i += 256) acc += bp + P.best_value[(long)e * P.num_local + i];
    const double tot = block_sum_256(acc, red);
    if (threadIdx.x == 0) {  // the record the host reads back: kg_sum | value passes | gradient passes | singular flag
      fin[4 * e] = tot;
      fin[4 * e + 1] = (double)counters[2 * e];
      fin[4 * e + 2] = (double)counters[2 * e + 1];
      fin[4 * e + 3] = (double)flags[e];
    }
  }
};
__global__ __launch_bounds__(256) void kg_sum_kernel(KgTailParams P, double* __restrict__ fin, const unsigned long long* __restrict__ counters,
                                                     const int* __restrict__ flags) {
  kg_sum_kernel_body::run(MOE_VBLOCK, MOE_VGRID, nullptr, P, fin, counters, flags);
}

struct EventTimer {
  hipEvent_t a = nullptr, b = nullptr;  // (created on first use: a recorded evaluation -- launch.hpp -- times nothing)
  bool on = false;
  EventTimer() = default;
  EventTimer(const EventTimer&) = delete;
  EventTimer& operator=(const EventTimer&) = delete;
  ~EventTimer() {
    if (a != nullptr) (void)hipEventDestroy(a);
    if (b != nullptr) (void)hipEventDestroy(b);
  }
  void start(hipStream_t s) {
    on = Recorder::current() == nullptr;
    if (!on) return;
    if (a == nullptr) {
      MOE_HIP_CHECK(hipEventCreate(&a));
      MOE_HIP_CHECK(hipEventCreate(&b));
    }
    MOE_HIP_CHECK(hipEventRecord(a, s));
  }
  void stop(hipStream_t s) {
    if (on) MOE_HIP_CHECK(hipEventRecord(b, s));
  }
  double ms() {
    if (!on) return 0.0;
    MOE_HIP_CHECK(hipEventSynchronize(b));
    float t = 0.f;
    MOE_HIP_CHECK(hipEventElapsedTime(&t, a, b));
    return t;
  }
};

int env_int(const char* name, int dflt) {
  const char* v = std::getenv(name);
  return (v && *v) ? std::atoi(v) : dflt;
}

// z, beta = L^-T z and the discretised-set winner of EVERY sample (they depend on the normal draws alone): one wavefront
// per sample, grid-stride, the same device functions the MC kernels use -- so the values are the ones they would compute.
struct kg_sample_prep_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const KgMcParams& P, int* __restrict__ best_j) {
    __shared__ double zbs[4][2 * kMaxM];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* zb = zbs[wave];
    const long total = (long)P.E * P.num_local;
    for (long idx = (long)blockIdx.x * 4 + wave; idx < total; idx += (long)gridDim.x * 4) {
      const int e = (int)(idx / P.num_local), sl = (int)(idx % P.num_local);
      const double* rec = P.blob + (long)e * P.rec.stride;
      double zc, bc;
      mc::draw_z_beta(P, rec + P.rec.L, P.first_sample + sl, lane, zb, zc, bc);
      const int bj = mc::discrete_scan(P, rec, zb, lane);
      if (lane < P.m) P.beta[idx * P.m + lane] = bc;
      if (lane == 0) best_j[idx] = bj;
      __builtin_amdgcn_wave_barrier();
    }
  }
};
__global__ __launch_bounds__(256) void kg_sample_prep_kernel(KgMcParams P, int* __restrict__ best_j) {
  kg_sample_prep_kernel_body::run(MOE_VBLOCK, MOE_VGRID, nullptr, P, best_j);
}

// The same for m > 64 (more components than lanes): one THREAD per sample, serial O(m^2) back substitution and O(A m) scan
// against operands in L2 -- a few hundred microseconds for 2 x 10^4 samples at m = 104; only the workgroup-per-sample kernel
// consumes it.  z_i (antithetic pairs, .cpp:171-180), beta_i = L^-T z_i, first best discretised point (.cpp:436-449).
struct kg_sample_prep_generic_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const KgMcParams& P, int* __restrict__ best_j) {
    const long idx = (long)blockIdx.x * 64 + threadIdx.x;
    const long total = (long)P.E * P.num_local;
    if (idx >= total) return;
    const int e = (int)(idx / P.num_local), sl = (int)(idx % P.num_local);
    const int m = P.m, s = P.first_sample + sl;
    const double* rec = P.blob + (long)e * P.rec.stride;
    const double* L = rec + P.rec.L;
    const double sign = (s & 1) ? -1.0 : 1.0;
    const double* zrow = P.normals + (long)(s >> 1) * m;
    double beta[kMaxMB];
    for (int r = m - 1; r >= 0; --r) {
      double acc = 0.0;
      for (int l = m - 1; l > r; --l) acc = fma(L[l + (long)r * m], beta[l], acc);
      beta[r] = (sign * zrow[r] - acc) / L[r + (long)r * m];
    }
    for (int c = 0; c < m; ++c) P.beta[idx * m + c] = beta[c];
    const double* mu_disc = rec + P.rec.mu_disc;
    const double* C_disc = rec + P.rec.C_disc;
    double best_f = -INFINITY;
    int bj = 0;
    for (int j = 0; j < P.A; ++j) {
      double v = mu_disc[j];
      for (int c = 0; c < m; ++c) v = fma(C_disc[(long)j * m + c], sign * zrow[c], v);
      if (-v > best_f) {  // strict: the first best point wins
        best_f = -v;
        bj = j;
      }
    }
    best_j[idx] = bj;
  }
};
__global__ __launch_bounds__(64) void kg_sample_prep_generic_kernel(KgMcParams P, int* __restrict__ best_j) {
  kg_sample_prep_generic_kernel_body::run(MOE_VBLOCK, MOE_VGRID, nullptr, P, best_j);
}

// Per-sample weights of the training rows for every sample, V[(e, sl)][r] = scale_a (KinvY[r] - sum_c W_e[r, c] beta[(e, sl), c])
// (r = (j, a); scale_0 = alpha, scale_a = -alpha / l_{d_a}): what point_weights computes inside the MC kernel, same
// operation order, hence the same bits.  Inside the workgroup-per-sample kernel that computation reads all of W_e
// (N x m) per sample and CU through a few dozen loads in flight -- a sixth of the kernel at m = 16; here one thread keeps
// a row of W in registers, beta arrives through scalar loads, and the chip writes N x M doubles at streaming speed.
// m > 64 (r4): the columns go down in passes of 64 -- pass [c_lo, c_lo + MB) continues the fma chain from the partial sum the previous
// pass left in V (plain store), the last pass scales and streams the result out; one pass (first = last) is the code of m <= 64.
template <int MB>
struct kg_sample_weights_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const KgMcParams& P, double* __restrict__ V, int samples_per_block, int c_lo, int first, int last) {
    // table entry t = (point j, slot a) <-> row r = j (1 + g) + a of the N training rows / the m fantasy rows; slots beyond the GP's
    // 1 + g (r4: a streamed-weights instantiation with more derivative slots than observed derivatives) hold zeros
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int e = blockIdx.z;
    const int m = P.m, g1 = 1 + P.g;
    const int pj = t / P.v_slots1, pa = t - pj * P.v_slots1;
    const bool slot_ok = pa < g1;
    const int r = slot_ok ? pj * g1 + pa : P.N + m;  // (an unused slot: behind everything, stored as 0)
    const double* __restrict__ We = P.W + (long)e * P.w_stride;
    const double* __restrict__ beta = P.beta;
    double l[MB];
    const int rr = min(r, P.N - 1);
  #pragma unroll
    for (int c = 0; c < MB; ++c) {  // zero beyond m: the unconditional fma below then leaves v untouched
      const double t = We[rr + (long)min(c_lo + c, m - 1) * P.N];
      l[c] = (c_lo + c < m) ? t : 0.0;
    }
    const double kiy = P.KinvY[rr];
    const int a = rr % g1;
    const double scale = (a == 0) ? P.alpha : -P.alpha * P.inv_lp[a > 0 ? a - 1 : 0];
    const int s0 = blockIdx.y * samples_per_block, s1 = min(P.num_local, s0 + samples_per_block);
    // rows behind the training set (v_stride > N: the streamed-weights MC kernel reads whole tiles): the fantasy points' weights are the
    // sample's beta itself, scaled like a training row (kg_mc.hpp kg_sample), then zeros up to the end of the last tile
    const int cf = r - P.N;
    const bool fantasy = cf >= 0 && cf < m;
    const double fscale = ((cf % g1) == 0) ? P.alpha : -P.alpha * P.inv_lp[max(cf % g1, 1) - 1];
    for (int sl = s0; sl < s1; ++sl) {
      const long so = (long)e * P.num_local + sl;
      const double* __restrict__ bs = beta + so * m + c_lo;
      double v = first ? kiy : V[so * P.v_stride + min(t, (int)P.v_stride - 1)];
  #pragma unroll
      for (int c = 0; c < MB; ++c) v = fma(-l[c], bs[c], v);  // uniform, contiguous: wide scalar loads (reads up to MB - m
                                                              // doubles past the row: next rows / the zeroed pad, times l = 0)
      if (t >= P.v_stride) continue;
      if (r < P.N) {
        if (last)
          __builtin_nontemporal_store(v * scale, &V[so * P.v_stride + t]);  // (streaming: 1.28 GB at C5, read once by the MC kernel)
        else
          V[so * P.v_stride + t] = v;
      } else if (last) {
        V[so * P.v_stride + t] = fantasy ? beta[so * m + min(cf, m - 1)] * fscale : 0.0;
      }
    }
  }
};
template <int MB>
__global__ __launch_bounds__(256) void kg_sample_weights_kernel(KgMcParams P, double* __restrict__ V, int samples_per_block,
                                                               int c_lo = 0, int first = 1, int last = 1) {
  kg_sample_weights_kernel_body<MB>::run(MOE_VBLOCK, MOE_VGRID, nullptr, P, V, samples_per_block, c_lo, first, last);
}

// The same table for m > 64 (r4; the stretch point's m = 104) with its sums on the matrix pipe: V_e (table rows x samples) = K^-1 y -
// W_e beta_e^T through gemm128.hpp's tile core, scaled and written ONCE (the kernel above needs two passes over a partially written
// table there: 2 x 5.1 ms per evaluation against the MC kernel's own 9.4).  Needs the table's rows to BE the training rows
// (v_slots1 == 1 + g).  The sums are formed in the MFMA's order, not in the c-order of the in-kernel weights: for these shapes a result
// depends at rounding level on whether the table fitted its cap (the m <= 64 paths keep their bit-for-bit equality).
// Grid (row tiles of 128 table entries, column tiles of 128 samples, E).
struct kg_table128_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const KgMcParams& P, double* __restrict__ V) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int e = blockIdx.z, m = P.m, g1 = 1 + P.g;
    const long s0 = (long)e * P.num_local;
    g128::Operand A{P.W + (long)e * P.w_stride, (long)P.N, P.N, m, 1};   // W[t + c N]: table rows contiguous
    g128::Operand B{P.beta + s0 * m, (long)m, P.num_local, m, 1};        // beta[sample m + c]: K (= c) contiguous
    g128::f64x4 acc[4][4];
    const int t0 = blockIdx.x * g128::TM, j0 = blockIdx.y * g128::TM;
    if (t0 < P.N) g128::tile_product<false, true>(A, B, t0, j0, 0, m, smem, acc);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wi = (wave & 1) * 64, wj = (wave >> 1) * 64;
    const int lk = lane >> 4, lx = lane & 15;
  #pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int t = t0 + wi + 16 * a + lx;
      const int tt = min(t, P.N - 1);
      const double kiy = P.KinvY[tt];
      const int sa = tt % g1;
      const double scale = (sa == 0) ? P.alpha : -P.alpha * P.inv_lp[sa > 0 ? sa - 1 : 0];
      // rows behind the training set: the fantasy points' weights are the sample's beta itself, scaled like a training row, then zeros
      const int cf = t - P.N;
      const bool fantasy = cf >= 0 && cf < m;
      const double fscale = ((cf % g1) == 0) ? P.alpha : -P.alpha * P.inv_lp[max(cf % g1, 1) - 1];
  #pragma unroll
      for (int b = 0; b < 4; ++b)
  #pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int smp = j0 + wj + 16 * b + lk + 4 * r;
          if (t < P.v_stride && smp < P.num_local) {
            const long so = s0 + smp;
            double v;
            if (t < P.N)
              v = (kiy - acc[a][b][r]) * scale;
            else
              v = fantasy ? P.beta[so * m + min(cf, m - 1)] * fscale : 0.0;
            __builtin_nontemporal_store(v, &V[so * P.v_stride + t]);
          }
        }
    }
  }
};
__global__ __launch_bounds__(256, 2) void kg_table128_kernel(KgMcParams P, double* __restrict__ V) {
  kg_table128_kernel_body::run(MOE_VBLOCK, MOE_VGRID, nullptr, P, V);
}

void launch_sample_weights(const KgMcParams& P, double* V, hipStream_t s, bool table_only = false) {
  // r6: `table_only` -- the consumer (the streamed-weights kernel) has no in-kernel form of the weights whose bits the table would have
  // to reproduce -- takes the matrix-pipe kernel from m = 16 on: per (row, sample) entry the fma version's loop is one dependent chain
  // of m fmas behind a scalar-load round trip and the previous store's acknowledgement (1.17 ms per 2.56 GB at C5 = 2.2 TB/s)
  const bool mfma = P.v_slots1 == 1 + P.g && (P.m > 64 || (table_only && P.m >= 16 && env_int("MOE_KG_TABLE_MFMA", 1) != 0));
  if (mfma) {
    MOE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kg_table128_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)g128::kSmemBytes));
    launch_kernel_ens<kg_table128_kernel_body, 256, 2>(kg_table128_kernel, dim3((unsigned)((P.v_stride + g128::TM - 1) / g128::TM), (P.num_local + g128::TM - 1) / g128::TM, P.E),
                       dim3(256), g128::kSmemBytes, s, P, V);
    MOE_HIP_CHECK(hipGetLastError());
    return;
  }
  const int spb = 64;
  dim3 grid((unsigned)((P.v_stride + 255) / 256), (P.num_local + spb - 1) / spb, P.E);
  if (P.m <= 16)
    launch_kernel_ens<kg_sample_weights_kernel_body<16>, 256>(kg_sample_weights_kernel<16>, grid, dim3(256), 0, s, P, V, spb, 0, 1, 1);
  else if (P.m <= 32)
    launch_kernel_ens<kg_sample_weights_kernel_body<32>, 256>(kg_sample_weights_kernel<32>, grid, dim3(256), 0, s, P, V, spb, 0, 1, 1);
  else  // (m > 64 in passes of 64 columns; ONE pass with a 128-column row of W in registers -- 256 of them, half in the accumulation file --
        //  measured slower: 34 against 21 ms of MC phase per evaluation at the stretch point, m = 104)
    for (int c_lo = 0; c_lo < P.m; c_lo += 64)
      launch_kernel_ens<kg_sample_weights_kernel_body<64>, 256>(kg_sample_weights_kernel<64>, grid, dim3(256), 0, s, P, V, spb, c_lo, c_lo == 0 ? 1 : 0, c_lo + 64 >= P.m ? 1 : 0);
  MOE_HIP_CHECK(hipGetLastError());
}

void launch_mc(const KgMcParams& P, int dp, int G, bool xlds, int blocks, int waves, size_t shm, hipStream_t s) {
  switch (dp) {
    case 4: launch_kg_mc_dp4(P, G, xlds, blocks, waves, shm, s); break;
    case 8: launch_kg_mc_dp8(P, G, xlds, blocks, waves, shm, s); break;
    case 12: launch_kg_mc_dp12(P, G, xlds, blocks, waves, shm, s); break;
    case 16: launch_kg_mc_dp16(P, G, xlds, blocks, waves, shm, s); break;
    case 24: launch_kg_mc_dp24(P, G, xlds, blocks, waves, shm, s); break;
    case 32: launch_kg_mc_dp32(P, G, xlds, blocks, waves, shm, s); break;
    default: throw Error(MOE_ERR_RUNTIME, "unsupported padded dimension");
  }
}

void launch_mc_lane(const KgMcParams& P, int dp, int G, int rec_head, int blocks, int waves, size_t shm, hipStream_t s) {
  switch (dp) {
    case 4: launch_kg_mc_lane_dp4(P, G, rec_head, blocks, waves, shm, s); break;
    case 8: launch_kg_mc_lane_dp8(P, G, rec_head, blocks, waves, shm, s); break;
    case 12: launch_kg_mc_lane_dp12(P, G, rec_head, blocks, waves, shm, s); break;
    case 16: launch_kg_mc_lane_dp16(P, G, rec_head, blocks, waves, shm, s); break;
    default: throw Error(MOE_ERR_RUNTIME, "unsupported padded dimension in the lane-parked MC kernel");
  }
}

void launch_mc_stream(const KgMcParams& P, int dp, int G, int blocks, int waves, size_t shm, hipStream_t s) {
  switch (dp) {
    case 4: launch_kg_mc_stream_dp4(P, G, blocks, waves, shm, s); break;
    case 8: launch_kg_mc_stream_dp8(P, G, blocks, waves, shm, s); break;
    case 12: launch_kg_mc_stream_dp12(P, G, blocks, waves, shm, s); break;
    case 16: launch_kg_mc_stream_dp16(P, G, blocks, waves, shm, s); break;
    case 24: launch_kg_mc_stream_dp24(P, G, blocks, waves, shm, s); break;
    case 32: launch_kg_mc_stream_dp32(P, G, blocks, waves, shm, s); break;
    default: throw Error(MOE_ERR_RUNTIME, "unsupported padded dimension");
  }
}

void launch_mc_block(const KgMcParams& P, int dp, int G, int tr, int num_lds_tiles, int blocks, int waves, hipStream_t s) {
  switch (dp) {
    case 4: launch_kg_mc_block_dp4(P, G, tr, num_lds_tiles, blocks, waves, s); break;
    case 8: launch_kg_mc_block_dp8(P, G, tr, num_lds_tiles, blocks, waves, s); break;
    case 12: launch_kg_mc_block_dp12(P, G, tr, num_lds_tiles, blocks, waves, s); break;
    case 16: launch_kg_mc_block_dp16(P, G, tr, num_lds_tiles, blocks, waves, s); break;
    case 24: launch_kg_mc_block_dp24(P, G, tr, num_lds_tiles, blocks, waves, s); break;
    case 32: launch_kg_mc_block_dp32(P, G, tr, num_lds_tiles, blocks, waves, s); break;
    default: throw Error(MOE_ERR_RUNTIME, "unsupported padded dimension");
  }
}

}  // namespace

// Everything up to and including the asynchronous launches on gp.stream; the returned object's collect() waits for the
// stream and assembles the results.  Launching on several GPs (MCMC ensemble members, each with its own stream and
// workspaces) before collecting any of them overlaps one member's host algebra with the others' kernels.
KgPending kg_launch(GpDev& gp, int num_fidelity, const moe_gd_params_t& gd, const double* bounds, const double* discrete, int P,
                    const double* Xq_all, int num_evals, const double* Xp, int q, int p, int num_mc, double best_so_far,
                    const double* normals, int first_sample, int num_local, bool want_grad, bool want_best_points,
                    double weight_table_gb, const double* disc_head) {
  gp.use_device();
  hipStream_t s = gp.stream;
  const int d = gp.d, dp = gp.dp, f = num_fidelity, u = q + p, n = gp.n, g = gp.g, g1 = 1 + gp.g, N = gp.N;
  const int E = num_evals;
  const int m = u * g1;
  if (q <= 0) throw Error(MOE_ERR_BOUNDS, "num_to_sample must be positive", q, 1, 1e9);
  if (p < 0) throw Error(MOE_ERR_BOUNDS, "num_being_sampled must be non-negative", p, 0, 1e9);
  if (E <= 0) throw Error(MOE_ERR_BOUNDS, "num_evals must be positive", E, 1, 1e9);
  if (m > kMaxMB)
    throw Error(MOE_ERR_BOUNDS, "(q + p)(1 + num_derivatives) > 128 is not supported by the device kernels", m, 1, kMaxMB);
  if (g > 12) throw Error(MOE_ERR_BOUNDS, "d-KG with more than 12 observed derivatives is not supported by the device kernels", g, 0, 12);
  if (f < 0 || f >= d) throw Error(MOE_ERR_BOUNDS, "num_fidelity out of range", f, 0, d - 1);
  if (num_mc <= 0) throw Error(MOE_ERR_BOUNDS, "num_mc must be positive", num_mc, 1, 1e12);
  if (first_sample < 0 || (first_sample & 1) || num_local <= 0 || first_sample + num_local > num_mc)
    throw Error(MOE_ERR_INVALID_VALUE, "MC shard must be an even-aligned slice of [0, num_mc)", first_sample, 0, 0);
  if (gd.max_num_steps <= 0) throw Error(MOE_ERR_BOUNDS, "max_num_steps must be positive", gd.max_num_steps, 1, 1e9);
  const int size = d - f;
  if (gd.domain_type != MOE_DOMAIN_TENSOR_PRODUCT && gd.domain_type != MOE_DOMAIN_SIMPLEX)
    throw Error(MOE_ERR_INVALID_VALUE, "unknown domain_type (0 = tensor product, 1 = simplex)", gd.domain_type, 0, 1);
  const bool simplex = gd.domain_type == MOE_DOMAIN_SIMPLEX;  // the inner optimisations' domain (r4)
  const int A = u + P;
  // derivative-weight slots of the MC kernel instantiation: one per observed derivative up to 4, then 8 or 12 (unused slots
  // carry zero weights); more than 4 always take the workgroup-per-sample kernel
  // (d > 16: slot counts {0, 4, 8, 12} only -- kg_mc.hpp launch_dp_wide)
  const bool wide_dp = dp > 16;
  const int G = wide_dp ? (g == 0 ? 0 : (g <= 4 ? 4 : (g <= 8 ? 8 : 12))) : (g <= 4 ? g : (g <= 8 ? 8 : 12));
  const int ntiles = (n + u + 63) / 64;
  const int ngrad = want_grad ? q * g1 * d : 0;

  // ---- MC launch geometry: workgroup = `waves` wavefronts sharing one LDS coordinate table ----
  const size_t tab_bytes = sizeof(double) * (size_t)ntiles * (dp + 1) * 64;  // LDS copy: + the |x|^2 row (kg_mc.hpp eval_loop)
  const size_t slab_bytes = sizeof(double) * ((size_t)ntiles * (1 + G) * 64 + 2 * kMaxM);
  // (streamed coordinates: + the wave's line-search vectors and packed-sum slot -- kg_mc.hpp WideEval)
  const size_t slab_stream_bytes = slab_bytes + (mc::wide_eval(dp, false) ? sizeof(double) * mc::kWideScratch : 0);
  // the exp table sits in front of everything; one weight tile of padding at the very end (eval_loop prefetches one tile
  // past the last wave's slab)
  const size_t pad_bytes = sizeof(double) * (size_t)(1 + G) * 64;
  const size_t fixed_bytes = sizeof(double) * (kExpTabLen + (size_t)mc::kCstRows * dp);  // exp table + frame constants
  const size_t lds_max = 160 * 1024 - fixed_bytes - pad_bytes;
  bool xlds = true;
  // The value passes of the LDS-table kernel take r^2 as |x_j|^2 + |q|^2 - 2 x_j.q (kg_mc.hpp eval_loop), whose absolute error is
  // ~eps (|x|^2 + |q|^2) in the centred, scaled frame of the tables: harmless while the point set spans tens of length
  // scales, but 1e-9 .. 1e-5 of r^2 at 1e3 .. 1e5 length scales (short length scales of a hyper-parameter MCMC ensemble) --
  // enough to break the 1e-8 parity with the reference.  Beyond a frame radius of 100 length scales the direct-difference
  // kernels are used instead (coordinates streamed from L2, or the workgroup-per-sample kernel): slower, exact.
  bool wide_frame = false, far_frame = false;
  {
    // The extent is taken from what does NOT depend on the batch -- the training points (cached in the GP), the points being
    // sampled, and the DOMAIN box the restarts live in -- so that a restart takes the same kernel whether it is evaluated alone,
    // in a batch, or in another rank's shard (r3, ADVICE: moe_kg_batch_multi / dist.py promise bit-identical restarts).  Only a
    // point_to_sample outside the box (or a fidelity coordinate, which has no inner bound) adds its own distance.
    double rad2 = 0.0;
    for (int k = 0; k < d; ++k) {
      const double c = gp.x_mean[k];
      double ext = gp.x_ext[k];
      if (k < size) ext = std::max(ext, std::max(std::fabs(bounds[2 * k] - c), std::fabs(bounds[2 * k + 1] - c)));
      for (size_t i = 0; i < (size_t)E * q; ++i) {
        const double v = Xq_all[i * d + k];
        if (k >= size || v < bounds[2 * k] || v > bounds[2 * k + 1]) ext = std::max(ext, std::fabs(v - c));
      }
      for (int i = 0; i < p; ++i) ext = std::max(ext, std::fabs(Xp[(size_t)i * d + k] - c));
      rad2 += (ext * gp.cp.inv_l[k]) * (ext * gp.cp.inv_l[k]);
    }
    wide_frame = !(rad2 <= (double)env_int("MOE_KG_DOT_MAX_RADIUS2", 10000));
    far_frame = wide_frame;
    // d > 16: the coordinate table of the wave-per-sample kernel would take (dp + 1) 512 bytes per tile; that kernel is built
    // without it there, so these shapes take the same route as a wide frame
    if (wide_dp) wide_frame = true;
    if (wide_frame) xlds = false;
  }
  // one or two tiles per pass: a pass is a short dependent chain, so 16 wavefronts per workgroup (the <= 128-VGPR instantiation,
  // single-trial passes).  From three tiles on the 8-wavefront instantiation with its multi-trial passes is faster (r2: n = 150,
  // d = 6, q = 4: 19 600 vs 12 900 evals/s; n = 200: 17 700 vs 12 100; two tiles, n = 100: 47 000 vs 49 000)
  // ... and only up to four dimensions: with eight or more coordinate rows the 128-VGPR instantiation spills heavily (0.5 - 1 KB of
  // scratch per lane) -- n = 60, d = 8: 27 300 vs 44 500 evals/s on the 8-wavefront instantiation, d = 12: 9 500 vs 35 800,
  // d = 16: 6 200 vs 25 200 (r2)
  // r5: a call with FEW samples on such a shape -- one evaluation of a KG-MCMC suggestion: n = 30, 20 restarts x 128 samples; the
  // drop-in boundary's one-at-a-time calls -- takes the lane-parked kernel instead, eight wavefronts, with the same single-trial passes:
  // the same bits as the 16-wavefront instantiation (tests/test_gpu_sweep.py: test_small_shape_kernels_agree), so which of the two runs
  // may depend on the size of the call.  MC kernel of one evaluation at n = 30, M = 128: 0.087 -> 0.056 ms (M = 2000: 0.165 -> 0.132);
  // a suggestion 0.212 -> 0.18-0.19 s.  In throughput the 16-wavefront instantiation keeps its lead (batch of 64, M = 2000: 0.0200 vs
  // 0.0249 ms per evaluation), hence the sample bound.  (The lane-parked kernel's MULTI-trial passes would be faster still -- 0.17 s --
  // but their dot-form distances move the end points of the reference's 100-step x 10-restart fixtures by up to 1.01e-6, beyond the
  // 1e-6 those fixtures are held to: not taken.)  `profiles/r05_z_*`.
  const bool small_shape = ntiles <= env_int("MOE_KG_SMALL_TILES", 2) && dp <= env_int("MOE_KG_SMALL_DP", 4);
  // r6: decided from the evaluation's OWN sample count, not from the call's -- which kernel an evaluation takes never
  // depends on the batch or the rank it is evaluated in (the two kernels agree bit for bit on the tested shapes; the promise of
  // batch-independent bits no longer rests on that)
  // (a simplex inner domain: the lane-parked kernel at every sample count -- the 16-wavefront instantiation has no simplex update)
  const bool small_lane = small_shape && env_int("MOE_KG_LANE", 1) != 0 &&
                          (simplex || (long)num_local <= (long)env_int("MOE_KG_SMALL_LANE_MAX_SAMPLES", 1024));
  const int max_waves = (small_shape && !small_lane) ? 16 : 8;
  int waves = 0;
  if (tab_bytes + slab_bytes <= lds_max) waves = (int)std::min<size_t>(max_waves, (lds_max - tab_bytes) / slab_bytes);
  // (3 wavefronts per CU on the LDS table still beat the workgroup-per-sample kernel without derivative observations --
  //  n = 1500, d = 8: 2.1 vs 3.1 ms per evaluation; with 2 they lose -- n = 1700: 3.6 vs 3.4)
  const int min_xlds_waves = env_int("MOE_KG_MIN_XLDS_WAVES", 4);  // (r2: 4 -- with 3 the streaming kernel and its 8 wavefronts win, n = 1500: 1.15 vs 1.44 ms)
  if (waves < min_xlds_waves || wide_frame) {  // coordinates stay in L2: more wavefronts per workgroup fit
    const int w2 = (int)std::min<size_t>(8, lds_max / slab_stream_bytes);
    if (w2 > waves || wide_frame) {  // (the instantiation without the LDS table is built for <= 8 wavefronts)
      waves = w2;
      xlds = false;
    }
  }
  // Variant selection: the wave-per-sample kernel needs the coordinate table AND >= 3 weight slabs in LDS; bigger point
  // sets (up to 32 tiles = 2048 points) go to the workgroup-per-sample kernel (coordinates in registers, 4 waves);
  // beyond that the wave-per-sample kernel streams coordinates from L2 (slow, but correct).
  // workgroup-per-sample geometry: 8 wavefronts; TR register tiles per wave, the remaining tiles in LDS
  const int bwaves = 8;
  int tr = -1, num_lds_tiles = 0;
  for (int cand : {0, 2, 4}) {
    if (wide_dp && cand != 0) break;  // (d > 16: all tiles in LDS or the streaming wave-per-sample kernel)
    const int tl = std::max(0, ntiles - bwaves * cand);
    if (kg_mc_block_lds_bytes(dp, G, tl) <= (size_t)160 * 1024) {
      tr = cand;
      num_lds_tiles = tl;
      break;
    }
  }
  if (!wide_dp) tr = env_int("MOE_KG_TR", tr);
  if (tr >= 0) num_lds_tiles = std::max(0, ntiles - bwaves * tr);
  int variant = (xlds && waves >= min_xlds_waves) ? 0 : (tr >= 0 ? 1 : 0);
  // Without derivative observations the wave-per-sample kernel streaming its coordinates from L2 now beats the workgroup-per-sample
  // kernel at every size both can run (r2: its multi-trial passes sweep the coordinates once per five Armijo trials -- n = 1600:
  // 1.23 vs 2.68 ms per evaluation, 2000: 1.39 vs 2.67, 3000: 2.15 vs 3.21, d = 8, q = 4, M = 1e4); far frames (single-trial
  // passes) keep the old choice
  if (G == 0 && !far_frame && waves >= 1) variant = 0;
  // with derivative observations (a point's 1 + G weights make the slabs bigger) the streamed kernel wins while five of them fit:
  // d = 12, g = 3, q = 8, M = 4000: n = 800 0.86 vs 1.10 ms; with four (n = 1200) 1.28 vs 1.21
  if (G > 0 && G <= 4 && !far_frame && !xlds && waves >= 5) variant = 0;
  if (G > 4 || m > kMaxM) variant = 1;  // (the wave-per-sample kernel: up to four derivative slots, one lane per component)
  // Variant 2, the streamed-weights wave-per-sample kernel (kg_mc.hpp kg_mc_stream_kernel), takes what would go to the
  // workgroup-per-sample kernel whenever the per-sample weight table (fantasy points and tile padding included) fits its cap and
  // every derivative slot is an observed derivative (a point's table rows ARE its weights)
  const int prep_mode = env_int("MOE_KG_PREP", -1);
  const long v_stride_tiles = (long)ntiles * 64 * (1 + G);  // (the table of the streamed-weights kernel: 1 + G slots per point)
  const double v_cap = std::getenv("MOE_KG_V_MAX_GB") ? (double)env_int("MOE_KG_V_MAX_GB", 4)
                                                      : (weight_table_gb >= 0.0 ? weight_table_gb : 4.0);
  // (the size test is per EVALUATION: which kernel an evaluation takes must not depend on the batch it shares a call with -- the
  //  callers size their batches with kg_max_batch, which budgets every evaluation's table)
  // (r4: 8 and 12 observed derivatives and m > 64 too -- the kernel itself needs neither m nor beta once the sample pre-pass and the
  //  weight table are there)
  // (r4: and any number of observed derivatives up to the slot count -- the table pads a point's weights to 1 + G)
  const bool stream_ok = prep_mode != 0 &&
                         8.0 * (double)v_stride_tiles * (double)num_local / 1e9 <= v_cap;
  if (stream_ok && env_int("MOE_KG_STREAM_WEIGHTS", 1) != 0) {
    // (r3, ms of MC per evaluation, `profiles/r03_variant2_sweep.txt`: C5 7.9 -> 5.0; d = 12, g = 3, q = 8, M = 4000: n = 1200 1.24 -> 0.68 against the
    //  workgroup-per-sample kernel, n = 800 0.80 -> 0.48 against the wave-per-sample kernel streaming its coordinates with five weight
    //  slabs in LDS; without derivative observations it pays once fewer than five slabs fit: n = 6000, d = 8 5.79 -> 4.96, but n = 3000
    //  -- six slabs -- 2.18 -> 2.34)
    if (variant == 1) variant = 2;
    else if (variant == 0 && !xlds && (G > 0 || waves <= 4)) variant = 2;
    // with derivative observations also against the LDS-table kernel, from 12 table rows on or once it runs fewer than eight waves
    // (`profiles/r03_variant2_small.txt`: n = 300 / 500, d = 12, g = 3: 0.29 -> 0.21 / 0.45 -> 0.30 ms; n = 600, d = 8, g = 4, four
    //  slabs: 0.49 -> 0.33; but n = 400, d = 8, g = 2 on eight waves: 0.18 against 0.20 -- stays)
    else if (variant == 0 && xlds && G > 0 && (dp >= 12 || waves < 8)) variant = 2;
  }
  {
    const int forced = env_int("MOE_KG_VARIANT", variant);
    if (forced == 2 && !stream_ok)
      throw Error(MOE_ERR_RUNTIME, "the streamed-weights MC kernel needs the weight table within its cap");
    // (shapes the LDS-slab wave-per-sample kernel is not built for keep to the other two)
    if (!(forced == 0 && (G > 4 || m > kMaxM))) variant = forced;
  }
  if (variant == 0 && waves < 1)
    throw Error(MOE_ERR_RUNTIME, "training set too large for the MC kernel (one sample's weights exceed LDS)");
  const int num_cu = gp.num_cu;
  size_t shm = 0;
  int wg_per_cu = 1, wide_lds_tiles = 0;
  // r5: the lane-parked form of the LDS-table kernel (kg_mc_lane.hpp) wherever it holds as many wavefronts: its LDS also carries the
  // evaluation's record head [L | mu_disc | C_disc | disc] (the same for every evaluation of a call and every batch: the choice
  // depends on the evaluation's shape alone).  Same results bit for bit (MOE_KG_LANE=0: the frame line search of rounds 2-4).
  auto even = [](int v) { return (v + 1) & ~1; };
  const int rec_head = even(m * m) + even(A) + even(A * m) + even(A * size);
  bool lane_kernel = false;
  if (variant == 0 && xlds && waves <= 8 && dp <= 16 && env_int("MOE_KG_LANE", 1) != 0) {
    const size_t lane_fixed = kg_mc_lane_fixed_bytes(dp, rec_head);
    const int wl = std::min(waves, env_int("MOE_KG_WAVES", waves));
    lane_kernel = wl >= 1 && lane_fixed + tab_bytes + (size_t)wl * slab_bytes + pad_bytes <= (size_t)160 * 1024;
  }
  // the simplex update (simplex_limit) lives in the lane-parked line search (r6) and in line_search_lds -- the streamed-weights and the
  // workgroup-per-sample kernel; the frame line search of the other wave-per-sample instantiations has none
  if (simplex && variant == 0 && !lane_kernel) variant = stream_ok ? 2 : 1;
  if (variant == 1 && (tr < 0 || kg_mc_block_lds_bytes(dp, G, num_lds_tiles) > (size_t)160 * 1024))
    throw Error(MOE_ERR_RUNTIME, "point set too large for the workgroup-per-sample MC kernel");
  if (variant == 0) {
    waves = std::max(1, std::min(waves, env_int("MOE_KG_WAVES", waves)));
    shm = (lane_kernel ? kg_mc_lane_fixed_bytes(dp, rec_head) : fixed_bytes) +
          (xlds ? tab_bytes + (size_t)waves * slab_bytes : (size_t)waves * slab_stream_bytes) + pad_bytes;
    wg_per_cu = std::max(1, std::min((int)((size_t)160 * 1024 / shm), (waves > 8 ? 16 : 8) / waves));
    if (mc::wide_eval(dp, xlds)) {  // what the slabs leave of a workgroup's share of LDS holds the leading tiles of the table (kg_mc.hpp WideEval)
      const size_t share = (size_t)160 * 1024 / wg_per_cu;
      wide_lds_tiles = (int)std::min<size_t>((size_t)ntiles, (share - shm) / (sizeof(double) * dp * 64));
      wide_lds_tiles = std::max(0, std::min(wide_lds_tiles, env_int("MOE_KG_WIDE_LDS_TILES", wide_lds_tiles)));
      shm += sizeof(double) * (size_t)wide_lds_tiles * dp * 64;
    }
  } else if (variant == 2) {
    waves = std::max(1, std::min(8, env_int("MOE_KG_WAVES", 8)));
    shm = sizeof(double) * (kExpTabLen + (size_t)waves * mc::kWideScratch);
    wide_lds_tiles = (int)std::min<size_t>((size_t)ntiles, ((size_t)160 * 1024 - shm) / (sizeof(double) * dp * 64));
    wide_lds_tiles = std::max(0, std::min(wide_lds_tiles, env_int("MOE_KG_WIDE_LDS_TILES", wide_lds_tiles)));
    shm += sizeof(double) * (size_t)wide_lds_tiles * dp * 64;
  } else {
    waves = bwaves;
  }
  int blocks = num_cu * wg_per_cu;
  if (blocks >= E) blocks = (blocks / E) * E;  // the same number of workgroups for every evaluation
  // r5, wave-per-sample kernels: no more workgroups per evaluation than its samples need for the same number of rounds -- 128 samples
  // on 12 workgroups of 8 wavefronts are two rounds, and so they are on 8; the CUs left alone go to whatever else is running (the other
  // members of an MCMC ensemble: a suggestion at the headline GP size 0.574 -> 0.540 s; alone on the chip 0.0136 vs 0.0140 ms per
  // evaluation, `profiles/r05_blk_*`).  Which wavefront takes which sample never mattered to the result.
  if (variant != 1 && blocks >= E && waves > 0 && env_int("MOE_KG_COMPACT_GRID", 1) != 0) {
    const long wpe = blocks / E, per_round = wpe * waves;
    const long rounds = (num_local + per_round - 1) / per_round;
    const long need = (num_local + rounds * waves - 1) / (rounds * waves);
    // (only where it frees a fifth of the workgroups or more: 250 instead of 256 for one evaluation of 10 000 samples trades the
    //  ticket counter's slack for nothing -- 0.617 vs 0.598 ms)
    if (need >= 1 && need * 5 <= wpe * 4) blocks = (int)(need * E);
  }
  // r6: an evaluation that shares its launch with the other members of an ensemble (mcmc.hip sets the hint while it records) shares the
  // chip with them, too: 16 members x 20 evaluations x 8 workgroups were ten rounds of workgroups whose wavefronts drew two samples
  // each -- every workgroup ends on its slowest wavefront, a quarter of the kernel's time.  Four draws per wavefront there.
  if (ensemble_members_hint() > 1 && variant != 1 && blocks >= E && waves > 0) {
    const long tickets = env_int("MOE_KG_ENS_TICKETS", 4);
    const long k = std::max<long>(1, (num_local + tickets * waves - 1) / (tickets * waves));
    blocks = (int)std::min<long>(blocks, k * E);
  }
  blocks = env_int("MOE_KG_BLOCKS", blocks);

  const auto wall0 = std::chrono::steady_clock::now();

  // ---- 1. state set-up for the whole batch ----
  std::vector<double> U_all((size_t)E * u * d), extra_all((size_t)E * A * d), disc_all((size_t)E * A * size);
  for (int e = 0; e < E; ++e) {
    double* U = &U_all[(size_t)e * u * d];
    const double* Xq = Xq_all + (size_t)e * q * d;
    // union of points, discretised set = [Xu without fidelity dims ; discrete points]  (.cpp:246-261)
    std::copy(Xq, Xq + (size_t)q * d, U);
    if (p > 0) std::copy(Xp, Xp + (size_t)p * d, U + (size_t)q * d);
    double* ds = &disc_all[(size_t)e * A * size];
    // (disc_head: the state this evaluation runs on was BUILT at other points_to_sample and moved here with SetCurrentPoint,
    //  which leaves the discretised set behind -- .cpp:232-243 vs 259-261; the reference's multistart drivers do that)
    for (int i = 0; i < u; ++i) {
      const double* src = (disc_head != nullptr && i < q) ? disc_head + (size_t)i * d : U + (size_t)i * d;
      std::copy(src, src + size, ds + (size_t)i * size);
    }
    std::copy(discrete, discrete + (size_t)P * size, ds + (size_t)u * size);
    double* ex = &extra_all[(size_t)e * A * d];
    for (int j = 0; j < A; ++j) {
      for (int k = 0; k < size; ++k) ex[(size_t)j * d + k] = ds[(size_t)j * size + k];
      for (int k = size; k < d; ++k) ex[(size_t)j * d + k] = 1.0;
    }
  }
  // ---- table-row order of the dimensions ----
  TabParams tp;
  {
    // table-row order: the GP's observed-derivative dimensions first (so derivative weight a multiplies row a), then
    // the remaining dimensions in ascending order; padded rows map to themselves (their lengths are 0)
    std::vector<int> order;
    std::vector<bool> used(kMaxDimPadded, false);
    for (int a = 0; a < g; ++a) {
      if (used[gp.derivs.idx[a]]) throw Error(MOE_ERR_INVALID_VALUE, "duplicate derivative index", gp.derivs.idx[a], 0, 0);
      order.push_back(gp.derivs.idx[a]);
      used[gp.derivs.idx[a]] = true;
    }
    for (int k = 0; k < kMaxDimPadded; ++k)  // derivative dims are < d <= dp, so rows [0, dp) are a permutation of [0, dp)
      if (!used[k]) order.push_back(k);
    if (dp < G) throw Error(MOE_ERR_RUNTIME, "padded dimension smaller than the derivative-slot count");
    for (int r = 0; r < kMaxDimPadded; ++r) {
      tp.perm[r] = order[r];
      // frame scale: the Matern kernel's sqrt(5) is folded into it (kg_mc.hpp radial3)
      tp.inv_lp[r] = gp.cp.inv_l[order[r]] * (gp.cp.type == MOE_COV_MATERN_NU_2P5 ? 2.236067977499789696409173668731276235 : 1.0);
      const double c = (order[r] < d) ? gp.x_mean[order[r]] : 0.0;  // training-set mean of the row's coordinate (0 for pad rows)
      tp.center[r] = c;
      // the MC kernels' exponent arithmetic covers |x - c| / l up to kTableExtent for every tabulated point
      if (order[r] < d) {
        const int k = order[r];
        double ext = gp.x_ext[k];
        for (int e = 0; e < E; ++e) {
          for (int i = 0; i < q; ++i) ext = std::max(ext, std::fabs(Xq_all[((size_t)e * q + i) * d + k] - c));
        }
        for (int i = 0; i < p; ++i) ext = std::max(ext, std::fabs(Xp[(size_t)i * d + k] - c));
        if (disc_head != nullptr)
          for (int i = 0; i < q; ++i) ext = std::max(ext, std::fabs(disc_head[(size_t)i * d + k] - c));
        if (!(ext * tp.inv_lp[r] <= mc::kTableExtent))
          throw Error(MOE_ERR_BOUNDS, "length scale too small for the extent of the points (|x - mean| / length > 1e5)",
                      ext * tp.inv_lp[r], 0.0, mc::kTableExtent);
      }
    }
  }
  // ---- per-evaluation records: the host fills what it knows (discretised set, padded union points); mu, the factor of
  // Var + noise, the discretised set's mu_n / c_j and the best posterior mean are written by kg_state_kernel ----
  KgRec rec;
  int off = 0;
  auto take = [&](int cnt) {
    const int o = off;
    off += (cnt + 1) & ~1;
    return o;
  };
  rec.L = take(m * m);
  rec.mu_disc = take(A);
  rec.C_disc = take(A * m);
  rec.disc = take(A * size);
  if (off != rec_head) throw Error(MOE_ERR_RUNTIME, "record head layout mismatch (lane-parked MC kernel)");
  rec.XuP = take(u * dp);
  const int rec_bp = take(1);
  rec.Mk = 0;
  rec.stride = off;
  // behind the records: [2 kMaxDimPadded] bounds | [kMaxDimPadded] frame centre | [kMaxDimPadded] frame scale, table-row order
  // (the MC kernels read them one row per lane)
  const size_t blob_size = (size_t)rec.stride * E + 4 * kMaxDimPadded;
  const size_t o_bounds = (size_t)rec.stride * E;
  const long num_norm = (long)((num_mc + 1) / 2) * m;
  gp.hKgIn.reserve(blob_size);  // (host-side assembly area; it travels inside the state set-up's staging buffer)
  double* blob = gp.hKgIn.p;
  std::memset(blob, 0, sizeof(double) * blob_size);
  unsigned int free_mask = 0;  // table-row order: bounds of row r = bounds of original dimension perm[r]
  // (simplex inner domain, SimplexIntersectTensorProductDomain's constructor, gpp_domain.cpp:107-141: the box clipped to the unit
  //  hypercube; an empty intersection is the reference's BoundsException)
  std::vector<double> box(bounds, bounds + 2 * (size_t)size);
  if (simplex) {
    double corner_sum = 0.0;
    bool empty = false;
    for (int k = 0; k < size; ++k) {
      box[2 * k] = std::fmax(bounds[2 * k], 0.0);
      box[2 * k + 1] = std::fmin(bounds[2 * k + 1], 1.0);
      empty = empty || box[2 * k] > box[2 * k + 1];
      corner_sum += box[2 * k];
    }
    if (corner_sum >= 1.0 || empty)
      throw Error(MOE_ERR_BOUNDS,
                  "Simplex/Tensor product intersection is EMPTY; 'lower left' corner coordinate sum out of bounds or bounding "
                  "boxes do not intersect.",
                  corner_sum, 0.0, 1.0);
  }
  for (int r = 0; r < kMaxDimPadded; ++r) {
    const int k = tp.perm[r];
    if (k < size) {
      blob[o_bounds + 2 * r] = box[2 * k];
      blob[o_bounds + 2 * r + 1] = box[2 * k + 1];
      free_mask |= 1u << r;
    }
    blob[o_bounds + 2 * kMaxDimPadded + r] = tp.center[r];
    blob[o_bounds + 3 * kMaxDimPadded + r] = tp.inv_lp[r];
  }
  for (int e = 0; e < E; ++e) {
    const double* U = &U_all[(size_t)e * u * d];
    double* r = &blob[(size_t)rec.stride * e];
    std::copy(&disc_all[(size_t)e * A * size], &disc_all[(size_t)(e + 1) * A * size], r + rec.disc);
    for (int i = 0; i < u; ++i)
      for (int k = 0; k < d; ++k) r[rec.XuP + (size_t)i * dp + k] = U[(size_t)i * d + k];
  }

  DerivList none;
  none.g = 0;
  for (int i = 0; i < kMaxDerivs; ++i) none.idx[i] = 0;
  auto timers = std::make_shared<std::array<EventTimer, 4>>();
  EventTimer &t_mc = (*timers)[0], &t_cov = (*timers)[1], &t_tail = (*timers)[2], &t_state = (*timers)[3];
  t_state.start(s);
  // everything N-sized of the state, left on the device (gp.hip): no wait, no download -- the m x m algebra follows as kernels.
  // The records and the normal draws ride down in the same host->device copy as the points (r4: one copy per call instead of three).
  StateAppendix apx;
  apx.doubles = blob_size + (size_t)num_norm;
  apx.fill = [&](double* dst) {
    std::memcpy(dst, blob, sizeof(double) * blob_size);
    std::memcpy(dst + blob_size, normals, sizeof(double) * num_norm);
  };
  const KgStateEnqueued se = enqueue_kg_state_batch(gp, U_all.data(), u, want_grad ? q : 0, extra_all.data(), A, E, &apx);
  const BatchLayout& bl = se.bl;
  double* dBlobP = const_cast<double*>(gp.dAppendix);  // (kg_state_kernel completes the records in place)
  const double* dNormalsP = gp.dAppendix + blob_size;

  // ---- device buffers ----
  DevBuf<double>&dTab = gp.kTab, &dBestPoint = gp.kBestPoint, &dBestValue = gp.kBestValue, &dBeta = gp.kBeta, &dT = gp.kT, &dC = gp.kC,
  &dTB = gp.kTB, &dOut = gp.kOut;
  DevBuf<unsigned long long>& dCounters = gp.kCounters;
  // ---- the m x m algebra of the state, on the device (kg_state.hip) ----
  const int qd = q * d;
  const int tri = m * (m + 1) / 2;
  gp.kStateI.reserve((size_t)2 * E);
  gp.kStateD.reserve((size_t)E * qd * (1 + (want_grad ? tri : 0)) + (size_t)E * (1 + qd + 3) + (size_t)E * tri);
  KgStateParams sp;
  sp.cp = gp.cp;
  sp.derivs = gp.derivs;
  for (int b = 0; b <= kMaxDerivs; ++b) sp.noise[b] = (b < g1) ? gp.noise[b] : 0.0;
  sp.mean = gp.mean;
  sp.best_so_far = best_so_far;
  sp.E = E;
  sp.u = u;
  sp.q = q;
  sp.m = m;
  sp.g = g;
  sp.d = d;
  sp.dp = dp;
  sp.A = A;
  sp.ng = ngrad;
  sp.gkk = se.gkk;
  sp.gx = se.gx;
  sp.gkk_slices = se.gkk_slices;
  sp.gx_slices = se.gx_slices;
  sp.ek = se.ek;
  sp.U = se.U;
  sp.extra = se.extra;
  sp.blob = dBlobP;
  sp.rec_stride = rec.stride;
  sp.rec_L = rec.L;
  sp.rec_mu_disc = rec.mu_disc;
  sp.rec_C_disc = rec.C_disc;
  sp.rec_bp = rec_bp;
  sp.flags = gp.kStateI.p;
  sp.winner = gp.kStateI.p + E;
  sp.gmu = gp.kStateD.p;
  sp.dL = gp.kStateD.p + (size_t)E * qd;
  double* dFin = gp.kStateD.p + (size_t)E * qd * (1 + (want_grad ? tri : 0));
  launch_kg_state(sp, s);
  if (want_grad) launch_kg_dchol(sp, s);
  const long tab_stride = (long)ntiles * dp * 64;
  dTab.reserve((size_t)tab_stride * E + (size_t)dp * 64);  // + one tile: eval_loop's last prefetch reads past the end
  dBestPoint.reserve((size_t)E * num_local * dp);
  dBestValue.reserve((size_t)E * num_local);
  dBeta.reserve((size_t)E * num_local * m + 64);  // + a zeroed pad: kg_sample_weights_kernel reads whole 16/32/64-blocks
  // [2 E] pass counters | [E] sample-ticket counters, one 128-byte line each (kTicketStride unsigned ints)
  const size_t n_ctr = (size_t)2 * E + (size_t)E * (kTicketStride / 2) + 32;  // (+16 alignment slack, +16 profiling words)
  dCounters.reserve(n_ctr);
  const int out_stride = 1 + m * m + 2 * ngrad;
  dOut.reserve((size_t)out_stride * E);
  // q-KG fast path: the N x M covariance matrix of the tail is never materialised (see launch_fused_tail)
  const bool fused_tail = want_grad && g == 0 && m <= 8 && env_int("MOE_KG_FUSED_TAIL", 1) != 0;
  const int chunk_len = fused_tail ? kFusedChunk : (m > 64 ? kTbChunkWide : kTbChunk);
  const int chunks = (num_local + chunk_len - 1) / chunk_len;
  if (want_grad) {
    if (!fused_tail) dT.reserve((size_t)N * E * num_local);
    dC.reserve((size_t)E * num_local * m);
    dTB.reserve((size_t)E * (chunks + 1) * m * N + (size_t)E * ngrad * kDirSlices);  // chunk partials + their sum + DIR partials
  }

  // ---- coordinate tables ----
  {
    dim3 grid((unsigned)((tab_stride + 255) / 256), E);
    const double *dXp = gp.dX.p, *dUnionP = gp.dUnion;
    double* dTabP = dTab.p;
    unsigned long long* dCtrP = dCounters.p;
    const int pair_rows = (wide_dp || variant == 2 || (variant == 0 && mc::wide_eval(dp, xlds))) ? 1 : 0;
    launch_kernel_ens<build_xs_tab_kernel_body, 256>(build_xs_tab_kernel, grid, dim3(256), 0, s, dXp, n, dUnionP, u, dp, ntiles, tp, dTabP, tab_stride,
                                                     pair_rows, dCtrP, (long)n_ctr);
    MOE_HIP_CHECK(hipGetLastError());
  }
  t_state.stop(s);

  // ---- 2. MC kernel ----
  KgMcParams mp;
  mp.cov_type = gp.cp.type;
  mp.dim = d;
  mp.alpha = gp.cp.alpha;
  for (int r = 0; r < kMaxDimPadded; ++r) {
    mp.inv_lp[r] = tp.inv_lp[r];
    mp.perm[r] = tp.perm[r];
    mp.center[r] = tp.center[r];
  }
  mp.n = n;
  mp.g = g;
  mp.N = N;
  mp.u = u;
  mp.m = m;
  mp.f = f;
  mp.A = A;
  mp.ntiles = ntiles;
  mp.E = E;
  mp.mean = gp.mean;
  mp.multi_trial = (far_frame || small_lane) ? 0 : env_int("MOE_KG_MULTI_TRIAL", 1);  // (0: A/B runs)
  // r6: the small shapes on the lane-parked kernel take their Armijo trials several per sweep, too -- each trial computed exactly as
  // a single-trial pass computes it (kg_mc.hpp eval_multi_exact: same bits as one trial per pass; MOE_KG_SMALL_MULTI=0: one per pass)
  if (small_lane && !far_frame && variant == 0 && lane_kernel && dp == 4 && env_int("MOE_KG_SMALL_MULTI", 1) != 0) mp.multi_trial = 2;
  mp.XsTab = dTab.p;
  mp.tab_stride = tab_stride;
  mp.wide_lds_tiles = wide_lds_tiles;
  mp.KinvY = gp.dKinvY.p;
  mp.W = gp.dWE.p + bl.col_kstar0(0) * N;
  mp.w_stride = (long)m * N;
  mp.blob = dBlobP;
  mp.rec = rec;
  mp.bounds = dBlobP + o_bounds;
  mp.free_mask = free_mask;
  mp.normals = dNormalsP;
  mp.first_sample = first_sample;
  mp.num_local = num_local;
  mp.max_num_steps = gd.max_num_steps;
  mp.max_num_restarts = gd.max_num_restarts;
  mp.gamma = gd.gamma;
  mp.pre_mult = gd.pre_mult;
  mp.max_relative_change = gd.max_relative_change;
  // (gpp_domain.cpp:245-247: the simplex update never lands precisely on a wall -- kRelativeChangeEpsilonTweak)
  if (simplex && mp.max_relative_change == 1.0) mp.max_relative_change -= 4.0 * 2.220446049250313e-16;
  mp.simplex = simplex ? 1 : 0;
  mp.inv_sqrt_size = 1.0 / std::sqrt((double)size);
  for (int r = 0; r < kMaxDimPadded; ++r) mp.inv_perm[r] = 0;
  for (int r = 0; r < kMaxDimPadded; ++r)
    if (tp.perm[r] >= 0 && tp.perm[r] < kMaxDimPadded) mp.inv_perm[tp.perm[r]] = r;
  mp.tolerance = gd.tolerance;
  mp.best_point = dBestPoint.p;
  mp.best_value = dBestValue.p;
  mp.beta = dBeta.p;
  mp.counters = dCounters.p;
  mp.next_sample = reinterpret_cast<unsigned int*>(dCounters.p + (((size_t)2 * E + 15) / 16) * 16);  // 128-byte aligned
  mp.prof = dCounters.p + n_ctr - 16;
  t_mc.start(s);
  mp.best_j = nullptr;
  mp.V = nullptr;
  // the sample pre-pass (beta and the discretised-set winner of every sample, computed with the whole chip): for the
  // workgroup-per-sample kernel, where seven wavefronts would wait for one; the wave-per-sample kernel hides those loads
  // behind the other wavefront of its SIMD (measured: no difference at C3), so it only takes the pre-pass on request
  // (MOE_KG_PREP=1; MOE_KG_PREP=0: never)
  if (m > kMaxM || (prep_mode != 0 && (variant >= 1 || prep_mode == 1))) {
    gp.kBestJ.reserve((size_t)E * num_local);
    mp.best_j = gp.kBestJ.p;
    const long total = (long)E * num_local;
    int* best_j_p = gp.kBestJ.p;
    if (m > kMaxM) {  // more components than lanes: thread-per-sample pre-pass (mandatory there)
      launch_kernel_ens<kg_sample_prep_generic_kernel_body, 64>(kg_sample_prep_generic_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, s, mp, best_j_p);
    } else {
      const int pb = (int)std::min<long>((total + 3) / 4, (long)num_cu * 8);
      launch_kernel_ens<kg_sample_prep_kernel_body, 256>(kg_sample_prep_kernel, dim3(pb), dim3(256), 0, s, mp, best_j_p);
    }
    MOE_HIP_CHECK(hipGetLastError());
  }
  mp.v_stride = (variant == 2) ? v_stride_tiles : N;
  mp.v_slots1 = (variant == 2) ? 1 + G : g1;
  if (variant >= 1 && mp.best_j != nullptr) {
    const long total = (long)E * num_local;
    // the weight table: N doubles per sample (1.28 GB per evaluation at C5); beyond its cap -- the caller's share of the
    // workspace budget (kg_evaluate_batch, kg_mcmc_sums), MOE_KG_V_MAX_GB (default 4) for a bare kg_launch -- the samples
    // compute their weights in the kernel (workgroup-per-sample kernel only: the streamed-weights one is not chosen beyond the cap)
    const double v_gb = 8.0 * (double)mp.v_stride * (double)num_local / 1e9;  // (per evaluation, as above)
    if (v_gb <= v_cap) {  // (m > 64: the table kernel takes the columns of W 64 at a time, r4)
      gp.kV.reserve((size_t)mp.v_stride * (size_t)total + (size_t)64 * mp.v_slots1);  // (+ one tile: the sweeps prefetch one tile ahead)
      memset_async(dBeta.p + (size_t)total * m, 0, sizeof(double) * 64, s);
      mp.V = gp.kV.p;
      launch_sample_weights(mp, gp.kV.p, s, variant == 2);
    }
  }
  if (variant == 2 && (mp.V == nullptr || mp.best_j == nullptr))
    throw Error(MOE_ERR_RUNTIME, "streamed-weights MC kernel selected without its weight table");
  if (variant == 0 && lane_kernel)
    launch_mc_lane(mp, dp, G, rec_head, blocks, waves, shm, s);
  else if (variant == 0)
    launch_mc(mp, dp, G, xlds, blocks, waves, shm, s);
  else if (variant == 2)
    launch_mc_stream(mp, dp, G, blocks, waves, shm, s);
  else
    launch_mc_block(mp, dp, G, tr, num_lds_tiles, blocks, waves, s);
  t_mc.stop(s);
  {
    // (bits 1 / 2 of the second word, r4: the frame-extent decisions -- a domain box or point set wider than 100 length scales
    //  silently costs the LDS-table kernel and the multi-trial passes; this is where a caller can see it)
    const int info[8] = {variant, ((variant == 0 && xlds) ? 1 : 0) | (far_frame ? 2 : 0) | (wide_frame ? 4 : 0) | (lane_kernel ? 8 : 0), waves, variant == 1 ? tr : wide_lds_tiles, mp.V != nullptr ? 1 : 0, 0, blocks,
                         mp.best_j != nullptr ? 1 : 0};
    std::copy(info, info + 8, gp.last_info);
  }

  // ---- 3. gradient tail ----
  KgTailParams tl;
  tl.cp = gp.cp;
  tl.derivs = gp.derivs;
  tl.u = u;
  tl.q = q;
  tl.m = m;
  tl.g = g;
  tl.N = N;
  tl.E = E;
  tl.num_local = num_local;
  tl.first_sample = first_sample;
  tl.ngrad = ngrad;
  tl.chunks = chunks;
  tl.chunk_len = chunk_len;
  tl.sw_chunk = ((long)E * num_local / kSwChunk >= 1024) ? kSwChunk : 16;
  tl.T = dT.p;
  tl.SW = nullptr;
  tl.W = mp.W;
  tl.w_stride = mp.w_stride;
  tl.Gm = gp.dE.p + bl.col_grad0(0) * N;  // dK*/dXq itself: K^-1 goes onto the m columns of TB instead (launch_gtb)
  tl.g_stride = (long)ngrad * N;
  tl.Linv = gp.dLinv.p;
  tl.ldL = gp.ldL;
  tl.tri_work = gp.dEK.p;  // (reserved by enqueue_kg_state_batch; the state's own use of it is behind us in stream order)
  tl.work = gp.dVE.p;  // [2][N x E m]: enqueue_kg_state_batch reserves it; V = L^-1 K* is not needed any more
  tl.blob = dBlobP;
  tl.rec = rec;
  tl.rec_bp = rec_bp;
  tl.best_point = dBestPoint.p;
  tl.best_value = dBestValue.p;
  tl.beta = dBeta.p;
  tl.normals = dNormalsP;
  tl.C = dC.p;
  tl.TBpart = dTB.p;
  tl.out = dOut.p;
  tl.out_stride = out_stride;
  gp.last_info[5] = fused_tail ? 1 : 0;
  if (fused_tail) {
    t_cov.start(s);
    t_cov.stop(s);  // no covariance matrix is built on this path
    t_tail.start(s);
    const int slices = fused_tail_slices(E, num_local, n, num_cu, m, dp);
    gp.kSW.reserve((size_t)E * num_local * slices * 8);  // [E][num_local][slices][MU <= 8]
    launch_fused_tail(tl, gp.dX.p, n, gp.kSW.p, slices, s);
    // (r4: m > 64 too -- one workgroup per entry of ZC walking every sample with a stride of m doubles took 1.8 ms per evaluation at m = 104)
    gp.kZcPart.reserve((size_t)E * ((num_local + zc_chunk_len(m) - 1) / zc_chunk_len(m)) * m * m);
    launch_zc(tl, gp.kZcPart.p, s, !zc_sum_in_finish(m, num_local));
  } else if (want_grad) {
    t_cov.start(s);
    launch_cov_build(gp.cp, gp.dX.p, n, gp.derivs, dBestPoint.p, E * num_local, none, nullptr, dT.p, N, 0, s, true);
    t_cov.stop(s);
    t_tail.start(s);
    if (m > 8 || (size_t)N * m * sizeof(double) > 96 * 1024) {
      // S_W = W_e^T T_e per evaluation as a tile GEMM (the one-wave-per-sample loop re-reads W from L2 for every sample,
      // N m 8 bytes each -- fine for q-KG's m = q + p, prohibitive for d-KG's m = (q + p)(1 + g))
      gp.kSW.reserve((size_t)m * E * num_local);
      // A skinny output (m x samples) over a long K (N): every evaluation of the call in ONE launch, K cut into slices so that the chip
      // holds several workgroups per CU (r4: one launch per evaluation, unsplit below N = 2048, was 62 % of the d-KG tail at n = 500 --
      // 63 workgroups walking 125 dependent stages, eight times per call).  The slice count is a function of (m, N) alone.
      const int sw_slices = N >= 1024 ? 8 : (N >= 256 ? 4 : 1);  // (m > 64 too: the stretch point's 104 x 20 000 over N = 26 000 walked 1625 stages unsplit)
      if (sw_slices > 1) gp.kSWpart.reserve((size_t)sw_slices * m * num_local * E);
      if (m > 64 && sw_slices > 1) {  // (T is set in `tl` below: the kernel takes W / T / strides from it)
        KgTailParams tq = tl;
        tq.T = dT.p;
        MOE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kg_sw128_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)g128::kSmemBytes));
        double* sw_part_p = gp.kSWpart.p;
        launch_kernel_ens<kg_sw128_kernel_body, 256, 2>(kg_sw128_kernel, dim3((num_local + g128::TM - 1) / g128::TM, sw_slices, E), dim3(256), g128::kSmemBytes, s, tq,
                   sw_part_p, sw_slices);
        launch_sum_slices(gp.kSWpart.p, sw_slices, (long)m * num_local * E, gp.kSW.p, s);
      } else {
        launch_gemm_tn_splitk(m, num_local, N, tl.W, N, dT.p, N, gp.kSW.p, gp.kSWpart.p, sw_slices, s, E, tl.w_stride, (long)num_local * N);
      }
      tl.SW = gp.kSW.p;
    }
    launch_tail(tl, s);
    // (r4: m > 64 too -- one workgroup per entry of ZC walking every sample with a stride of m doubles took 1.8 ms per evaluation at m = 104)
    gp.kZcPart.reserve((size_t)E * ((num_local + zc_chunk_len(m) - 1) / zc_chunk_len(m)) * m * m);
    launch_zc(tl, gp.kZcPart.p, s, !zc_sum_in_finish(m, num_local));
  } else {
    const unsigned long long* ctr_p = dCounters.p;
    const int* flags_p = gp.kStateI.p;
    launch_kernel_ens<kg_sum_kernel_body, 256>(kg_sum_kernel, dim3(E), dim3(256), 0, s, tl, dFin, ctr_p, flags_p);
  }
  MOE_HIP_CHECK(hipGetLastError());
  // ---- grad KG from the sample sums, on the device (kg_state.hip) ----
  if (want_grad) {
    KgFinishParams fp;
    fp.E = E;
    fp.q = q;
    fp.m = m;
    fp.g = g;
    fp.d = d;
    fp.ng = ngrad;
    fp.num_mc = num_mc;
    fp.first_sample = first_sample;
    fp.blob = dBlobP;
    fp.rec_stride = rec.stride;
    fp.rec_L = rec.L;
    fp.out = dOut.p;
    fp.out_stride = out_stride;
    fp.winner = sp.winner;
    fp.gmu = sp.gmu;
    fp.dL = sp.dL;
    fp.fin = dFin;
    fp.counters = dCounters.p;
    fp.flags = gp.kStateI.p;
    fp.dir_part = dir_partials(tl);
    fp.dir_slices = dir_slices(num_local);
    fp.zc_part = zc_sum_in_finish(m, num_local) ? gp.kZcPart.p : nullptr;
    fp.zc_chunks = (num_local + zc_chunk_len(m) - 1) / zc_chunk_len(m);
    fp.zc_gs = zc_sum_lanes(m);
    fp.best_value = dBestValue.p;
    fp.num_local = num_local;
    fp.rec_bp = rec_bp;
    launch_kg_finish(fp, dFin + (size_t)E * (1 + qd + 3), s);
    t_tail.stop(s);
  }
  // results through pinned memory in ONE copy: per evaluation kg_sum | grad_sum (q d) | value passes | gradient passes | flag
  const int fin_stride = (want_grad ? 1 + qd : 1) + 3;
  const size_t n_out = (size_t)fin_stride * E;
  gp.hKgOut.reserve(n_out);
  double* out = gp.hKgOut.p;
  copy_async(out, dFin, sizeof(double) * n_out, hipMemcpyDeviceToHost, s, true);  // (hKgOut: pinned)
#if MOE_BLOCK_PROF
  auto prof_p = std::make_shared<std::vector<unsigned long long>>(16);
  copy_async(prof_p->data(), dCounters.p + n_ctr - 16, sizeof(unsigned long long) * 16, hipMemcpyDeviceToHost, s);
#endif
  const bool fetch_bp = want_best_points && E == 1;
  auto bp_p = std::make_shared<std::vector<double>>();
  if (fetch_bp) {
    bp_p->resize((size_t)num_local * dp);
    dBestPoint.download(bp_p->data(), bp_p->size(), s);
  }
  GpDev* gpp = &gp;
  KgPending pending;
  pending.collect = [=](double* kg_sum, double* grad_sum, double* best_points, moe_kg_stats_t* stats) {
    GpDev& gp = *gpp;
    gp.use_device();
    if (stats) std::memset(stats, 0, sizeof(*stats));
    MOE_HIP_CHECK(hipStreamSynchronize(s));
#if MOE_BLOCK_PROF
    {
      const unsigned long long* w = prof_p->data();
      const double tot = (double)(w[0] + w[1] + w[2] + w[3]);
      if (tot > 0 && w[8] > 0)
        std::fprintf(stderr,
                     "[moe prof] wave-0 cycles: ticket %.1f%%  z/beta/scan %.1f%%  weights %.1f%%  line search %.1f%%;  inside "
                     "%llu passes: accumulate %.0f  reduce %.0f  barrier %.0f  post %.0f  (cycles per pass; line search total "
                     "%.0f per pass)\n",
                     100.0 * w[0] / tot, 100.0 * w[1] / tot, 100.0 * w[2] / tot, 100.0 * w[3] / tot, w[8],
                     (double)w[4] / w[8], (double)w[5] / w[8], (double)w[6] / w[8], (double)w[7] / w[8], (double)w[3] / w[8]);
      if (w[10] > 0)
        std::fprintf(stderr, "[moe prof] gradient passes: %llu, %.0f cycles each (all phases); cycles per SAMPLE: total %.0f, in passes %.0f\n",
                     w[10], (double)w[9] / w[10], tot / (w[10] / 6.0), (double)(w[4] + w[5] + w[6] + w[7]) / (w[10] / 6.0));
      if (w[10] > 0)
        std::fprintf(stderr, "[moe prof] per sample: z/beta %.0f, scan %.0f cycles\n", (double)w[11] / (w[10] / 6.0), (double)w[12] / (w[10] / 6.0));
      if (w[10] > 0 && w[8] > 0)  // workgroup-per-sample kernel: line_search_lds outside its passes, per pass
        std::fprintf(stderr, "[moe prof] outside the passes, cycles per pass: gradient post + trial-line set-up %.0f, Armijo dispatch / decisions %.0f, "
                     "LimitUpdate + re-evaluation set-up + step end %.0f\n", (double)w[13] / w[8], (double)w[14] / w[8], (double)w[15] / w[8]);
      if (w[13] > 0 && w[10] == 0)  // wave-per-sample kernel (kg_mc.hpp kg_sample): clock ticks of lane 0 of every wave
        std::fprintf(stderr,
                     "[moe prof] wave-per-sample kernel, ticks per sample: z/beta %.0f  weights %.0f  scan %.0f  line search %.0f "
                     "(of which %.1f value passes x %.0f + %.1f gradient passes x %.0f)\n",
                     (double)w[0] / w[13], (double)w[1] / w[13], (double)w[2] / w[13], (double)w[3] / w[13], (double)w[6] / w[13],
                     (double)w[4] / std::max<unsigned long long>(w[6], 1), (double)w[7] / w[13],
                     (double)w[5] / std::max<unsigned long long>(w[7], 1));
    }
#endif
    for (int e = 0; e < E; ++e)
      if (out[(size_t)fin_stride * (e + 1) - 1] != 0.0)
        throw Error(MOE_ERR_SINGULAR,
                    "GP-Variance matrix singular. Check for duplicate points_to_sample/being_sampled or "
                    "points_to_sample/being_sampled duplicating points_sampled with 0 noise.",
                    m, out[(size_t)fin_stride * (e + 1) - 1]);
    if (best_points && fetch_bp)
      for (int i = 0; i < num_local; ++i)
        for (int k = 0; k < d; ++k) best_points[(size_t)i * d + k] = (*bp_p)[(size_t)i * dp + k];

    for (int e = 0; e < E; ++e) {
      const double* o = &out[(size_t)fin_stride * e];
      kg_sum[e] = o[0];
      if (want_grad) std::copy(o + 1, o + 1 + qd, grad_sum + (size_t)e * qd);
      if (stats) {
        stats->posterior_mean_evals += (long long)o[fin_stride - 3];
        stats->posterior_grad_evals += (long long)o[fin_stride - 2];
      }
    }
    const double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    const double ms_mc = (*timers)[0].ms(), ms_cov = want_grad ? (*timers)[1].ms() : 0.0,
                 ms_tail = want_grad ? (*timers)[2].ms() : 0.0;
    gp.last_ms[0] = ms_mc / E;
    gp.last_ms[1] = ms_cov / E;
    gp.last_ms[2] = ms_tail / E;
    const double ms_state = (*timers)[3].ms();
    gp.last_ms[3] = ms_state / E;
    gp.last_ms[4] = wall / E;
    if (stats) {
      stats->ms_state = ms_state;
      stats->ms_mc = ms_mc;
      stats->ms_tail = ms_cov + ms_tail;
    }
  };
  return pending;
}

// How many evaluations one kg_launch may carry on this GP within `budget_gb` of device workspace (>= 1): the state
// matrices E / VE / WE, the materialised tail matrix T (d-KG and m > 8) with its TB partials, the per-sample outputs.
// A d-KG evaluation at C5 is 1.4 GB (N x M doubles of T alone), so a 200-start multistart cannot go down in one piece.
int kg_max_batch(const GpDev& gp, int P, int q, int p, int num_local, bool want_grad, double budget_gb) {
  const double N = gp.N, g1 = 1 + gp.g, u = q + p, m = u * g1, A = u + P;
  const double ngrad = want_grad ? q * g1 * gp.d : 0.0;
  const bool fused = want_grad && gp.g == 0 && m <= 8;
  const double chunks = std::ceil((double)num_local / (fused ? kFusedChunk : (m > 64 ? kTbChunkWide : kTbChunk)));
  // state matrix E (all columns), V = L^-1 K* with the tail's K^-1 TB workspace behind it, W = K^-1 K*; the packed d chol / d Xq
  double doubles = N * (m + ngrad + A) + 3.0 * N * m + (double)num_local * (gp.dp + 1 + 2 * m) +
                   (want_grad ? (double)q * gp.d * m * (m + 1) / 2 : 0.0);
  if (want_grad) doubles += (fused ? 0.0 : N * (double)num_local) + (chunks + 1.0) * m * N;
  if (want_grad && !fused) doubles += 9.0 * m * (double)num_local;  // S_W = W^T T and its (up to eight) K slices
  // the per-sample weight table of the workgroup-per-sample / streamed-weights kernels (the latter: whole tiles, fantasy points included)
  // (always: whether an evaluation takes one of those kernels is decided per evaluation in kg_launch -- far frames send small
  //  q-KG shapes there too -- and the batch must fit whatever it decides)
  const double slots1 = gp.g <= 4 ? g1 : (gp.g <= 8 ? 9.0 : 13.0);  // (the streamed-weights table pads a point's weights to 1 + G)
  doubles += (gp.n + u + 64.0) * slots1 * (double)num_local;
  //  the split-K workspace of the state's triangular / Gram products (gp.dEK: ~8 N m per evaluation) and the evaluation's
  // share of the staging appendix (its record: L, mu_disc, C_disc, the discretised set, the padded union points)
  doubles += 8.0 * N * m + (m * m + A * (1.0 + m + gp.d) + u * gp.dp + 8.0);
  const double per_eval_gb = 8.0 * doubles / 1e9;
  return (int)std::max(1.0, std::floor(budget_gb / std::max(per_eval_gb, 1e-9)));
}

void kg_evaluate_batch(GpDev& gp, int num_fidelity, const moe_gd_params_t& gd, const double* bounds, const double* discrete,
                       int P, const double* Xq_all, int num_evals, const double* Xp, int q, int p, int num_mc,
                       double best_so_far, const double* normals, int first_sample, int num_local, bool want_grad,
                       double* kg_sum, double* grad_sum, double* best_points, moe_kg_stats_t* stats, const double* disc_head) {
  // batches beyond the workspace budget (MOE_KG_BATCH_GB, default 48) go down in pieces
  const double budget = (double)env_int("MOE_KG_BATCH_GB", 48);
  const int max_e = kg_max_batch(gp, P, q, p, num_local, want_grad, budget);
  if (num_evals <= max_e) {
    KgPending pending = kg_launch(gp, num_fidelity, gd, bounds, discrete, P, Xq_all, num_evals, Xp, q, p, num_mc, best_so_far,
                                  normals, first_sample, num_local, want_grad, best_points != nullptr, budget, disc_head);
    pending.collect(kg_sum, grad_sum, best_points, stats);
    return;
  }
  const size_t qd = (size_t)q * gp.d;
  moe_kg_stats_t total{}, part{};
  for (int e0 = 0; e0 < num_evals; e0 += max_e) {
    const int ne = std::min(max_e, num_evals - e0);
    KgPending pending = kg_launch(gp, num_fidelity, gd, bounds, discrete, P, Xq_all + (size_t)e0 * qd, ne, Xp, q, p, num_mc,
                                  best_so_far, normals, first_sample, num_local, want_grad, false, budget, disc_head);
    pending.collect(kg_sum + e0, grad_sum ? grad_sum + (size_t)e0 * qd : nullptr, nullptr, stats ? &part : nullptr);
    total.posterior_mean_evals += part.posterior_mean_evals;
    total.posterior_grad_evals += part.posterior_grad_evals;
    total.ms_state += part.ms_state;
    total.ms_mc += part.ms_mc;
    total.ms_tail += part.ms_tail;
  }
  if (stats) *stats = total;
}

}  // namespace moe
