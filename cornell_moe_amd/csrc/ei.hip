// cornell_moe_amd/csrc/ei.hip -- q,p-EI by Monte Carlo (value + gradient) on gfx950.
//
// ExpectedImprovementEvaluator::ComputeExpectedImprovement / ComputeGradExpectedImprovement (gpp_math.cpp:1991-2126):
//   V = Var(Xu) + 1e-6 I,  L = chol(V),  per sample: y = mu + L z,  I = max(0, max_j (best_so_far - y_j)),  w = argmax;
//   EI = sum I / M;   grad EI[k,:] = (1/M) sum_{I>0} ( -[w == k] grad mu_k  -  sum_j dL[w][j]/dXs_k z_j ).
// One lane per MC sample (the per-sample work is O(u^2), u <= 16); sums are block-reduced in a fixed order and finished
// by a single workgroup, so results are bitwise reproducible.
#include <cmath>
#include <cstring>

#include "device_cov.hpp"
#include "kg.hpp"

namespace moe {

namespace {

constexpr int kMaxUnionEi = 16;      // union size (q + p) up to which the u x u algebra of an evaluation runs on the device (ei_state_kernel)
constexpr int kMaxUnionEiWide = 64;  // r6: up to here the MC kernel is built (MU = 32 / 64); the u x u algebra of such a state runs on the host
                                     // (host_math.hip -- this library's own code; two waits per call instead of one)
#ifndef MOE_EI_PROF
#define MOE_EI_PROF 0
#endif
#if MOE_EI_PROF
__device__ unsigned long long g_ei_prof[16];
#define MOE_EI_T(i) if (threadIdx.x == 0 && blockIdx.x == 0) g_ei_prof[i] = __builtin_amdgcn_s_memtime()
#else
#define MOE_EI_T(i)
#endif

struct EiParams {
  int u, q, d, num_mc;
  double best_so_far;
  const double* mu;       // [u]
  const double* L;        // [u x u] col-major lower
  const double* grad_mu;  // [q][d]
  const double* gchol;    // [q][u][u][d]: gchol[k*d*u*u + dd + c*d + r*d*u] = dL[r][c]/dXs_{k,dd}, r >= c
  const double* normals;  // [num_mc][u]  (shared by all evaluations: common random numbers)
  double* partial;        // [E][gridDim.x][1 + q*d]
  int want_grad;
  long blob_stride;       // doubles between consecutive evaluations' (mu, L, grad_mu, gchol) records
  double* out;            // [E][1 + q*d]: the block sums added up by the LAST workgroup of an evaluation to finish (r4); or NULL
  unsigned int* ticket;   // [E] arrival counters, zero before the launch and zero again after it
};

__device__ __forceinline__ double wave_sum_ei(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// out[c] = sum over the workgroups of partial[b][c], one workgroup of 256 threads: component c is summed by the 256 / ncomp' lanes that
// share c = lane % ncomp' (block order within a lane, then a fixed-order tree over the lanes of the component): one pass and one barrier
// round instead of ncomp of them.  (Until r3 a launch of its own, sum_partials_kernel; the order of the additions is unchanged.)
__device__ __forceinline__ void sum_partials_body(const double* __restrict__ partial, int num_blocks, int ncomp,
                                                  double* __restrict__ out, double* __restrict__ red) {
  for (int c0 = 0; c0 < ncomp; c0 += 256) {
    const int nc = min(256, ncomp - c0);              // components of this round
    int stride = 1;
    while (stride * 2 * nc <= 256) stride *= 2;        // lanes per component (a power of two)
    const int comp = threadIdx.x % nc, part = threadIdx.x / nc;
    double acc = 0.0;
    if (part < stride)
      for (int b = part; b < num_blocks; b += stride)  // (device-scope loads: the sums come from other CUs, past this CU's L1)
        acc += __hip_atomic_load(&partial[(long)b * ncomp + c0 + comp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = stride / 2; off > 0; off >>= 1) {
      if (part < off) red[threadIdx.x] += red[threadIdx.x + off * nc];
      __syncthreads();
    }
    if (part == 0 && threadIdx.x < nc) out[c0 + comp] = red[threadIdx.x];
    __syncthreads();
  }
}

// MU: the union-size class the sample loop is unrolled for (16: every state the device algebra builds; 32 / 64: wider unions, whose
// u x u algebra runs on the host, r6)
template <int MU>
struct ei_mc_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const EiParams& P_in) {
    EiParams P = P_in;  // (the record pointers below are moved to this evaluation's record)
    __shared__ double red[4];
    __shared__ double red256[256];
    __shared__ int s_last;
    const long eoff = (long)blockIdx.y * P.blob_stride;  // this evaluation's record
    P.mu += eoff;
    P.L += eoff;
    P.grad_mu += eoff;
    P.gchol += eoff;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i < P.num_mc;
    const int u = P.u, d = P.d;
    const int ncomp = 1 + (P.want_grad ? P.q * d : 0);
    double z[MU];
    double imp = 0.0;
    int winner = u + 1;
    if (active) {
  #pragma unroll
      for (int j = 0; j < MU; ++j) z[j] = (j < u) ? P.normals[(long)i * u + j] : 0.0;
  #pragma unroll
      for (int r = 0; r < MU; ++r) {
        if (r < u) {
          double y = P.mu[r];
  #pragma unroll
          for (int c = 0; c < MU; ++c)
            if (c <= r) y = fma(P.L[r + c * u], z[c], y);
          const double t = P.best_so_far - y;
          if (t > imp) {
            imp = t;
            winner = r;
          }
        }
      }
    }
    for (int comp = 0; comp < ncomp; ++comp) {
      double contrib = 0.0;
      if (active && imp > 0.0) {
        if (comp == 0) {
          contrib = imp;
        } else {
          const int k = (comp - 1) / d, dd = (comp - 1) % d;
          double v = 0.0;
          if (winner == k) v = -P.grad_mu[k * d + dd];
          const double* g = P.gchol + (long)k * d * u * u + dd + (long)winner * d * u;
  #pragma unroll
          for (int j = 0; j < MU; ++j)
            if (j <= winner && j < u) v = fma(-g[j * d], z[j], v);
          contrib = v;
        }
      }
      const double w = wave_sum_ei(contrib);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
      __syncthreads();
      if (threadIdx.x == 0)
        P.partial[((long)blockIdx.y * gridDim.x + blockIdx.x) * ncomp + comp] = (red[0] + red[1]) + (red[2] + red[3]);
      __syncthreads();
    }
    if (P.out == nullptr) return;
    // the workgroup that arrives last adds the block sums up (r4: one launch less on the latency path; the order of the additions
    // is fixed, whoever performs them)
    if (threadIdx.x == 0) {
      __threadfence();
      const unsigned int t = atomicAdd(&P.ticket[blockIdx.y], 1u);
      s_last = (t == gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    sum_partials_body(P.partial + (long)blockIdx.y * gridDim.x * ncomp, (int)gridDim.x, ncomp, P.out + (long)blockIdx.y * ncomp, red256);
    if (threadIdx.x == 0) P.ticket[blockIdx.y] = 0u;
  }
};
template <int MU>
__global__ __launch_bounds__(256) void ei_mc_kernel(EiParams P) {
  ei_mc_kernel_body<MU>::run(MOE_VBLOCK, MOE_VGRID, nullptr, P);
}

// ---------------------------------------------------------------------------------------------------------------------
// The u x u algebra of an EI evaluation ON THE DEVICE (r3): what ei_evaluate_batch did on the host between two stream syncs --
// mean, Var + 1e-6 I, its Cholesky factor, grad mean, Smith's derivative of the factor (host_math.hip: host_mean, host_variance,
// host_cholesky, host_grad_mean, host_grad_cholesky_per_point; EI points carry no derivative observations) -- from the Gram
// matrix and E^T K^-1 y the state kernels leave in device memory, straight into the record the MC kernel reads.  One wavefront
// per evaluation; lane = row for the factorisation, lane = dimension for the factor's derivative (every dimension is an
// independent Smith recursion).  An evaluation then needs ONE sync and no mid-way round trip through the host.  The same
// formulas as the host code with the device's exp / sqrt, whose results differ from libm's in the last bit: EI agrees with the
// host-algebra path (MOE_EI_DEVICE_ALGEBRA=0) to ~1e-15 (tests/test_gpu_parity.py).  What it buys at C2, one evaluation at a time:
// value only 80 -> 71 us, value + gradient 93 -> 93 (profiles/r03_n_latency.txt) -- a kernel trace shows why it is not more: the
// GPU is busy for the whole call, nine back-to-back kernels of 3 - 18 us each (the 500 x 500 triangular product alone 18), so
// the call is bound by the small kernels' own latencies, not by launches or syncs.
// ---------------------------------------------------------------------------------------------------------------------
struct EiStateParams {
  CovParams cp;
  double mean;
  const double* gram;  // [E][c][c] col-major, c = u + nd d
  const double* ek;    // [ctot]: K* block of evaluation e at e u, gradient block at E u + e nd d
  const double* U;     // [E][u][dp] padded union points
  int E, u, nd, d, dp, want_grad;
  double* blob;        // records: mu [u] | L [u x u] | grad_mu [nd d] | gchol [nd][u][u][d]
  long rec, o_mu, o_L, o_gmu, o_gc;
  double* flags;       // [E]: 0, or the failing leading minor of the variance matrix (doubles: they travel behind the results in
                       // the call's one device->host copy)
  // small states (fused != 0): gram / ek above are not read -- the kernel forms them from V = L^-1 E and E (N rows, ld N, columns
  // grouped by kind) and K^-1 (y - mean)
  int fused, N;
  const double *V, *Emat, *KinvY;
};

// d chol(Var + 1e-6 I) / d Xs_{k, dd} for one differentiated point k and one dimension dd: host_grad_variance_per_point (no
// derivative observations) followed by Smith's recursion (gpp_math.cpp:1389-1452), in a lane-private array; out[j u d + i d + dd] =
// d chol[j][i] / d x for i <= j (first index i, second j, as host_math.hip's GC macro).
template <int UM>
__device__ __forceinline__ void ei_grad_chol_lane(const EiStateParams& P, const double* __restrict__ G, const double* __restrict__ U,
                                                  const double* __restrict__ Ls, int c, int k, int dd, double* __restrict__ out) {
  const int u = P.u, d = P.d;
  const double kMinimumStdDev = 2.220446049250313e-16;  // gpp_math.hpp:291
  double gc[UM * UM];
#define MOE_GC(i, j) gc[(j)*UM + (i)]
#pragma unroll
  for (int j = 0; j < UM; ++j)
#pragma unroll
    for (int i = 0; i < UM; ++i) MOE_GC(i, j) = 0.0;
  // grad variance: column block k, the (k, k) entry, d Kss / d Xs_k
  const int cg = u + k * d + dd;
#pragma unroll
  for (int row = 0; row < UM; ++row)
    if (row < u) {
      const double v = -G[cg + (long)row * c];
#pragma unroll
      for (int kk = 0; kk < UM; ++kk)
        if (kk == k) MOE_GC(row, kk) = v;
    }
#pragma unroll
  for (int j = 0; j < UM; ++j)
    if (j < u) {
      double r2 = 0.0, dfd = 0.0;
      for (int kk = 0; kk < d; ++kk) {
        const double df = U[k * P.dp + kk] - U[j * P.dp + kk];
        r2 = fma(df * df, P.cp.inv_l2[kk], r2);
        if (kk == dd) dfd = df;
      }
      const double tmp = (-dfd * P.cp.inv_l2[dd]) * radial_scalars(P.cp.type, P.cp.alpha, r2).first;  // grad_cov_entry(0, 0, dd)
#pragma unroll
      for (int kk = 0; kk < UM; ++kk)
        if (kk == k) {
          if (j == k)
            MOE_GC(j, kk) = (MOE_GC(j, kk) + MOE_GC(j, kk)) + (tmp + tmp);  // (the (k, k) entry: both factors depend on Xs_k)
          else
            MOE_GC(j, kk) += tmp;
        }
    }
  // mirror block column k into block row k, keep first <= second
#pragma unroll
  for (int j = 0; j < UM; ++j)
#pragma unroll
    for (int kk = 0; kk < UM; ++kk)
      if (kk == k && j != k) MOE_GC(kk, j) = MOE_GC(j, kk);
#pragma unroll
  for (int i = 0; i < UM; ++i)
#pragma unroll
    for (int i2 = i + 1; i2 < UM; ++i2) MOE_GC(i2, i) = 0.0;
#define MOE_CH(i, j) Ls[(i) + (j)*u]
#pragma unroll
  for (int kk = 0; kk < UM; ++kk)
    if (kk < u) {
      const double Lkk = MOE_CH(kk, kk);
      if (Lkk > kMinimumStdDev) {
        MOE_GC(kk, kk) = 0.5 * MOE_GC(kk, kk) / Lkk;
#pragma unroll
        for (int j = kk + 1; j < UM; ++j)
          if (j < u) MOE_GC(kk, j) = (MOE_GC(kk, j) - MOE_CH(j, kk) * MOE_GC(kk, kk)) / Lkk;
#pragma unroll
        for (int j = kk + 1; j < UM; ++j)
#pragma unroll
          for (int i = j; i < UM; ++i)
            if (i < u) MOE_GC(j, i) = MOE_GC(j, i) - MOE_GC(kk, i) * MOE_CH(j, kk) - MOE_CH(i, kk) * MOE_GC(kk, j);
      }
    }
#undef MOE_CH
#pragma unroll
  for (int j = 0; j < UM; ++j)
#pragma unroll
    for (int i = 0; i < UM; ++i)
      if (i < u && j < u) out[(long)j * u * d + (long)i * d + dd] = MOE_GC(i, j);
#undef MOE_GC
}

// r4, small states (FUSED: the evaluation's columns of V = L^-1 E and of E fit LDS -- every q-EI call at C2): the Gram matrix V^T V and
// E^T K^-1 (y - mean) are formed HERE, by the evaluation's own workgroup, from ONE batch of loads -- every thread issues its share of
// the N x c entries of V and E at once -- instead of by the K-sliced Gram kernel, its slice sum and a gemv: three launches whose K loops
// paid a memory round trip of 2 - 3 us per stage (7.5 + 4.7 + 5.4 us at C2).  An output is one wavefront's lane-strided dot product
// and a fixed butterfly: deterministic, and a function of the evaluation alone.
// (UM: the union-size class the factor-derivative recursion is unrolled for -- one kernel per class: with all three classes in one
//  kernel the register allocation and the scratch spills of the 16 x 16 class were paid by every call)
template <bool FUSED, int UM>
struct ei_state_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const EiStateParams& P) {
    __shared__ double Ls[kMaxUnionEi * kMaxUnionEi];
    __shared__ int s_bad;
    extern __shared__ __attribute__((aligned(16))) double ei_sm[];  // FUSED: Gs [c][c] | eks [c] | ys [N] | Vs [c][N] | Es [c][N]
    __shared__ double Us[kMaxUnionEi * kMaxDimPadded];
    const int e = blockIdx.x, lane = threadIdx.x;
    const int u = P.u, d = P.d, c = u + P.nd * d;
    const double* G = P.gram + (long)e * c * c;
    const double* ek_k = P.ek + (long)e * u;
    const double* ek_g = P.ek + (long)P.E * u + (long)e * P.nd * d;
    MOE_EI_T(0);
    if constexpr (FUSED) {
      const int N = P.N, ng = P.nd * d;
      double* Gs = ei_sm;
      double* eks = Gs + c * c;
      double* ys = eks + c;
      double* Vs = ys + N;
      double* Es = Vs + (long)c * N;
      // every global load of a batch is issued before the first LDS store (a store straight behind its load makes each step wait for
      // its own round trip of 2 - 3 us); the first batch carries the union points and K^-1 (y - mean) along, so a state of up to
      // 256 x 24 entries -- C2 with its gradient columns: 5000 -- costs ONE round trip
      constexpr int NB = 24;
      double uv[2], yv[4];
  #pragma unroll
      for (int i = 0; i < 2; ++i) uv[i] = P.U[(long)e * u * P.dp + min(lane + 256 * i, u * P.dp - 1)];
  #pragma unroll
      for (int i = 0; i < 4; ++i) yv[i] = P.KinvY[min(lane + 256 * i, N - 1)];
      for (int t0 = 0; t0 < N * c; t0 += 256 * NB) {
        double va[NB], ea[NB];
  #pragma unroll
        for (int i = 0; i < NB; ++i) {
          const int t = min(t0 + i * 256 + lane, N * c - 1);
          const int l = t / N, k = t - l * N;
          const long col = (l < u) ? ((long)e * u + l) : ((long)P.E * u + (long)e * ng + (l - u));  // columns grouped by kind (BatchLayout)
          va[i] = P.V[k + col * N];
          ea[i] = P.Emat[k + col * N];
        }
  #pragma unroll
        for (int i = 0; i < NB; ++i) {
          const int t = t0 + i * 256 + lane;
          if (t < N * c) {
            Vs[t] = va[i];
            Es[t] = ea[i];
          }
        }
      }
  #pragma unroll
      for (int i = 0; i < 2; ++i)
        if (lane + 256 * i < u * P.dp) Us[lane + 256 * i] = uv[i];
  #pragma unroll
      for (int i = 0; i < 4; ++i)
        if (lane + 256 * i < N) ys[lane + 256 * i] = yv[i];
      for (int k = lane + 1024; k < N; k += 256) ys[k] = P.KinvY[k];
      __syncthreads();
      MOE_EI_T(1);
      // one output per 8-lane group at a time (32 groups): lanes stride k with two accumulators, three in-group butterfly steps; the
      // outputs are enumerated without gaps (the pairs, then the c entries of ek)
      // Only the entries the algebra below reads: G(K*, K*) and G(dK*, K*) -- pairs (i <= j) with i < u; the dK* x dK* block (36 of the 55
      // pairs at C2) is never used.
      const int tri_u = u * (u + 1) / 2;
      const int npair = tri_u + (c - u) * u;
      const int gid = lane >> 3, gl = lane & 7;
      for (int o = gid; o < npair + c; o += 32) {
        const double *a, *b;
        int i = 0, j = 0;
        if (o < npair) {
          if (o < tri_u) {
            int rem = o;
            while (rem > j) {  // column j of the upper triangle holds j + 1 pairs
              rem -= j + 1;
              ++j;
            }
            i = rem;
          } else {
            const int o2 = o - tri_u;
            j = u + o2 / u;
            i = o2 - (j - u) * u;
          }
          a = Vs + (long)i * N;
          b = Vs + (long)j * N;
        } else {
          i = o - npair;
          a = Es + (long)i * N;
          b = ys;
        }
        // (eight LDS reads per operand in flight: one read per step would make the loop a chain of LDS latencies)
        double acc0 = 0.0, acc1 = 0.0;
        int k = gl;
        for (; k + 7 * 8 < N; k += 8 * 8) {
          double av[8], bv[8];
  #pragma unroll
          for (int t = 0; t < 8; ++t) {
            av[t] = a[k + 8 * t];
            bv[t] = b[k + 8 * t];
          }
  #pragma unroll
          for (int t = 0; t < 8; t += 2) {
            acc0 = fma(av[t], bv[t], acc0);
            acc1 = fma(av[t + 1], bv[t + 1], acc1);
          }
        }
        for (; k < N; k += 8) acc0 = fma(a[k], b[k], acc0);
        double acc = acc0 + acc1;
  #pragma unroll
        for (int off = 4; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 8);
        if (gl == 0) {
          if (o < npair) {
            Gs[i + j * c] = acc;
            Gs[j + i * c] = acc;
          } else {
            eks[i] = acc;
          }
        }
      }
      __syncthreads();
      MOE_EI_T(2);
      G = Gs;
      ek_k = eks;
      ek_g = eks + u;
    }
    // the union points in LDS: the distance loops below (runtime trip count d, two loads per step) otherwise pay a memory round trip
    // per step -- 16 of them in a row in the factor-derivative part at C2
    if constexpr (!FUSED) {
      for (int t = lane; t < u * P.dp; t += 256) Us[t] = P.U[(long)e * u * P.dp + t];
      __syncthreads();
    }
    const double* U = Us;
    double* r = P.blob + (long)e * P.rec;
    if (lane == 0) s_bad = 0;
    if (lane < u) r[P.o_mu + lane] = P.mean + ek_k[lane];  // host_mean
    for (int idx = lane; idx < u * u; idx += 256) {         // host_variance, + 1e-6 on the diagonal (gpp_math.cpp:2000-2002)
      const int i = idx % u, j = idx / u;
      double r2 = 0.0;
      for (int k = 0; k < d; ++k) {
        const double df = U[i * P.dp + k] - U[j * P.dp + k];
        r2 = fma(df * df, P.cp.inv_l2[k], r2);
      }
      double v = radial_scalars(P.cp.type, P.cp.alpha, r2).base - G[i + (long)j * c];
      if (i == j) v += 1.0e-6;
      Ls[i + j * u] = v;
    }
    __syncthreads();
    MOE_EI_T(3);
    // host_cholesky (ComputeCholeskyFactorL, gpp_linear_algebra.cpp:109-148): lane = row
    for (int k = 0; k < u; ++k) {
      const double akk = Ls[k + k * u];
      if (!(akk > 1.0e-16)) {
        if (lane == 0) s_bad = k + 1;
        break;
      }
      const double lkk = sqrt(akk);
      __syncthreads();
      if (lane == k) Ls[k + k * u] = lkk;
      if (lane > k && lane < u) Ls[lane + k * u] = Ls[lane + k * u] / lkk;
      __syncthreads();
      for (int j = k + 1; j < u; ++j)
        if (lane >= j && lane < u) Ls[lane + j * u] = Ls[lane + j * u] - Ls[lane + k * u] * Ls[j + k * u];
      __syncthreads();
    }
    __syncthreads();
    if (lane == 0) P.flags[e] = (double)s_bad;
    if (s_bad != 0) return;
    for (int idx = lane; idx < u * u; idx += 256) {
      const int i = idx % u, j = idx / u;
      r[P.o_L + idx] = (j <= i) ? Ls[idx] : 0.0;
    }
    MOE_EI_T(4);
    if (!P.want_grad) return;
    for (int idx = lane; idx < P.nd * d; idx += 256) r[P.o_gmu + idx] = ek_g[idx];  // host_grad_mean
    // host_grad_cholesky_per_point for every differentiated point k; lane = dimension dd (independent recursions), each in a
    // lane-private u x u array (registers for u <= 4 / 8) written to the record once
    // (r4: one lane per (point, dimension) pair -- nd d independent recursions -- instead of one per dimension looping over the points)
    for (int t = lane; t < P.nd * d; t += 256) {
      const int k = t / d, dd = t - k * d;
      double* gc = r + P.o_gc + (long)k * d * u * u;
      ei_grad_chol_lane<UM>(P, G, U, Ls, c, k, dd, gc);
    }
    MOE_EI_T(5);
  }
};
template <bool FUSED, int UM>
__global__ __launch_bounds__(256) void ei_state_kernel(EiStateParams P) {
  ei_state_kernel_body<FUSED, UM>::run(MOE_VBLOCK, MOE_VGRID, nullptr, P);
}

}  // namespace

bool ei_device_algebra() {  // MOE_EI_DEVICE_ALGEBRA=0: the host-algebra path (two syncs per call; A/B runs and tests)
  const char* v = std::getenv("MOE_EI_DEVICE_ALGEBRA");
  return !(v && *v == '0');
}

// r6: the evaluation in two halves -- everything that is ENQUEUED (recordable: launch.hpp), and the collection of the results once the
// stream has run -- so that the members of a GP ensemble can share their launches (mcmc.hip: ei_mcmc_batch).
EiPending ei_launch(GpDev& gp, const double* Xq_all, int num_evals, const double* Xp, int q, int p, int num_mc, double best_so_far,
                    const double* normals, bool want_value, bool want_grad) {
  gp.use_device();
  hipStream_t s = gp.stream;
  const int d = gp.d, u = q + p, E = num_evals;
  if (q <= 0) throw Error(MOE_ERR_BOUNDS, "num_to_sample must be positive", q, 1, 1e9);
  if (p < 0) throw Error(MOE_ERR_BOUNDS, "num_being_sampled must be non-negative", p, 0, 1e9);
  if (E <= 0) throw Error(MOE_ERR_BOUNDS, "num_evals must be positive", E, 1, 1e9);
  if (u > kMaxUnionEiWide) throw Error(MOE_ERR_BOUNDS, "q + p > 64 is not supported by the device kernels", u, 1, kMaxUnionEiWide);
  if (num_mc <= 0) throw Error(MOE_ERR_BOUNDS, "num_mc must be positive", num_mc, 1, 1e12);
  std::vector<double> U_all((size_t)E * u * d);
  for (int e = 0; e < E; ++e) {
    std::copy(Xq_all + (size_t)e * q * d, Xq_all + (size_t)(e + 1) * q * d, &U_all[(size_t)e * u * d]);
    if (p > 0) std::copy(Xp, Xp + (size_t)p * d, &U_all[(size_t)e * u * d + (size_t)q * d]);
  }
  DerivList none;
  none.g = 0;
  for (int i = 0; i < kMaxDerivs; ++i) none.idx[i] = 0;
  // EI points carry no derivative observations even when the GP does (ExpectedImprovementState, gpp_math.cpp:2149-2150)
  // per-evaluation record: mu [u] | L [u*u] | grad_mu [q*d] | gchol [q*d*u*u]
  const size_t o_mu = 0, o_L = u, o_gmu = o_L + (size_t)u * u, o_gc = o_gmu + (size_t)q * d;
  const size_t rec = o_gc + (want_grad ? (size_t)q * d * u * u : 0);
  const size_t n_norm = (size_t)num_mc * u;
  const bool on_device = ei_device_algebra() && u <= kMaxUnionEi;
  const int ncomp = 1 + (want_grad ? q * d : 0);
  const int blocks = (num_mc + 255) / 256;
  DevBuf<double>& dBlobDev = gp.kBlob;
  DevBuf<double>&dPartial = gp.kTB, &dOut = gp.kOut;
  dPartial.reserve((size_t)E * blocks * ncomp);
  dOut.reserve((size_t)E * ncomp + (size_t)E);  // results | singular-matrix flags (doubles): one device->host copy
  const double* d_normals = nullptr;
  double* blob = nullptr;
  if (on_device) {
    // ---- the whole evaluation stays on the device: state kernels -> u x u algebra kernel -> MC (+ final sum) -> ONE sync; r4: the
    // normal draws ride down with the points in the state set-up's copy, the flags ride up with the results (5 copies -> 2) ----
    StateAppendix apx;
    apx.doubles = n_norm;
    apx.fill = [&](double* dst) { std::memcpy(dst, normals, sizeof(double) * n_norm); };
    const StateEnqueued se = enqueue_state_batch(gp, U_all.data(), u, none, want_grad ? q : 0, nullptr, 0, false, E, &apx, true);
    d_normals = gp.dAppendix;
    dBlobDev.reserve(rec * E);
    EiStateParams sp;
    sp.cp = gp.cp;
    sp.mean = gp.mean;
    sp.gram = gp.dGram.p;
    sp.ek = gp.dGram.p + se.nG;
    sp.U = gp.dUnion;
    sp.E = E;
    sp.u = u;
    sp.nd = want_grad ? q : 0;
    sp.d = d;
    sp.dp = gp.dp;
    sp.want_grad = want_grad ? 1 : 0;
    sp.blob = dBlobDev.p;
    sp.rec = (long)rec;
    sp.o_mu = (long)o_mu;
    sp.o_L = (long)o_L;
    sp.o_gmu = (long)o_gmu;
    sp.o_gc = (long)o_gc;
    sp.flags = dOut.p + (size_t)E * ncomp;
    sp.fused = se.fused ? 1 : 0;
    sp.N = gp.N;
    sp.V = gp.dVE.p;
    sp.Emat = gp.dE.p;
    sp.KinvY = gp.dKinvY.p;
    const int cst = u + (want_grad ? q : 0) * d;
    const size_t shm = se.fused ? sizeof(double) * ((size_t)cst * cst + cst + gp.N + 2 * (size_t)cst * gp.N) : 0;
#define MOE_EI_STATE(FUSED, UM)                                                                                                      \
  {                                                                                                                                \
    auto kern = ei_state_kernel<FUSED, UM>;                                                                                        \
    if (shm > 48 * 1024)                                                                                                           \
      MOE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm)); \
    launch_kernel_ens<ei_state_kernel_body<FUSED, UM>, 256>(kern, dim3(E), dim3(256), shm, s, sp);                                 \
  }
    if (se.fused) {
      if (u <= 4) MOE_EI_STATE(true, 4)
      else if (u <= 8) MOE_EI_STATE(true, 8)
      else MOE_EI_STATE(true, kMaxUnionEi)
    } else {
      if (u <= 4) MOE_EI_STATE(false, 4)
      else if (u <= 8) MOE_EI_STATE(false, 8)
      else MOE_EI_STATE(false, kMaxUnionEi)
    }
#undef MOE_EI_STATE
    MOE_HIP_CHECK(hipGetLastError());
  } else {
  gp.hKgIn.reserve(rec * E + n_norm);
  blob = gp.hKgIn.p;
  std::vector<StateHost> hosts;
  compute_state_batch(gp, U_all.data(), u, none, want_grad ? q : 0, nullptr, 0, false, E, nullptr, &hosts);
  for (int e = 0; e < E; ++e) {
    const StateHost& sh = hosts[e];
    double* r = blob + rec * e;
    double* chol = r + o_L;
    host_mean(sh, r + o_mu);
    host_variance(sh, chol);
    for (int i = 0; i < u; ++i) chol[i + (size_t)i * u] += 1.0e-6;  // gpp_math.cpp:2000-2002
    const int lm = host_cholesky(u, chol);
    if (lm != 0)
      throw Error(MOE_ERR_SINGULAR,
                  "GP-Variance matrix singular. Check for duplicate points_to_sample/being_sampled or "
                  "points_to_sample/being_sampled duplicating points_sampled with 0 noise.",
                  u, lm);
    if (want_grad) {
      host_grad_mean(sh, r + o_gmu);
      for (int k = 0; k < q; ++k) host_grad_cholesky_per_point(sh, k, chol, r + o_gc + (size_t)k * d * u * u);
    }
  }
  std::memcpy(blob + rec * E, normals, sizeof(double) * n_norm);
  gp.kBlob.upload(blob, rec * E + n_norm, s);
  d_normals = gp.kBlob.p + rec * E;
  }
  // persistent workspaces (hipMalloc / hipFree per call cost more than the whole evaluation)
  DevBuf<double>& dBlob = gp.kBlob;
  // arrival counters of ei_mc_kernel: zero at allocation, left zero by every launch that COMPLETES; a launch that was enqueued and
  // never seen to finish (a device fault, a sticky error from an earlier kernel: the wait below throws) leaves them undefined, and a
  // non-zero counter would keep every later call from electing its last workgroup -- so such a call is followed by a clear
  if (gp.kEiTicket.cap < (size_t)E || gp.ei_ticket_dirty) {
    gp.kEiTicket.reserve((size_t)E);
    memset_async(gp.kEiTicket.p, 0, sizeof(unsigned int) * gp.kEiTicket.cap, s);
  }
  gp.ei_ticket_dirty = true;
  EiParams P;
  P.u = u;
  P.q = q;
  P.d = d;
  P.num_mc = num_mc;
  P.best_so_far = best_so_far;
  P.mu = dBlob.p + o_mu;
  P.L = dBlob.p + o_L;
  P.grad_mu = dBlob.p + o_gmu;
  P.gchol = dBlob.p + o_gc;
  P.normals = d_normals;
  P.partial = dPartial.p;
  P.want_grad = want_grad ? 1 : 0;
  P.blob_stride = (long)rec;
  P.out = dOut.p;
  P.ticket = gp.kEiTicket.p;
  if (u <= 16)
    launch_kernel_ens<ei_mc_kernel_body<16>, 256>(ei_mc_kernel<16>, dim3(blocks, E), dim3(256), 0, s, P);
  else if (u <= 32)
    launch_kernel_ens<ei_mc_kernel_body<32>, 256>(ei_mc_kernel<32>, dim3(blocks, E), dim3(256), 0, s, P);
  else
    launch_kernel_ens<ei_mc_kernel_body<64>, 256>(ei_mc_kernel<64>, dim3(blocks, E), dim3(256), 0, s, P);
  MOE_HIP_CHECK(hipGetLastError());
  const size_t n_down = (size_t)E * ncomp + (on_device ? (size_t)E : 0);
  gp.hKgOut.reserve(n_down);
  double* out = gp.hKgOut.p;
  copy_async(out, dOut.p, sizeof(double) * n_down, hipMemcpyDeviceToHost, s, true);  // (hKgOut: pinned)
  GpDev* gpp = &gp;
  EiPending pending;
  pending.collect = [=](double* ei, double* grad_ei) {
  GpDev& gp = *gpp;
  gp.use_device();
  MOE_HIP_CHECK(hipStreamSynchronize(s));
  gp.ei_ticket_dirty = false;
#if MOE_EI_PROF
  {
    unsigned long long h[16];
    MOE_HIP_CHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ei_prof), sizeof(h)));
    std::fprintf(stderr, "[ei prof] ticks (100 MHz): load %llu dots %llu var %llu chol+L %llu grad %llu\n", h[1] - h[0], h[2] - h[1], h[3] - h[2],
                 h[4] - h[3], want_grad ? h[5] - h[4] : 0ull);
  }
#endif
  if (on_device)
    for (int e = 0; e < E; ++e)
      if (out[(size_t)E * ncomp + e] != 0.0)
        throw Error(MOE_ERR_SINGULAR,
                    "GP-Variance matrix singular. Check for duplicate points_to_sample/being_sampled or "
                    "points_to_sample/being_sampled duplicating points_sampled with 0 noise.",
                    u, out[(size_t)E * ncomp + e]);
  for (int e = 0; e < E; ++e) {
    if (ei) ei[e] = out[(size_t)e * ncomp] / (double)num_mc;
    if (want_grad && grad_ei)
      for (int c = 0; c < q * d; ++c) grad_ei[(size_t)e * q * d + c] = out[(size_t)e * ncomp + 1 + c] / (double)num_mc;
  }
  };
  (void)want_value;
  return pending;
}

void ei_evaluate_batch(GpDev& gp, const double* Xq_all, int num_evals, const double* Xp, int q, int p, int num_mc,
                       double best_so_far, const double* normals, double* ei, double* grad_ei) {
  ei_launch(gp, Xq_all, num_evals, Xp, q, p, num_mc, best_so_far, normals, ei != nullptr, grad_ei != nullptr).collect(ei, grad_ei);
}

void ei_evaluate(GpDev& gp, const double* Xq, const double* Xp, int q, int p, int num_mc, double best_so_far,
                 const double* normals, double* ei, double* grad_ei) {
  ei_evaluate_batch(gp, Xq, 1, Xp, q, p, num_mc, best_so_far, normals, ei, grad_ei);
}

// OnePotentialSampleExpectedImprovementEvaluator (gpp_math.cpp:2195-2259) for `num_evals` single points: the posterior at
// one point is a scalar Gaussian, so EI = (best - mu) Phi(c) + sigma phi(c), c = (best - mu) / sigma.  The N-sized work
// (K*, K^-1 K*, grad K*) for ALL points runs in one batched device pass; what is left per point is scalar.
void ei_analytic_batch(GpDev& gp, const double* pts, int num_evals, double best_so_far, double* ei, double* grad_ei) {
  gp.use_device();
  const int d = gp.d, E = num_evals;
  if (E <= 0) throw Error(MOE_ERR_BOUNDS, "num_evals must be positive", E, 1, 1e9);
  const bool want_grad = grad_ei != nullptr;
  DerivList none;
  none.g = 0;
  for (int i = 0; i < kMaxDerivs; ++i) none.idx[i] = 0;
  std::vector<StateHost> hosts;
  compute_state_batch(gp, pts, 1, none, want_grad ? 1 : 0, nullptr, 0, false, E, nullptr, &hosts);
  constexpr double kMinVarEI = 2.2250738585072014e-308;                                    // gpp_math.hpp:1316
  constexpr double kMinVarGradEI = 150.0 * 2.220446049250313e-16 * 2.220446049250313e-16;  // gpp_math.hpp:1323
  auto pdf = [](double z) { return std::exp(-0.5 * z * z) / 2.5066282746310002; };
  auto cdf = [](double z) { return 0.5 * std::erfc(-z * 0.70710678118654752440); };
  std::vector<double> gmu(d), gchol(d);
  for (int e = 0; e < E; ++e) {
    const StateHost& sh = hosts[e];
    double mu, var;
    host_mean(sh, &mu);
    host_variance(sh, &var);
    const double t = best_so_far - mu;
    if (ei) {
      const double sigma = std::sqrt(std::fmax(kMinVarEI, var));
      ei[e] = std::fmax(0.0, t * cdf(t / sigma) + sigma * pdf(t / sigma));
    }
    if (want_grad) {
      const double v = std::fmax(kMinVarGradEI, var);
      double sigma = std::sqrt(v);
      host_grad_mean(sh, gmu.data());
      host_grad_cholesky_per_point(sh, 0, &sigma, gchol.data());
      const double c = t / sigma, pdf_c = pdf(c), cdf_c = cdf(c);
      for (int i = 0; i < d; ++i) {
        const double d_c = (-sigma * gmu[i] - gchol[i] * t) / v;
        const double d_a = -gmu[i] * cdf_c + t * pdf_c * d_c;
        const double d_b = gchol[i] * pdf_c + sigma * (-c) * pdf_c * d_c;
        grad_ei[(size_t)e * d + i] = d_a + d_b;
      }
    }
  }
}

}  // namespace moe
