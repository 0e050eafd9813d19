// cornell_moe_amd/csrc/kg_state.hip -- see kg_state.hpp: PreCompute of a KnowledgeGradientState, Smith's derivative of the factor and
// the final assembly of grad KG as gfx950 kernels.  Every matrix here is m x m with m = (q + p)(1 + g) <= 128: it lives in LDS, a
// workgroup owns one evaluation (or one (evaluation, point, coordinate) for the derivative of the factor), and the elimination orders are
// the reference's (ComputeCholeskyFactorL, gpp_linear_algebra.cpp:109-148; the recursion of gpp_math.cpp:1389-1452).
#include "kg_state.hpp"

#include <algorithm>

#include "device_cov.hpp"

namespace moe {

namespace {

constexpr size_t kLdsBudget = 150 * 1024;  // of the 160 KB per CU

__device__ __forceinline__ double wave_sum64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ double pts_cov(const CovParams& cp, const double* p1, const double* p2, int d, int a, int b,
                                          const DerivList& d1, const DerivList& d2) {
  const PointDiff df{p1, p2};
  const Radial rd = pair_radial(cp, df, d);
  return cov_entry_g(cp, rd, df, a, b, d1, d2);
}

// Entry `idx` of evaluation e's Gram block of `cc` entries: the plain value (slices == 1, base = [E][cc]) or the sum of the K-slice
// partials the Gram kernel left (base = [E][slices][cc]) in slice order -- the sum gram_sum_kernel would have taken (r5).
struct GramView {
  const double* base;  // already offset to evaluation e
  int slices;
  long cc;
  __device__ __forceinline__ double operator[](long idx) const {
    if (slices == 1) return base[idx];
    // (at most 16 slices: the loads go out together from clamped addresses, the adds keep the slice order)
    double t[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) t[k] = base[(long)min(k, slices - 1) * cc + idx];
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (k < slices) v += t[k];
    return v;
  }
};
__device__ __forceinline__ GramView gram_view(const double* g, int slices, long cc, int e) {
  return GramView{g + (long)e * slices * cc, slices, cc};
}

// One workgroup per evaluation.  Dynamic LDS: Ls [m][m] (column-major) | Rs [m][rchunk] (right-hand sides of the discretised set).
struct kg_state_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const KgStateParams& P, int rchunk) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int e = blockIdx.x, tid = threadIdx.x;
    const int m = P.m, g1 = 1 + P.g, u = P.u, d = P.d, dp = P.dp, A = P.A, R = P.ng + P.A;
    double* Ls = sm;
    double* Rs = sm + m * m;
    const double* U = P.U + (long)e * u * dp;
    const double* X = P.extra + (long)e * A * dp;
    const GramView gkk = gram_view(P.gkk, P.gkk_slices, (long)m * m, e);
    const GramView gx = gram_view(P.gx, P.gx_slices, (long)R * m, e);
    const double* ek_k = P.ek + (long)e * m;
    const double* ek_g = P.ek + (long)P.E * m + (long)e * P.ng;
    const double* ek_x = P.ek + (long)P.E * (m + P.ng) + (long)e * A;
    double* r = P.blob + (long)e * P.rec_stride;
    // Var(Xu) + noise  (ComputeVarianceOfPoints, gpp_math.cpp:924-970; .cpp:304-309)
    for (int idx = tid; idx < m * m; idx += 256) {
      const int row = idx % m, col = idx / m;
      const int i = row / g1, a = row - i * g1, j = col / g1, b = col - j * g1;
      double v = pts_cov(P.cp, U + i * dp, U + j * dp, d, a, b, P.derivs, P.derivs) - gkk[idx];
      if (row == col) v += P.noise[a];
      Ls[idx] = v;
    }
    __syncthreads();
    // ComputeCholeskyFactorL: outer-product form, pivot rule 1e-16
    int bad = 0;
    for (int k = 0; k < m; ++k) {
      const double akk = Ls[k + k * m];
      if (!(akk > 1.0e-16)) {
        bad = k + 1;
        break;
      }
      const double lkk = sqrt(akk);
      __syncthreads();
      if (tid == 0) Ls[k + k * m] = lkk;
      for (int i = k + 1 + tid; i < m; i += 256) Ls[i + k * m] = Ls[i + k * m] / lkk;
      __syncthreads();
      const int rem = m - k - 1;
      for (int t = tid; t < rem * rem; t += 256) {
        const int ii = t % rem, jj = t / rem;
        if (ii >= jj) {
          const int i = k + 1 + ii, j = k + 1 + jj;
          Ls[i + j * m] = Ls[i + j * m] - Ls[i + k * m] * Ls[j + k * m];
        }
      }
      __syncthreads();
    }
    __syncthreads();
    if (tid == 0) P.flags[e] = bad;
    if (bad != 0) {  // reported by the host after the call's only wait; the kernels behind this one get a finite (identity) factor
      for (int idx = tid; idx < m * m; idx += 256) Ls[idx] = (idx % m == idx / m) ? 1.0 : 0.0;
      __syncthreads();
    }
    for (int idx = tid; idx < m * m; idx += 256) r[P.rec_L + idx] = (idx % m >= idx / m) ? Ls[idx] : 0.0;
    // best posterior mean among the points being sampled, function values only (.cpp:146-154)
    if (tid == 0) {
      double best = P.best_so_far;
      int w = -1;
      for (int j = 0; j < u; ++j) {
        const double mu = P.mean + ek_k[j * g1];
        if (mu < best) {
          best = mu;
          w = j;
        }
      }
      r[P.rec_bp] = best;
      P.winner[e] = w;
    }
    for (int j = tid; j < A; j += 256) r[P.rec_mu_disc + j] = P.mean + ek_x[j];
    for (int idx = tid; idx < P.q * d && P.ng > 0; idx += 256) {
      const int k = idx / d, dd = idx - k * d;
      P.gmu[(long)e * P.q * d + idx] = ek_g[(k * g1) * d + dd];  // ComputeGradMeanOfPoints, function-value rows (.cpp:136-140)
    }
    // discretised set: c_j = L^-1 cov_n(Xu, x_j), `rchunk` right-hand sides at a time; column k of the forward substitution for all of
    // them at once
    DerivList none;
    none.g = 0;
    for (int j0 = 0; j0 < A; j0 += rchunk) {
      const int cnt = min(rchunk, A - j0);
      __syncthreads();
      for (int t = tid; t < cnt * m; t += 256) {
        const int c = t % m, jj = t / m;
        const int i = c / g1, b = c - i * g1;
        Rs[t] = pts_cov(P.cp, U + i * dp, X + (long)(j0 + jj) * dp, d, b, 0, P.derivs, none) - gx[(P.ng + j0 + jj) + (long)c * R];
      }
      __syncthreads();
      for (int k = 0; k < m; ++k) {
        const double lkk = Ls[k + k * m];
        for (int jj = tid; jj < cnt; jj += 256) Rs[k + jj * m] = Rs[k + jj * m] / lkk;
        __syncthreads();
        const int rem = m - k - 1;
        for (int t = tid; t < rem * cnt; t += 256) {
          const int ii = t % rem, jj = t / rem;
          const int i = k + 1 + ii;
          Rs[i + jj * m] = Rs[i + jj * m] - Rs[k + jj * m] * Ls[i + k * m];
        }
        __syncthreads();
      }
      for (int t = tid; t < cnt * m; t += 256) r[P.rec_C_disc + (long)j0 * m + t] = Rs[t];
    }
  }
};
__global__ __launch_bounds__(256) void kg_state_kernel(KgStateParams P, int rchunk) {
  kg_state_kernel_body::run(MOE_VBLOCK, MOE_VGRID, nullptr, P, rchunk);
}

// d Var(row, col) / d Xq_p,dd for col in block p, row = (j, a)  (ComputeGradVarianceOfPointsPerPoint, gpp_math.cpp:1267-1357)
__device__ __forceinline__ double grad_var_entry(const KgStateParams& P, const double* U, const GramView& gx, int R, int p, int dd,
                                                 int row, int col) {
  const int g1 = 1 + P.g, d = P.d;
  const int j = row / g1, a = row - j * g1, b = col - p * g1;
  const double v = -gx[((p * g1 + b) * d + dd) + (long)row * R];  // -(dK*_p)^T K^-1 K*_j
  const PointDiff df{U + p * P.dp, U + j * P.dp};
  const Radial rd = pair_radial(P.cp, df, d);
  const double t1 = grad_cov_entry_g(P.cp, rd, df, b, a, dd, P.derivs, P.derivs);  // + dKss / dXs_p
  if (j == p) {  // both factors depend on Xs_p
    const double vt = -gx[((p * g1 + a) * d + dd) + (long)col * R];
    const double t2 = grad_cov_entry_g(P.cp, rd, df, a, b, dd, P.derivs, P.derivs);
    return (v + vt) + (t1 + t2);
  }
  return v + t1;
}

// One workgroup per (point k, coordinate dd; evaluation).  Dynamic LDS: Sp [m (m + 1) / 2] packed (entry (first j <= second i) at
// i (i + 1) / 2 + j -- after the recursion S(j, i) = d L(i, j)) | lcol [2][m] | srow [m].
struct kg_dchol_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const KgStateParams& P) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int kd = blockIdx.x, e = blockIdx.y, tid = threadIdx.x;
    const int m = P.m, g1 = 1 + P.g, d = P.d, R = P.ng + P.A;
    const int p = kd / d, dd = kd - p * d;
    const int tri = m * (m + 1) / 2;
    const double kMinimumStdDev = 2.220446049250313e-16;  // gpp_math.hpp:291
    double* Sp = sm;
    double* lcol = sm + tri;
    double* srow = lcol + 2 * m;
    const double* U = P.U + (long)e * P.u * P.dp;
    const GramView gx = gram_view(P.gx, P.gx_slices, (long)R * m, e);
    const double* Lg = P.blob + (long)e * P.rec_stride + P.rec_L;
    for (int idx = tid; idx < m * m; idx += 256) {
      const int row = idx % m, col = idx / m;
      if (row > col) continue;
      double v = 0.0;
      if (col / g1 == p)
        v = grad_var_entry(P, U, gx, R, p, dd, row, col);
      else if (row / g1 == p)
        v = grad_var_entry(P, U, gx, R, p, dd, col, row);  // block row p mirrors block column p (gpp_math.cpp:1328-1343)
      Sp[col * (col + 1) / 2 + row] = v;
    }
    if (tid < m) lcol[tid] = Lg[tid];
    __syncthreads();
    for (int kk = 0; kk < m; ++kk) {
      const double* lc = lcol + (kk & 1) * m;
      double* ln = lcol + ((kk + 1) & 1) * m;
      const double Lkk = lc[kk];
      double lnext = 0.0;
      if (kk + 1 < m && tid < m && tid > kk) lnext = Lg[tid + (long)(kk + 1) * m];  // next column of L, in flight during this step
      if (Lkk > kMinimumStdDev) {
        const double skk = 0.5 * Sp[kk * (kk + 1) / 2 + kk] / Lkk;
        if (tid >= kk && tid < m) srow[tid] = (tid == kk) ? skk : (Sp[tid * (tid + 1) / 2 + kk] - lc[tid] * skk) / Lkk;
        __syncthreads();
        if (tid >= kk && tid < m) Sp[tid * (tid + 1) / 2 + kk] = srow[tid];
        const int rem = m - kk - 1;
        for (int t = tid; t < rem * rem; t += 256) {
          const int jj = t % rem, ii = t / rem;
          if (jj <= ii) {
            const int j = kk + 1 + jj, i = kk + 1 + ii;
            const int o = i * (i + 1) / 2 + j;
            Sp[o] = Sp[o] - srow[i] * lc[j] - lc[i] * srow[j];
          }
        }
      }
      if (tid < m) ln[tid] = lnext;
      __syncthreads();
    }
    double* out = P.dL + ((long)e * P.q * d + kd) * tri;
    for (int t = tid; t < tri; t += 256) out[t] = Sp[t];
  }
};
__global__ __launch_bounds__(256) void kg_dchol_kernel(KgStateParams P) {
  kg_dchol_kernel_body::run(MOE_VBLOCK, MOE_VGRID, nullptr, P);
}

// Y = L^-T tril(ZC), one workgroup per evaluation.  Dynamic LDS: Lp | Yp, both packed lower triangles (entry (row l >= column j) at
// l (l + 1) / 2 + j); Yp goes to global memory for kg_finish_kernel.  (general m; small m: kg_finish_small_kernel)
struct kg_y_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const KgFinishParams& P, double* __restrict__ Y) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int e = blockIdx.x, tid = threadIdx.x;
    const int m = P.m;
    const int tri = m * (m + 1) / 2;
    double* Lp = sm;
    double* Yp = sm + tri;
    const double* Lg = P.blob + (long)e * P.rec_stride + P.rec_L;
    const double* ZC = P.out + (long)e * P.out_stride + 1;
    for (int idx = tid; idx < m * m; idx += 128) {
      const int l = idx % m, j = idx / m;
      if (l >= j) {
        Lp[l * (l + 1) / 2 + j] = Lg[idx];
        Yp[l * (l + 1) / 2 + j] = ZC[idx];
      }
    }
    __syncthreads();
    // Y[:, j] = L^-T (column j of tril(ZC)), rows l >= j only (dL[l, j] = 0 above the diagonal): thread j, back substitution in place;
    // the reads of L are uniform over the workgroup
    for (int rr = m - 1; rr >= 0; --rr) {
      if (tid <= rr) {
        double t = Yp[rr * (rr + 1) / 2 + tid];
  #pragma unroll 4
        for (int i = m - 1; i > rr; --i) t = fma(-Lp[i * (i + 1) / 2 + rr], Yp[i * (i + 1) / 2 + tid], t);
        Yp[rr * (rr + 1) / 2 + tid] = t / Lp[rr * (rr + 1) / 2 + rr];
      }
    }
    __syncthreads();
    double* out = Y + (long)e * tri;
    for (int t = tid; t < tri; t += 128) out[t] = Yp[t];
  }
};
__global__ __launch_bounds__(128) void kg_y_kernel(KgFinishParams P, double* __restrict__ Y) {
  kg_y_kernel_body::run(MOE_VBLOCK, MOE_VGRID, nullptr, P, Y);
}

// DIR[gc] of evaluation e: kg_dir_kernel's partial sums over sample ranges, added up in slice order (r5: where DIR is consumed; a
// kernel of its own before).  The loads go out together, one slice per lane; lane 0 adds them in order.  Valid in lane 0; at most 64
// slices (kDirSlices = 32).
__device__ __forceinline__ double dir_sum(const KgFinishParams& P, int e, int gc, int lane) {
  const double* p = P.dir_part + ((long)e * P.ng + gc) * P.dir_slices;
  const double mine = (lane < P.dir_slices) ? p[lane] : 0.0;
  double tot = 0.0;
  for (int sl = 0; sl < P.dir_slices; ++sl) tot += __shfl(mine, sl, 64);
  return tot;
}

// The gradient component of (point k, coordinate dd) from < dL_k,dd , Y >, DIR - GTB and the winner's grad mu; lane 0 writes it.
__device__ __forceinline__ void finish_component(const KgFinishParams& P, int e, int idx, int lane, double zmc, double kg_sum) {
  const int m = P.m, g1 = 1 + P.g, d = P.d, qd = P.q * P.d;
  const double* GTB = P.out + (long)e * P.out_stride + 1 + m * m + P.ng;
  const int k = idx / d, dd = idx - k * d;
  double direct = 0.0;
  for (int b = 0; b < g1; ++b) {
    const int gc = (k * g1 + b) * d + dd;
    const double dir = dir_sum(P, e, gc, lane);
    direct += dir - GTB[gc];
  }
  if (lane != 0) return;
  double val = -(direct - zmc);  // aggregate -= gic . z  (.cpp:214-221)
  // winner term: + M grad mu[winner]  (.cpp:157-161); added once, by the shard that owns sample 0
  if (P.winner[e] == k && P.first_sample == 0) val += (double)P.num_mc * P.gmu[(long)e * qd + idx];
  double* fin = P.fin + (long)e * (1 + qd + 3);
  fin[1 + idx] = val;
  if (idx == 0) {
    fin[0] = kg_sum;
    fin[1 + qd] = (double)P.counters[2 * e];
    fin[2 + qd] = (double)P.counters[2 * e + 1];
    fin[3 + qd] = (double)P.flags[e];
  }
}

// One wavefront per (point k, coordinate dd; evaluation): < dL_k,dd , Y > and the assembly of the gradient component.
struct kg_finish_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const KgFinishParams& P, const double* __restrict__ Y) {
    const int idx = blockIdx.x, e = blockIdx.y, lane = threadIdx.x;
    const int m = P.m, qd = P.q * P.d;
    const int tri = m * (m + 1) / 2;
    const double* dl = P.dL + ((long)e * qd + idx) * tri;
    const double* y = Y + (long)e * tri;
    double acc = 0.0;
  #pragma unroll 4
    for (int t = lane; t < tri; t += 64) acc = fma(y[t], dl[t], acc);
    const double zmc = wave_sum64(acc);
    finish_component(P, e, idx, lane, zmc, P.out[(long)e * P.out_stride]);
  }
};
__global__ __launch_bounds__(64) void kg_finish_kernel(KgFinishParams P, const double* __restrict__ Y) {
  kg_finish_kernel_body::run(MOE_VBLOCK, MOE_VGRID, nullptr, P, Y);
}

// Small m (<= 8) and few chunk partials -- the latency path of a q-KG call (r5): ONE launch, a 256-thread workgroup per (component,
// evaluation), each forming what it needs itself instead of waiting for three more kernels: ZC from kg_zc_part_kernel's chunk partials
// (kg_zc_sum_kernel's own scheme: gs lanes per entry striding the chunks + a fixed butterfly), Y = L^-T tril(ZC) in LDS (kg_y_kernel's
// substitution), its own < dL, Y >, DIR - GTB term; workgroup 0 of an evaluation also adds up kg_sum as kg_zc_sum_kernel's block 0 does.
// The same operations in the same order as the separate kernels: the same bits.
struct kg_finish_small_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const KgFinishParams& P) {
    __shared__ double zcs[64], Lp[36], Yp[36], red[4];
    const int idx = blockIdx.x, e = blockIdx.y, tid = threadIdx.x;
    const int m = P.m, qd = P.q * P.d;
    const int tri = m * (m + 1) / 2;
    {
      const int gs = P.zc_gs, per_block = 256 / gs;
      for (int base = 0; base < m * m; base += per_block) {
        const int oo = base + tid / gs, g = tid % gs;
        const bool ok = oo < m * m;
        const double* p = P.zc_part + (long)e * P.zc_chunks * m * m + (ok ? oo : 0);
        double v = 0.0;
  #pragma unroll 8
        for (int ch = g; ch < P.zc_chunks; ch += gs) v += p[(long)ch * m * m];
        for (int off = gs >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (ok && g == 0) zcs[oo] = v;
      }
    }
    double kg_sum = 0.0;
    if (idx == 0) {  // (uniform over the workgroup)
      const double bp = P.blob[(long)e * P.rec_stride + P.rec_bp];
      double acc = 0.0;
  #pragma unroll 8
      for (int i = tid; i < P.num_local; i += 256) acc += bp + P.best_value[(long)e * P.num_local + i];
      const double w = wave_sum64(acc);
      __syncthreads();
      if ((tid & 63) == 0) red[tid >> 6] = w;
      __syncthreads();
      kg_sum = (red[0] + red[1]) + (red[2] + red[3]);
    }
    __syncthreads();
    const double* Lg = P.blob + (long)e * P.rec_stride + P.rec_L;
    if (tid < m * m) {
      const int l = tid % m, j = tid / m;
      if (l >= j) {
        Lp[l * (l + 1) / 2 + j] = Lg[tid];
        Yp[l * (l + 1) / 2 + j] = zcs[tid];
      }
    }
    __syncthreads();
    for (int rr = m - 1; rr >= 0; --rr) {
      if (tid <= rr) {
        double t = Yp[rr * (rr + 1) / 2 + tid];
  #pragma unroll 4
        for (int i = m - 1; i > rr; --i) t = fma(-Lp[i * (i + 1) / 2 + rr], Yp[i * (i + 1) / 2 + tid], t);
        Yp[rr * (rr + 1) / 2 + tid] = t / Lp[rr * (rr + 1) / 2 + rr];
      }
    }
    __syncthreads();
    if (tid >= 64) return;
    const int lane = tid;
    const double* dl = P.dL + ((long)e * qd + idx) * tri;
    double acc = 0.0;
  #pragma unroll 4
    for (int t = lane; t < tri; t += 64) acc = fma(Yp[t], dl[t], acc);
    const double zmc = wave_sum64(acc);
    finish_component(P, e, idx, lane, zmc, kg_sum);
  }
};
__global__ __launch_bounds__(256) void kg_finish_small_kernel(KgFinishParams P) {
  kg_finish_small_kernel_body::run(MOE_VBLOCK, MOE_VGRID, nullptr, P);
}

template <class K>
void opt_in_lds(K kern, size_t shm) {
  // (set on every launch: the attribute is per device where the runtime enforces it, and one process may drive several devices)
  if (shm > 48 * 1024)
    MOE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
}

}  // namespace

void launch_kg_state(const KgStateParams& P, hipStream_t s) {
  const size_t mm = (size_t)P.m * P.m;
  const size_t room = kLdsBudget / sizeof(double) - mm;
  const int rchunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(P.A, 1), room / P.m));
  const size_t shm = sizeof(double) * (mm + (size_t)rchunk * P.m);
  opt_in_lds(kg_state_kernel, shm);
  launch_kernel_ens<kg_state_kernel_body, 256>(kg_state_kernel, dim3(P.E), dim3(256), shm, s, P, rchunk);
  MOE_HIP_CHECK(hipGetLastError());
}

void launch_kg_dchol(const KgStateParams& P, hipStream_t s) {
  if (P.ng <= 0) return;
  const size_t shm = sizeof(double) * ((size_t)P.m * (P.m + 1) / 2 + 3 * (size_t)P.m);
  opt_in_lds(kg_dchol_kernel, shm);
  launch_kernel_ens<kg_dchol_kernel_body, 256>(kg_dchol_kernel, dim3(P.q * P.d, P.E), dim3(256), shm, s, P);
  MOE_HIP_CHECK(hipGetLastError());
}

void launch_kg_finish(const KgFinishParams& P, double* Y, hipStream_t s) {
  if (P.zc_part != nullptr) {  // (m <= 8: the caller did not launch kg_zc_sum_kernel)
    launch_kernel_ens<kg_finish_small_kernel_body, 256>(kg_finish_small_kernel, dim3(P.q * P.d, P.E), dim3(256), 0, s, P);
  } else {
    const size_t shm = sizeof(double) * (size_t)P.m * (P.m + 1);
    opt_in_lds(kg_y_kernel, shm);
    launch_kernel_ens<kg_y_kernel_body, 128>(kg_y_kernel, dim3(P.E), dim3(128), shm, s, P, Y);
    launch_kernel_ens<kg_finish_kernel_body, 64>(kg_finish_kernel, dim3(P.q * P.d, P.E), dim3(64), 0, s, P, (const double*)Y);
  }
  MOE_HIP_CHECK(hipGetLastError());
}

}  // namespace moe
