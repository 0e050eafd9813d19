// tools/covbench.hip -- what bounds the N x M covariance build on gfx950?  Variants of the output-write-bound kernel
// (one thread per output row, B points broadcast from LDS) timed at the C3 tail shape (N = 1000, M = 80000, d = 8):
//   0  stores only (r2 of one dimension): the store-path ceiling for this access pattern
//   1  the shipped arithmetic (unscaled diff, inv_l2 multiply, sqrt_nonneg + exp_nonpos)
//   2  pre-scaled coordinates + sqrt_pos + table exp (the MC kernel's arithmetic)
// with COLS B points per workgroup and optional non-temporal stores.
// Build: hipcc --offload-arch=gfx950 -O3 -I cornell_moe_amd/csrc tools/covbench.hip -o tools/bin/covbench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "fastmath.hpp"

using namespace moe;

#define CHECK(x)                                                                 \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
      return 1;                                                                  \
    }                                                                            \
  } while (0)

struct P8 {
  double inv_l2[8], inv_l[8], alpha;
};

template <int VAR, int COLS, int ROWS, bool NT>
__global__ __launch_bounds__(ROWS) void cov_kernel(P8 cp, const double* __restrict__ A, int nA, const double* __restrict__ B,
                                                   int nB, double* __restrict__ out, long ld) {
  constexpr int DP = 8;
  __shared__ double Bs[COLS][DP];
  __shared__ double etab[64];
  const int j0 = blockIdx.x * COLS;
  const int nj = min(COLS, nB - j0);
  for (int t = threadIdx.x; t < nj * DP; t += blockDim.x)
    Bs[t / DP][t % DP] = B[(long)(j0 + t / DP) * DP + (t % DP)] * (VAR == 2 ? cp.inv_l[t % DP] : 1.0);
  if (threadIdx.x < 64) etab[threadIdx.x] = kExp2Tab64[threadIdx.x];
  __syncthreads();
  const int r = blockIdx.y * ROWS + threadIdx.x;
  if (r >= nA) return;
  double xi[DP];
#pragma unroll
  for (int k = 0; k < DP; ++k) xi[k] = A[(long)r * DP + k] * (VAR == 2 ? cp.inv_l[k] : 1.0);
  double* o = out + r + (long)j0 * ld;
#pragma unroll 4
  for (int jj = 0; jj < nj; ++jj) {
    double v;
    if (VAR == 0) {
      v = xi[0] - Bs[jj][0];
    } else if (VAR == 1) {
      double r2 = 0.0;
#pragma unroll
      for (int k = 0; k < DP; ++k) {
        const double d = xi[k] - Bs[jj][k];
        r2 = fma(d * d, cp.inv_l2[k], r2);
      }
      const double s = sqrt_nonneg(r2);
      const double a = 2.236067977499789696409173668731276235 * s;
      const double e = exp_nonpos(-a);
      v = cp.alpha * e * (1.0 + a + (5.0 / 3.0) * r2);
    } else {
      double r2 = 1.0e-300;
#pragma unroll
      for (int k = 0; k < DP; ++k) {
        const double d = xi[k] - Bs[jj][k];
        r2 = fma(d, d, r2);
      }
      const double a = 2.236067977499789696409173668731276235 * sqrt_pos(r2);
      const double e = exp_nonpos_tab(-a, etab);
      v = (cp.alpha * e) * fma(a, fma(a, 1.0 / 3.0, 1.0), 1.0);
    }
    if (NT)
      __builtin_nontemporal_store(v, o + (long)jj * ld);
    else
      o[(long)jj * ld] = v;
  }
}

template <int VAR, int COLS, int ROWS, bool NT>
int run(const P8& cp, const double* A, int nA, const double* B, int nB, double* out) {
  dim3 grid((nB + COLS - 1) / COLS, (nA + ROWS - 1) / ROWS);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((cov_kernel<VAR, COLS, ROWS, NT>), grid, dim3(ROWS), 0, 0, cp, A, nA, B, nB, out, (long)nA);
  CHECK(hipEventRecord(e0));
  const int reps = 10;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((cov_kernel<VAR, COLS, ROWS, NT>), grid, dim3(ROWS), 0, 0, cp, A, nA, B, nB, out, (long)nA);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double bytes = 8.0 * ((double)nA * nB + 8.0 * nA + 8.0 * nB);
  std::printf("var %d cols %3d rows %3d nt %d : %.4f ms  %.0f GB/s  (%.3f of 8 TB/s)\n", VAR, COLS, ROWS, (int)NT, ms,
              bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0);
  return 0;
}

int main(int argc, char** argv) {
  const int nA = argc > 1 ? atoi(argv[1]) : 1000, nB = argc > 2 ? atoi(argv[2]) : 80000;
  std::vector<double> hA((size_t)nA * 8), hB((size_t)nB * 8);
  unsigned s = 12345;
  auto rnd = [&] { s = s * 1664525u + 1013904223u; return (s >> 8) / 16777216.0; };
  for (auto& v : hA) v = rnd();
  for (auto& v : hB) v = rnd();
  double *A, *B, *out;
  CHECK(hipMalloc(&A, hA.size() * 8));
  CHECK(hipMalloc(&B, hB.size() * 8));
  CHECK(hipMalloc(&out, (size_t)nA * nB * 8));
  CHECK(hipMemcpy(A, hA.data(), hA.size() * 8, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(B, hB.data(), hB.size() * 8, hipMemcpyHostToDevice));
  P8 cp;
  for (int k = 0; k < 8; ++k) {
    cp.inv_l[k] = 1.0 / (0.3 + 0.05 * k);
    cp.inv_l2[k] = cp.inv_l[k] * cp.inv_l[k];
  }
  cp.alpha = 1.3;
  std::printf("N = %d, M = %d\n", nA, nB);
  run<0, 16, 256, false>(cp, A, nA, B, nB, out);
  run<0, 16, 256, true>(cp, A, nA, B, nB, out);
  run<0, 64, 256, false>(cp, A, nA, B, nB, out);
  run<0, 64, 128, false>(cp, A, nA, B, nB, out);
  run<0, 64, 64, false>(cp, A, nA, B, nB, out);
  run<1, 16, 256, false>(cp, A, nA, B, nB, out);
  run<1, 64, 256, false>(cp, A, nA, B, nB, out);
  run<2, 16, 256, false>(cp, A, nA, B, nB, out);
  run<2, 16, 256, true>(cp, A, nA, B, nB, out);
  run<2, 32, 256, false>(cp, A, nA, B, nB, out);
  run<2, 64, 256, false>(cp, A, nA, B, nB, out);
  run<2, 64, 256, true>(cp, A, nA, B, nB, out);
  run<2, 64, 128, false>(cp, A, nA, B, nB, out);
  run<2, 64, 64, false>(cp, A, nA, B, nB, out);
  run<2, 128, 64, false>(cp, A, nA, B, nB, out);
  return 0;
}
