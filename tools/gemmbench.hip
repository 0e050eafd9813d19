// tools/gemmbench.hip -- the 128 x 128 FP64 MFMA GEMM core (csrc/gemm128.hpp) on its own: correctness against a host product at an
// awkward small size (every operand layout / mask the library uses), then TFLOP/s at the sizes of the N = 8000 build.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -Icornell_moe_amd/csrc tools/gemmbench.hip -o tools/bin/gemmbench
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gemm128.hpp"

using namespace moe::g128;

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__);    \
      std::exit(1);                                                            \
    }                                                                          \
  } while (0)

template <bool AKC, int AMASK, bool BKC, int BMASK, bool NEG>
float run(GemmArgs g, int batch, int reps) {
  auto kern = gemm128_kernel<AKC, AMASK, BKC, BMASK, NEG>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSmemBytes));
  const int R = (g.M + TM - 1) / TM, Ct = (g.N + TM - 1) / TM;
  g.batch = batch;
  const int grid = R * Ct * batch;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), kSmemBytes, 0, g);
  CK(hipGetLastError());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), kSmemBytes, 0, g);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

static double urand() { return (double)std::rand() / RAND_MAX - 0.5; }

template <bool AKC, int AMASK, bool BKC, int BMASK, bool NEG>
void check(const char* name, int M, int N, int K, int scale) {
  // A(i, k): KC -> a[k + i lda] else a[i + k lda]; B(k, j): KC -> b[k + j ldb] else b[j + k ldb]
  const long lda = (AKC ? (long)K * scale : M) + 3, ldb = (BKC ? K : N) + 5, ldc = M + 1;
  const long rowsA = AKC ? M : K;  // number of "columns" (ld strides)
  std::vector<double> a((size_t)lda * (rowsA + 1) * (AKC ? 1 : 1)), b((size_t)ldb * ((BKC ? N : K) + 1)), c((size_t)ldc * N, 0.0);
  for (auto& v : a) v = urand();
  for (auto& v : b) v = urand();
  auto Aref = [&](int i, int k) -> double& { return AKC ? a[(size_t)k + (size_t)i * lda] : a[(size_t)i + (size_t)k * lda]; };
  auto Bref = [&](int k, int j) -> double& { return BKC ? b[(size_t)k + (size_t)j * ldb] : b[(size_t)j + (size_t)k * ldb]; };
  // (the triangular operands hold their zeros: the kernel's contract)
  for (int i = 0; i < M; ++i)
    for (int k = 0; k < K; ++k)
      if ((AMASK == 1 && k > i) || (AMASK == 2 && k < i * scale)) Aref(i, k) = 0.0;
  for (int j = 0; j < N; ++j)
    for (int k = 0; k < K; ++k)
      if (BMASK == 2 && k < j) Bref(k, j) = 0.0;
  auto Ael = [&](int i, int k) -> double { return Aref(i, k); };
  auto Bel = [&](int k, int j) -> double { return Bref(k, j); };
  double *da, *db, *dc;
  CK(hipMalloc(&da, a.size() * 8));
  CK(hipMalloc(&db, b.size() * 8));
  CK(hipMalloc(&dc, c.size() * 8));
  CK(hipMemcpy(da, a.data(), a.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, b.data(), b.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemset(dc, 0, c.size() * 8));
  GemmArgs g{};
  g.A = Operand{da, lda, M, K, scale};
  g.B = Operand{db, ldb, N, K, 1};
  g.C = dc;
  g.ldc = ldc;
  g.M = M;
  g.N = N;
  g.K = K;
  run<AKC, AMASK, BKC, BMASK, NEG>(g, 1, 1);
  CK(hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost));
  double worst = 0.0;
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < N; ++j) {
      double s = 0.0;
      for (int k = 0; k < K; ++k) s += Ael(i, k) * Bel(k, j);
      if (NEG) s = -s;
      worst = std::fmax(worst, std::fabs(s - c[(size_t)i + (size_t)j * ldc]));
    }
  std::printf("check %-34s M=%d N=%d K=%d: max |diff| %.3e %s\n", name, M, N, K, worst, worst < 1e-11 ? "ok" : "FAIL");
  CK(hipFree(da));
  CK(hipFree(db));
  CK(hipFree(dc));
}

int main(int argc, char** argv) {
  check<false, 0, true, 0, false>("A rows, B k-contig (plain NN)", 300, 200, 150, 1);
  check<false, 0, true, 2, false>("A rows, B lower (trtri 1st)", 260, 260, 260, 1);
  check<false, 1, true, 0, true>("A lower rows, B k-contig, neg", 333, 190, 333, 1);
  check<true, 2, true, 0, false>("A^T lower k-contig (x3), B k-contig", 100, 90, 300, 3);
  check<false, 0, false, 0, false>("A rows, B rows (syrk operands)", 257, 129, 64, 1);
  // timing: square NN at 4096, the top level of the N = 8000 inverse (3904 x 4096 x 4096 triangular)
  const int n = (argc > 1) ? std::atoi(argv[1]) : 4096;
  double *da, *db, *dc;
  CK(hipMalloc(&da, (size_t)n * n * 8));
  CK(hipMalloc(&db, (size_t)n * n * 8));
  CK(hipMalloc(&dc, (size_t)n * n * 8));
  {
    std::vector<double> h((size_t)n * n);
    for (auto& v : h) v = urand();
    CK(hipMemcpy(da, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  }
  GemmArgs g{};
  g.A = Operand{da, n, n, n, 1};
  g.B = Operand{db, n, n, n, 1};
  g.C = dc;
  g.ldc = n;
  g.M = g.N = g.K = n;
  float ms = run<false, 0, true, 0, false>(g, 1, 5);
  std::printf("NN %d^3: %.3f ms, %.1f TFLOP/s\n", n, ms, 2.0 * n * (double)n * n / ms / 1e9);
  ms = run<false, 0, true, 2, false>(g, 1, 5);
  std::printf("A x lower-tri B %d^3 (half the flops): %.3f ms, %.1f TFLOP/s\n", n, ms, 1.0 * n * (double)n * n / ms / 1e9);
  ms = run<false, 1, true, 0, true>(g, 1, 5);
  std::printf("lower-tri A x B %d^3 (half the flops): %.3f ms, %.1f TFLOP/s\n", n, ms, 1.0 * n * (double)n * n / ms / 1e9);
  ms = run<true, 2, true, 0, false>(g, 1, 5);
  std::printf("lower-tri A^T x B %d^3 (half the flops): %.3f ms, %.1f TFLOP/s\n", n, ms, 1.0 * n * (double)n * n / ms / 1e9);
  // the rank-512 updates of the N = 8000 factorisation, one by one
  if (n >= 8000) {
    int* info;
    CK(hipMalloc(&info, 4));
    CK(hipMemset(info, 0, 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(syrk128_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSmemBytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int N = 8000;
    double total = 0.0;
    for (int ko = 0; ko + 512 < N; ko += 512) {
      const int trailing = N - ko - 512, T = (trailing + 127) / 128;
      hipLaunchKernelGGL(syrk128_kernel, dim3(T * (T + 1) / 2), dim3(256), kSmemBytes, 0, da, (long)n, N, ko + 512, ko, 512, (const int*)info);
      CK(hipEventRecord(e0));
      for (int r = 0; r < 3; ++r)
        hipLaunchKernelGGL(syrk128_kernel, dim3(T * (T + 1) / 2), dim3(256), kSmemBytes, 0, da, (long)n, N, ko + 512, ko, 512, (const int*)info);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= 3;
      total += ms;
      std::printf("syrk trailing %5d (%4d tiles): %.3f ms, %.1f TFLOP/s (lower-triangle flops)\n", trailing, T * (T + 1) / 2, ms,
                  (double)trailing * trailing * 512.0 / ms / 1e9);
    }
    std::printf("syrk total %.3f ms\n", total);
  }
  return 0;
}
