// tools/diagbench.hip -- where the 64 x 64 diagonal-block kernel of the two-level Cholesky (chol_diag_lds_kernel) spends its time:
// launches it on a random SPD block, prints the event-timed duration and thread 0's clock stamps per phase.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMOE_DIAG_PROF -I cornell_moe_amd/csrc -I include tools/diagbench.hip -o tools/bin/diagbench
#include "../cornell_moe_amd/csrc/kernels_linalg.hip"

#include <cstdio>
#include <random>
#include <vector>

int main() {
  const int n = 64, lda = 64;
  std::vector<double> A(n * n);
  std::mt19937 rng(1);
  std::normal_distribution<double> nd;
  std::vector<double> M(n * n);
  for (auto& v : M) v = nd(rng);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double s = (i == j) ? 64.0 : 0.0;
      for (int k = 0; k < n; ++k) s += M[i + k * n] * M[j + k * n];
      A[i + j * lda] = s;
    }
  double *dA, *dL;
  int* dInfo;
  hipMalloc(&dA, sizeof(double) * n * n);
  hipMalloc(&dL, sizeof(double) * n * n);
  hipMalloc(&dInfo, sizeof(int));
  hipMemset(dInfo, 0, sizeof(int));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 5; ++rep) {
    hipMemcpy(dA, A.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
    unsigned long long zero[16] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(moe::moe_diag_prof), zero, sizeof(zero));
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(moe::chol_diag_lds_kernel, dim3(1), dim3(256), 0, 0, dA, (long)lda, dL, (long)lda, 0, n, dInfo);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long p[16];
    hipMemcpyFromSymbol(p, HIP_SYMBOL(moe::moe_diag_prof), sizeof(p));
    std::printf("rep %d: %.1f us by events; clock ticks (100 MHz): load %llu  factor %llu  inverse %llu  store %llu | inside the 4 sub-steps: "
                "16x16 factor+inverse %llu  panel %llu  update %llu\n",
                rep, 1e3 * ms, p[1] - p[0], p[2] - p[1], p[3] - p[2], p[4] - p[3], p[9], p[10], p[11]);
  }
  return 0;
}
