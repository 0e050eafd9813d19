// tools/mathcheck.hip -- accuracy (ulp vs long-double host references) of candidate FP64 exp / sqrt sequences for the
// covariance inner loops.  Build: hipcc --offload-arch=gfx950 -O3 tools/mathcheck.hip -o tools/bin/mathcheck
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "../cornell_moe_amd/csrc/fastmath.hpp"

using namespace moe;

__device__ double sqrt_one_heron(double s) {
  const double y = __builtin_amdgcn_rsq(s);
  double g = s * y;
  double h = 0.5 * y;
  const double e = fma(-h, g, 0.5);
  g = fma(g, e, g);
  h = fma(h, e, h);
  const double d = fma(-g, g, s);
  g = fma(d, h, g);
  return g;
}

// candidates: v_rsq_f64 seed + ONE Heron correction (no coupled Newton step): 5 instructions instead of 8
__device__ double sqrt_seed_heron(double s) {
  const double y = __builtin_amdgcn_rsq(s);
  const double g = s * y;
  const double d = fma(-g, g, s);
  return fma(d, 0.5 * y, g);
}

__global__ void k(const double* x, int n, const double* tab, double* e_poly, double* e_tab, double* s_two, double* s_one,
                  double* e_tab64, double* s_fast) {
  __shared__ double T[32];
  __shared__ double T64[64];
  if (threadIdx.x < 32) T[threadIdx.x] = tab[threadIdx.x];
  if (threadIdx.x < 64) T64[threadIdx.x] = tab[32 + threadIdx.x];
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  e_poly[i] = exp_nonpos(-x[i]);
  e_tab[i] = exp_nonpos_tab(-x[i], T64);
  s_two[i] = sqrt_nonneg(x[i]);
  s_one[i] = sqrt_one_heron(fmax(x[i], 1e-300));
  e_tab64[i] = exp_nonpos_tab(-x[i], T64);
  s_fast[i] = sqrt_seed_heron(fmax(x[i], 1e-300));
}

static double ulp_of(double ref) {
  if (ref == 0) return 4.9e-324;
  int ex;
  std::frexp(ref, &ex);
  return std::ldexp(1.0, ex - 53);
}

int main() {
  std::mt19937_64 rng(7);
  std::vector<double> x;
  for (double v : {0.0, 1e-300, 1e-30, 1e-16, 1e-8, 0.5, 1.0, 2.0, 10.0, 100.0, 700.0, 744.0, 745.2, 760.0, 800.0}) x.push_back(v);
  std::uniform_real_distribution<double> u(0.0, 40.0), lg(-12.0, 2.8);
  for (int i = 0; i < 2000000; ++i) x.push_back(u(rng));
  for (int i = 0; i < 2000000; ++i) x.push_back(std::pow(10.0, lg(rng)));
  // r5: the whole range of squared distances the MC kernels can hand to sqrt_pos_fast -- 1e-300 (the floor of the distance
  // accumulation, coincident points) up to 1e14 (trial points of far frames) -- log-uniform
  const size_t n_narrow = x.size();
  std::uniform_real_distribution<double> lgw(-300.0, 14.0);
  for (int i = 0; i < 2000000; ++i) x.push_back(std::pow(10.0, lgw(rng)));
  const int n = (int)x.size();
  std::vector<double> tab(96);
  for (int j = 0; j < 32; ++j) tab[j] = (double)std::exp2((long double)j / 32.0L);
  for (int j = 0; j < 64; ++j) tab[32 + j] = (double)std::exp2((long double)j / 64.0L);
  double *dx, *dt, *d[6];
  hipMalloc(&dx, 8 * n);
  hipMalloc(&dt, 8 * 96);
  for (auto& p : d) hipMalloc(&p, 8 * n);
  hipMemcpy(dx, x.data(), 8 * n, hipMemcpyHostToDevice);
  hipMemcpy(dt, tab.data(), 8 * 96, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, dx, n, dt, d[0], d[1], d[2], d[3], d[4], d[5]);
  std::vector<double> h[6];
  for (int a = 0; a < 6; ++a) {
    h[a].resize(n);
    hipMemcpy(h[a].data(), d[a], 8 * n, hipMemcpyDeviceToHost);
  }
  const char* names[6] = {"exp poly-11", "exp table-64", "sqrt 2 heron", "sqrt 1 heron", "exp table-64 (again)", "sqrt seed+heron"};
  for (int a = 0; a < 6; ++a) {
    double worst = 0, wx = 0;
    const bool is_exp = a < 2 || a == 4;
    for (int i = 0; i < n; ++i) {
      long double ref = is_exp ? expl(-(long double)x[i]) : sqrtl((long double)x[i]);
      if (is_exp && ref < 1e-300L) continue;
      if (!is_exp && x[i] < 1e-290) continue;
      const double err = (double)fabsl((long double)h[a][i] - ref) / ulp_of((double)ref);
      if (err > worst) {
        worst = err;
        wx = x[i];
      }
    }
    std::printf("%-14s max error %.3f ulp at x = %.17g\n", names[a], worst, wx);
    if (a == 5) {  // sqrt seed + heron (sqrt_pos_fast): the wide range on its own
      double ww = 0, wwx = 0;
      for (size_t i = n_narrow; i < (size_t)n; ++i) {
        const long double ref = sqrtl((long double)x[i]);
        const double err = (double)fabsl((long double)h[a][i] - ref) / ulp_of((double)ref);
        if (err > ww) {
          ww = err;
          wwx = x[i];
        }
      }
      std::printf("%-14s over [1e-300, 1e14] (2e6 log-uniform arguments): max error %.3f ulp at x = %.17g\n", names[a], ww, wwx);
    }
  }
  std::printf("exp(-0) poly %.17g tab %.17g ; exp(-745.2) poly %g tab %g ; exp(-800) poly %g tab %g\n", h[0][0], h[1][0], h[0][12],
              h[1][12], h[0][14], h[1][14]);
  return 0;
}
