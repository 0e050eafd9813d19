// tools/ubench.hip -- gfx950 FP64 VALU micro-benchmarks that size the q-KG Monte-Carlo kernel's inner loop:
// issue cost (cycles per wave-instruction per SIMD) and dependent-issue latency of the FP64 instructions the covariance
// loop is made of.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/bin/ubench ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);  \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

constexpr int ITER = 100000;

#define OP_FMA(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b))
#define OP_ADD(x) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(a))
#define OP_MUL(x) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(a))
#define OP_MAX(x) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x) : "v"(a))
#define OP_RSQ(x) asm volatile("v_rsq_f64 %0, %0" : "+v"(x))
#define OP_RNDNE(x) asm volatile("v_rndne_f64 %0, %0" : "+v"(x))
#define OP_LDEXP(x) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(x) : "v"(ia))
#define OP_MOV(x) asm volatile("v_mov_b64 %0, %1" : "=v"(x) : "v"(a))
#define OP_FMA32(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(xf) : "v"(af), "v"(bf))

#define KERNEL(NAME, OP)                                                                             \
  template <int K>                                                                                   \
  __global__ void NAME(double* out, long long* cyc, double a, double b, int ia) {                    \
    double x0 = a + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5,    \
           x6 = x0 + 6, x7 = x0 + 7;                                                                 \
    float xf = (float)a, af = (float)a, bf = (float)b;                                               \
    (void)xf; (void)af; (void)bf;                                                                    \
    const long long t0 = clock64();                                                                  \
    for (int i = 0; i < ITER; ++i) {                                                                 \
      OP(x0);                                                                                        \
      if (K > 1) OP(x1);                                                                             \
      if (K > 2) { OP(x2); OP(x3); }                                                                 \
      if (K > 4) { OP(x4); OP(x5); OP(x6); OP(x7); }                                                 \
    }                                                                                                \
    const long long t1 = clock64();                                                                  \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                 \
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + xf;         \
  }

KERNEL(k_fma, OP_FMA)
KERNEL(k_add, OP_ADD)
KERNEL(k_mul, OP_MUL)
KERNEL(k_max, OP_MAX)
KERNEL(k_rsq, OP_RSQ)
KERNEL(k_rndne, OP_RNDNE)
KERNEL(k_ldexp, OP_LDEXP)
KERNEL(k_mov, OP_MOV)
KERNEL(k_fma32, OP_FMA32)

template <typename F>
int run(const char* name, F kern, int K, int threads, double* out, long long* cyc) {
  const int blocks = 256;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0000001, 1e-9, 0);
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0000001, 1e-9, 0);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> h(blocks);
  CHECK(hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost));
  double avg = 0;
  for (long long v : h) avg += (double)v;
  avg /= blocks;
  const int waves_per_simd = threads / 256;
  const double instr_per_wave = (double)ITER * K;
  // cycles (s_memtime ticks) the SIMD spends per wave-instruction when `waves_per_simd` waves share it
  // wall-clock view: ns per wave-instruction per SIMD (x 2.4 = cycles at the 2.4 GHz nominal clock)
  const double ns_simd = ms * 1e6 / instr_per_wave / waves_per_simd;
  std::printf("%-8s chains=%d waves/SIMD=%d : %6.2f memtime-ticks/instr/wave | wall %7.3f ms -> %5.2f ns/instr/SIMD = %5.2f cyc@2.4GHz\n",
              name, K, waves_per_simd, avg / instr_per_wave, ms, ns_simd, ns_simd * 2.4);
  return 0;
}

int main() {
  double* out;
  long long* cyc;
  CHECK(hipMalloc(&out, sizeof(double) * 256 * 1024));
  CHECK(hipMalloc(&cyc, sizeof(long long) * 256));
  for (int threads : {256, 512, 1024}) {
#define RUNALL(NAME, KERN)                             \
  run(NAME, KERN<1>, 1, threads, out, cyc);            \
  run(NAME, KERN<2>, 2, threads, out, cyc);            \
  run(NAME, KERN<4>, 4, threads, out, cyc);            \
  run(NAME, KERN<8>, 8, threads, out, cyc);
    RUNALL("fma64", k_fma)
    RUNALL("add64", k_add)
    RUNALL("rsq64", k_rsq)
    RUNALL("rndne64", k_rndne)
    RUNALL("ldexp64", k_ldexp)
    RUNALL("mov64", k_mov)
  }
  int clk = 0;
  CHECK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0));
  int wclk = 0;
  (void)hipDeviceGetAttribute(&wclk, hipDeviceAttributeWallClockRate, 0);
  std::printf("clock rate attr: %d kHz, wall clock rate: %d kHz (clock64 ticks are s_memtime ticks)\n", clk, wclk);
  return 0;
}
