// tools/coissue.hip -- do v_mfma_f64_16x16x4_f64 and FP64 VALU instructions issue concurrently on gfx950?
// One wavefront (or two) per SIMD runs `iters` rounds of NM MFMAs interleaved with NV v_fma_f64 (independent chains).
// If the matrix pipe and the vector ALU overlap, time(NM, NV) ~ max(time(NM, 0), time(0, NV)); if they serialise it is the sum.
// Build: hipcc --offload-arch=gfx950 -O3 tools/coissue.hip -o tools/bin/coissue ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>

using f64x4 = __attribute__((ext_vector_type(4))) double;

#define FMA(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b))
#define MFMA(acc) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

// NV VALU FMAs spread over 8 chains between consecutive MFMAs (NM MFMAs per round over 4 accumulators)
template <int NM, int NVPER>
__global__ void mix_kernel(int iters, double* out, double a, double b) {
  f64x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f64x4{0.0, 0.0, 0.0, 0.0};
  double x[8];
  for (int i = 0; i < 8; ++i) x[i] = a + threadIdx.x + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < (NM > 0 ? NM : 1); ++j) {
      if (NM > 0) MFMA(acc[j & 3]);
#pragma unroll
      for (int v = 0; v < NVPER; ++v) FMA(x[v & 7]);
    }
  }
  double s = 0.0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NM, int NVPER>
static void run(const char* name, int waves_per_simd, double* out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int threads = 256 * waves_per_simd;
  const int iters = 20000;
  hipLaunchKernelGGL((mix_kernel<NM, NVPER>), dim3(256), dim3(threads), 0, 0, 100, out, 1.0000001, 1e-9);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((mix_kernel<NM, NVPER>), dim3(256), dim3(threads), 0, 0, iters, out, 1.0000001, 1e-9);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const int nm = NM > 0 ? NM : 0, nv = (NM > 0 ? NM : 1) * NVPER;
  // cycles per round per SIMD at the clock implied by wall time (2.4 GHz nominal)
  std::printf("%-28s waves/SIMD=%d  %8.3f ms  %7.1f ns/round/SIMD  (%d MFMA + %d FMA per round and wave)\n", name, waves_per_simd, ms,
              ms * 1e6 / iters, nm, nv);
}

int main() {
  double* out;
  if (hipMalloc(&out, sizeof(double) * 256 * 1024 * 4) != hipSuccess) return 1;
  for (int w : {1, 2}) {
    run<5, 0>("mfma only (5)", w, out);
    run<0, 110>("valu only (110)", w, out);
    run<5, 22>("mixed 5 mfma + 110 fma", w, out);
    run<5, 16>("mixed 5 mfma + 80 fma", w, out);
    run<9, 14>("mixed 9 mfma + 126 fma", w, out);
    run<0, 126>("valu only (126)", w, out);
    run<9, 0>("mfma only (9)", w, out);
  }
  return 0;
}
