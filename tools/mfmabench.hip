// tools/mfmabench.hip -- what does v_mfma_f64_16x16x4_f64 sustain on gfx950?  W wavefronts per SIMD, each issuing `iters`
// rounds of 8 independent MFMAs (register operands only).  Build: hipcc --offload-arch=gfx950 -O3 -mllvm
// -amdgpu-mfma-vgpr-form tools/mfmabench.hip -o tools/bin/mfmabench
#include <hip/hip_runtime.h>

#include <cstdio>

using f64x4 = __attribute__((ext_vector_type(4))) double;

__global__ void mfma_kernel(int iters, double* out) {
  f64x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f64x4{0.0, 0.0, 0.0, 0.0};
  double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  double* out;
  if (hipMalloc(&out, sizeof(double) * 256 * 1024 * 4) != hipSuccess) return 1;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int waves_per_simd : {1, 2, 4}) {
    const int threads = 256 * waves_per_simd;  // one workgroup per CU
    const int iters = 20000;
    hipLaunchKernelGGL(mfma_kernel, dim3(256), dim3(threads), 0, 0, 100, out);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_kernel, dim3(256), dim3(threads), 0, 0, iters, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * (threads / 64) * (double)iters * 8 * 2048.0;
    std::printf("%d wavefront(s) per SIMD: %.3f ms, %.1f TFLOP/s FP64 MFMA  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n",
                waves_per_simd, ms, flops / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)iters * 8 * waves_per_simd));
  }
  return 0;
}
