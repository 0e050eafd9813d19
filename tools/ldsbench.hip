// tools/ldsbench.hip -- LDS read cost on gfx950: ds_read_b64 pairs vs one ds_read2st64_b64 (what the compiler merges the MC
// kernel's row-strided tile loads into).  512 threads per workgroup (2 wavefronts per SIMD, like kg_mc_kernel at C3), one
// workgroup per CU, each wavefront issues `iters` x 8 double loads from a 64-double-strided table and nothing else.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ldsbench.hip -o tools/bin/ldsbench
#include <hip/hip_runtime.h>

#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(512) void lds_kernel(int iters, double* out) {
  __shared__ double tab[16 * 64 * 9];
  for (int t = threadIdx.x; t < 16 * 64 * 9; t += 512) tab[t] = 1.0 + t;
  __syncthreads();
  const unsigned base = (unsigned)(size_t)(tab) + (threadIdx.x & 63) * 8;
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
  for (int it = 0; it < iters; ++it) {
    unsigned addr = base + (it & 3) * 64 * 36 * 8;
    double v0, v1, v2, v3, v4, v5, v6, v7;
    if (MODE == 0) {
      // 4 x 8 loads in flight before the wait (the first three groups only occupy the pipe: same destination registers)
      asm volatile(
          "ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:512\n ds_read_b64 %2, %8 offset:1024\n ds_read_b64 %3, %8 offset:1536\n"
          "ds_read_b64 %4, %8 offset:2048\n ds_read_b64 %5, %8 offset:2560\n ds_read_b64 %6, %8 offset:3072\n"
          "ds_read_b64 %7, %8 offset:3584\n"
          "ds_read_b64 %0, %8 offset:4096\n ds_read_b64 %1, %8 offset:4608\n ds_read_b64 %2, %8 offset:5120\n ds_read_b64 %3, %8 offset:5632\n"
          "ds_read_b64 %4, %8 offset:6144\n ds_read_b64 %5, %8 offset:6656\n ds_read_b64 %6, %8 offset:7168\n"
          "ds_read_b64 %7, %8 offset:7680\n"
          "ds_read_b64 %0, %8 offset:8192\n ds_read_b64 %1, %8 offset:8704\n ds_read_b64 %2, %8 offset:9216\n ds_read_b64 %3, %8 offset:9728\n"
          "ds_read_b64 %4, %8 offset:10240\n ds_read_b64 %5, %8 offset:10752\n ds_read_b64 %6, %8 offset:11264\n"
          "ds_read_b64 %7, %8 offset:11776\n"
          "ds_read_b64 %0, %8 offset:12288\n ds_read_b64 %1, %8 offset:12800\n ds_read_b64 %2, %8 offset:13312\n ds_read_b64 %3, %8 offset:13824\n"
          "ds_read_b64 %4, %8 offset:14336\n ds_read_b64 %5, %8 offset:14848\n ds_read_b64 %6, %8 offset:15360\n"
          "ds_read_b64 %7, %8 offset:15872\n s_waitcnt lgkmcnt(0)\n"
          : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
          : "v"(addr));
    } else {
      typedef double d2 __attribute__((ext_vector_type(2)));
      d2 p0, p1, p2, p3;
      asm volatile(
          "ds_read2st64_b64 %0, %4 offset1:1\n ds_read2st64_b64 %1, %4 offset0:2 offset1:3\n"
          "ds_read2st64_b64 %2, %4 offset0:4 offset1:5\n ds_read2st64_b64 %3, %4 offset0:6 offset1:7\n"
          "ds_read2st64_b64 %0, %4 offset0:8 offset1:9\n ds_read2st64_b64 %1, %4 offset0:10 offset1:11\n"
          "ds_read2st64_b64 %2, %4 offset0:12 offset1:13\n ds_read2st64_b64 %3, %4 offset0:14 offset1:15\n"
          "ds_read2st64_b64 %0, %4 offset0:16 offset1:17\n ds_read2st64_b64 %1, %4 offset0:18 offset1:19\n"
          "ds_read2st64_b64 %2, %4 offset0:20 offset1:21\n ds_read2st64_b64 %3, %4 offset0:22 offset1:23\n"
          "ds_read2st64_b64 %0, %4 offset0:24 offset1:25\n ds_read2st64_b64 %1, %4 offset0:26 offset1:27\n"
          "ds_read2st64_b64 %2, %4 offset0:28 offset1:29\n ds_read2st64_b64 %3, %4 offset0:30 offset1:31\n s_waitcnt lgkmcnt(0)\n"
          : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3)
          : "v"(addr));
      v0 = p0.x; v1 = p0.y; v2 = p1.x; v3 = p1.y; v4 = p2.x; v5 = p2.y; v6 = p3.x; v7 = p3.y;
    }
    a0 += v0; a1 += v1; a2 += v2; a3 += v3; a4 += v4; a5 += v5; a6 += v6; a7 += v7;
  }
  out[blockIdx.x * 512 + threadIdx.x] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
}

template <int MODE>
void run(int iters, double* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(lds_kernel<MODE>, dim3(256), dim3(512), 0, 0, iters, out);
  hipEventRecord(e0);
  hipLaunchKernelGGL(lds_kernel<MODE>, dim3(256), dim3(512), 0, 0, iters, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // per CU: 8 wavefronts x iters x 8 loads x 512 B
  const double bytes = 8.0 * iters * 32 * 512;
  std::printf("mode %d (%s): %.3f ms  -> %.1f B/clk/CU at 2.4 GHz\n", MODE, MODE ? "ds_read2st64_b64" : "ds_read_b64", ms,
              bytes / (ms * 1e-3 * 2.4e9));
}

int main() {
  double* out;
  hipMalloc(&out, 256 * 512 * 8);
  run<0>(5000, out);
  run<1>(5000, out);
  run<0>(5000, out);
  run<1>(5000, out);
  return 0;
}
