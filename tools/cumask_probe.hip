// tools/cumask_probe.hip -- which CUs a hipExtStreamCreateWithCUMask stream runs on (r6: the look-ahead Cholesky keeps part of every XCD
// free for its 64-column steps).  For a few masks: launch 4096 single-wave workgroups that spin ~20 us, record (XCC_ID, SE, CU) of each,
// print how many distinct CUs per XCD were used.   hipcc --offload-arch=gfx950 -O2 tools/cumask_probe.hip -o tools/bin/cumask_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <map>
#include <set>
#include <vector>

__global__ void where_kernel(unsigned* out, int spin) {
  unsigned xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) {
  }
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = xcc;
    out[2 * blockIdx.x + 1] = hw;
  }
}

static void run(const char* name, const std::vector<uint32_t>& mask) {
  hipStream_t s;
  hipError_t e = mask.empty() ? hipStreamCreate(&s) : hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
  if (e != hipSuccess) {
    printf("%s: stream creation failed: %s\n", name, hipGetErrorString(e));
    return;
  }
  const int nb = 4096;
  unsigned* d;
  hipMalloc(&d, sizeof(unsigned) * 2 * nb);
  hipLaunchKernelGGL(where_kernel, dim3(nb), dim3(64), 0, s, d, 2000);
  hipStreamSynchronize(s);
  std::vector<unsigned> h(2 * nb);
  hipMemcpy(h.data(), d, sizeof(unsigned) * 2 * nb, hipMemcpyDeviceToHost);
  std::map<unsigned, std::set<unsigned>> per_xcc;
  for (int i = 0; i < nb; ++i) {
    const unsigned xcc = h[2 * i] & 0xf, hw = h[2 * i + 1];
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;  // HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
    per_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
  }
  printf("%-28s:", name);
  int total = 0;
  for (auto& kv : per_xcc) {
    printf(" xcc%u:%zu", kv.first, kv.second.size());
    total += (int)kv.second.size();
  }
  printf("  total %d CUs\n", total);
  hipFree(d);
  hipStreamDestroy(s);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("device: %s, %d CUs\n", p.gcnArchName, p.multiProcessorCount);
  run("no mask", {});
  run("bits 0..127", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0});
  run("bits 0..191", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0});
  run("low 24 bits of each word", std::vector<uint32_t>(8, 0x00ffffffu));
  run("even bits", std::vector<uint32_t>(8, 0x55555555u));
  run("3 of every 4 bits", std::vector<uint32_t>(8, 0x77777777u));
  run("word 0 only", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0});
  return 0;
}
