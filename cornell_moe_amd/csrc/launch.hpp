// cornell_moe_amd/csrc/launch.hpp -- kernel launches and asynchronous copies of the library, immediate or RECORDED (r6).
//
// Why: one KG evaluation of a GP is a chain of ~25 small kernels; a KG-MCMC suggestion
// (gpp_knowledge_gradient_mcmc_optimization.cpp:24-180: the acquisition is the average over an ensemble of GPs, one per hyper-parameter
// sample) runs that chain once per ensemble member and optimiser step -- 16 members x 25 kernels of 5-8 us each, dependent within a
// member, and the device retires them at ~230 k kernels / s whichever stream or host thread issues them.  The members
// run the SAME kernels with different operands.  So a member's chain can be recorded instead of launched -- every launch site of
// the library goes through MOE_LAUNCH / launch_kernel_ens / copy_async / memset_async below, which append to the calling thread's
// Recorder when one is active -- and the recordings of all members are then zipped position by position (mcmc.hip: replay_ensemble;
// the MCMC-averaged Monte-Carlo EI goes the same way):
//   * a kernel that has an ENSEMBLE TWIN becomes ONE launch over all members: its body is a device function
//     (Body::run(blockIdx, gridDim, argbase, args...): the built-in block coordinates are shadowed by parameters, so a body is the
//     kernel's text unchanged), the twin reads member z's arguments from a table in device memory and runs the body on that member's
//     share of grid.z.  Same instructions on the same operands in the same workgroup shapes: the same bits as the per-member launch;
//   * small host <-> device copies become one copy kernel over a table of (dst, src, bytes) (pinned host memory is device-visible);
//   * anything else is replayed member after member at its position.
// Recordings that do not line up (a member whose length scales send it to another kernel variant) are replayed per member on the
// members' own streams -- what an immediate launch would have done.
// (included by common.hpp, behind Error and MOE_HIP_CHECK)
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstring>
#include <functional>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

namespace moe {

// block coordinates as a kernel body sees them (its parameters `blockIdx` and `gridDim` shadow the built-ins)
struct VIdx {
  unsigned x, y, z;
};

// a kernel's arguments as one trivially copyable record (first argument at offset 0)
template <class... T>
struct ArgPack;
template <>
struct ArgPack<> {};
template <class H, class... T>
struct ArgPack<H, T...> {
  H head;
  ArgPack<T...> tail;
};
template <size_t I, class H, class... T>
__host__ __device__ inline const auto& arg_get(const ArgPack<H, T...>& p) {
  if constexpr (I == 0)
    return p.head;
  else
    return arg_get<I - 1>(p.tail);
}
template <class... P>
inline void arg_fill(ArgPack<P...>&) {}
template <class H, class... T, class A0, class... A>
inline void arg_fill(ArgPack<H, T...>& p, A0&& a0, A&&... a) {
  p.head = static_cast<H>(std::forward<A0>(a0));
  if constexpr (sizeof...(T) > 0) arg_fill(p.tail, std::forward<A>(a)...);
}

// The ensemble twin of a kernel: member = blockIdx.z / gz runs Body on its own argument record.
template <class Body, int MaxThreads, int MinBlocks, class... P, size_t... I>
__device__ __forceinline__ void ens_run(const ArgPack<P...>* a, const VIdx& b, const VIdx& g, std::index_sequence<I...>) {
  // the record is read as CONSTANT memory (scalar loads, like the kernarg segment of the single-member kernel): bodies take their
  // struct arguments by reference, so a dynamically indexed field is a load from the table, not from a private copy
  const ArgPack<P...>& r = *(const ArgPack<P...>*)(const __attribute__((address_space(4))) ArgPack<P...>*)a;
  Body::run(b, g, (const void*)a, arg_get<I>(r)...);
}
template <class Body, int MaxThreads, int MinBlocks, class... P>
__global__ __launch_bounds__(MaxThreads, MinBlocks) void ens_kernel(const ArgPack<P...>* __restrict__ table, unsigned gz) {
  // (the division runs on the vector unit: the record's address goes back to scalar registers, so that the arguments are read with
  //  scalar loads and a body may hand the address to scalar inline assembly)
  const unsigned member = blockIdx.z / gz;
  const unsigned long long addr = (unsigned long long)(table + member);
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)addr);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(addr >> 32));
  const ArgPack<P...>* rec = (const ArgPack<P...>*)(((unsigned long long)hi << 32) | lo);
  const unsigned member_s = (unsigned)__builtin_amdgcn_readfirstlane((int)member);
  const VIdx b{blockIdx.x, blockIdx.y, blockIdx.z - member_s * gz};
  const VIdx g{gridDim.x, gridDim.y, gz};
  ens_run<Body, MaxThreads, MinBlocks>(rec, b, g, std::index_sequence_for<P...>{});
}

// one copy kernel for the small copies of all members at one position of their recordings
struct CopyEntry {
  void* dst;
  const void* src;
  size_t bytes;  // a multiple of 8
};
template <int kUnused = 0>
__global__ __launch_bounds__(256) void ens_copy_kernel(const CopyEntry* __restrict__ table) {
  const CopyEntry c = table[blockIdx.y];
  const size_t n = c.bytes / 8;
  const unsigned long long* __restrict__ s = (const unsigned long long*)c.src;
  unsigned long long* __restrict__ d = (unsigned long long*)c.dst;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = s[i];
}

struct LaunchOp {
  enum Kind { kKernel, kCopy, kOther } kind = kOther;
  std::function<void(hipStream_t)> run;  // what an immediate launch would have done, on the stream it is replayed on
  // kKernel with an ensemble twin (ens_launch != nullptr): the twin's launcher, the geometry and the argument record
  void (*ens_launch)(const void* table_dev, int members, dim3 grid, dim3 block, size_t shm, hipStream_t s) = nullptr;
  dim3 grid, block;
  size_t shm = 0;
  std::vector<unsigned char> args;
  CopyEntry copy{nullptr, nullptr, 0};  // kCopy
  hipMemcpyKind copy_kind = hipMemcpyDefault;
  bool host_pinned = false;  // kCopy: the host side is pinned memory (device-visible: the copy may run as a kernel)
};

struct Recorder {
  std::vector<LaunchOp> ops;
  std::vector<std::pair<void*, size_t>> retired_dev, retired_host;  // blocks a buffer outgrew while ops that name them were pending
  static Recorder*& current() {
    static thread_local Recorder* r = nullptr;
    return r;
  }
  struct Scope {
    Recorder* prev;
    explicit Scope(Recorder* r) : prev(current()) { current() = r; }
    ~Scope() { current() = prev; }
    Scope(const Scope&) = delete;
    Scope& operator=(const Scope&) = delete;
  };
};

template <class K, class Pack, size_t... I>
inline void launch_from_pack(K kernel, dim3 grid, dim3 block, size_t shm, hipStream_t s, const Pack& pack, std::index_sequence<I...>) {
  kernel<<<grid, block, shm, s>>>(arg_get<I>(pack)...);
}

// a launch site without an ensemble twin, recorded (MOE_LAUNCH below): replayed member after member
template <class F>
inline void record_closure(F&& f) {
  LaunchOp op;
  op.kind = LaunchOp::kOther;
  op.run = std::forward<F>(f);
  Recorder::current()->ops.push_back(std::move(op));
}

template <class Body, int MaxThreads, int MinBlocks, class... P>
inline void ens_launch_impl(const void* table_dev, int members, dim3 grid, dim3 block, size_t shm, hipStream_t s) {
  auto twin = ens_kernel<Body, MaxThreads, MinBlocks, P...>;
  // (set on every launch: the attribute is per device where the runtime enforces it, and one process may drive several devices)
  if (shm > 48 * 1024)
    MOE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(twin), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  twin<<<dim3(grid.x, grid.y, grid.z * (unsigned)members), block, shm, s>>>((const ArgPack<P...>*)table_dev, grid.z);
}

// a kernel with an ensemble twin: `kernel` is the __global__ wrapper of Body (immediate launches and per-member replays use it)
template <class Body, int MaxThreads, int MinBlocks = 1, class... P, class... A>
inline void launch_kernel_ens(void (*kernel)(P...), dim3 grid, dim3 block, size_t shm, hipStream_t s, A&&... a) {
  static_assert(sizeof...(P) == sizeof...(A), "kernel argument count");
  Recorder* r = Recorder::current();
  if (r == nullptr) {
    kernel<<<grid, block, shm, s>>>(static_cast<P>(std::forward<A>(a))...);
    return;
  }
  using Pack = ArgPack<std::decay_t<P>...>;
  static_assert(std::is_trivially_copyable<Pack>::value, "kernel arguments must be trivially copyable");
  LaunchOp op;
  op.kind = LaunchOp::kKernel;
  op.grid = grid;
  op.block = block;
  op.shm = shm;
  op.args.resize(sizeof(Pack));
  Pack pack;
  std::memset(&pack, 0, sizeof(Pack));
  arg_fill(pack, std::forward<A>(a)...);
  std::memcpy(op.args.data(), &pack, sizeof(Pack));
  op.ens_launch = &ens_launch_impl<Body, MaxThreads, MinBlocks, std::decay_t<P>...>;
  op.run = [kernel, grid, block, shm, pack](hipStream_t rs) {
    launch_from_pack(kernel, grid, block, shm, rs, pack, std::index_sequence_for<P...>{});
  };
  r->ops.push_back(std::move(op));
}

inline void copy_async(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s, bool host_pinned = false) {
  if (bytes == 0) return;
  Recorder* r = Recorder::current();
  if (r == nullptr) {
    MOE_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, kind, s));
    return;
  }
  LaunchOp op;
  op.kind = LaunchOp::kCopy;
  op.copy = CopyEntry{dst, src, bytes};
  op.copy_kind = kind;
  op.host_pinned = host_pinned || kind == hipMemcpyDeviceToDevice;
  op.run = [dst, src, bytes, kind](hipStream_t rs) { MOE_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, kind, rs)); };
  r->ops.push_back(std::move(op));
}

inline void memset_async(void* dst, int value, size_t bytes, hipStream_t s) {
  if (bytes == 0) return;
  Recorder* r = Recorder::current();
  if (r == nullptr) {
    MOE_HIP_CHECK(hipMemsetAsync(dst, value, bytes, s));
    return;
  }
  LaunchOp op;
  op.kind = LaunchOp::kOther;
  op.run = [dst, value, bytes](hipStream_t rs) { MOE_HIP_CHECK(hipMemsetAsync(dst, value, bytes, rs)); };
  r->ops.push_back(std::move(op));
}

}  // namespace moe

// hipLaunchKernelGGL, or its recording (the closure copies what the argument expressions name -- only while recording)
#define MOE_LAUNCH(kernel, grid, block, shm, stream, ...)                                                              \
  do {                                                                                                                 \
    if (::moe::Recorder::current() == nullptr) {                                                                       \
      kernel<<<(grid), (block), (shm), (stream)>>>(__VA_ARGS__);                                                       \
    } else {                                                                                                           \
      ::moe::record_closure([=](hipStream_t moe_replay_stream_) { kernel<<<(grid), (block), (shm), moe_replay_stream_>>>(__VA_ARGS__); }); \
    }                                                                                                                  \
  } while (0)

// a launch site outside the evaluation chains (never reached while recording)
#define MOE_LAUNCH_NOW(kernel, grid, block, shm, stream, ...)                                                                     \
  do {                                                                                                                            \
    if (::moe::Recorder::current() != nullptr) throw ::moe::Error(MOE_ERR_RUNTIME, "launch site reached while launches are being recorded"); \
    kernel<<<(grid), (block), (shm), (stream)>>>(__VA_ARGS__);                                                                    \
  } while (0)

// a kernel body's view of the built-in block coordinates, from its __global__ wrapper
#define MOE_VBLOCK (::moe::VIdx{blockIdx.x, blockIdx.y, blockIdx.z})
#define MOE_VGRID (::moe::VIdx{gridDim.x, gridDim.y, gridDim.z})
