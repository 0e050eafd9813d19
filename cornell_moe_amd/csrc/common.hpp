// cornell_moe_amd/csrc/common.hpp -- shared host-side helpers for libmoe_hip.so (gfx950 only; no CPU fallback).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/moe_hip.h"

namespace moe {

// Exception carrying a reference-style error class (gpp_exception.hpp) across the host code up to the C ABI.
struct Error : public std::runtime_error {
  int code;
  double payload[3];
  Error(int code_in, const std::string& msg, double p0 = 0.0, double p1 = 0.0, double p2 = 0.0)
      : std::runtime_error(msg), code(code_in), payload{p0, p1, p2} {}
};

#define MOE_HIP_CHECK(expr)                                                                                   \
  do {                                                                                                        \
    hipError_t e_ = (expr);                                                                                   \
    if (e_ != hipSuccess) {                                                                                   \
      throw ::moe::Error(MOE_ERR_RUNTIME, std::string("HIP error: ") + hipGetErrorString(e_) + " at " __FILE__ \
                                              ":" + std::to_string(__LINE__) + " (" #expr ")");              \
    }                                                                                                         \
  } while (0)

}  // namespace moe
#include "launch.hpp"
namespace moe {

// Device-memory pool (r5).  A GP build at N = 8000 allocates two 520 MB matrices and half a dozen small buffers, and every
// hyper-parameter sample of an ensemble builds a fresh GP: hipMalloc / hipFree of that size cost more than a millisecond of a 14 ms
// build.  Blocks a DevBuf releases are kept per device, keyed by size, and handed to the next request they fit (at most half as large
// again); 288 GB of HBM make holding on to them cheap.  MOE_POOL=0 switches the pool off, MOE_POOL_MAX_GB (default 48) bounds what it
// holds -- a block that would exceed the bound is freed instead, and a failed hipMalloc empties the pool and retries.  Releasing a
// block keeps hipFree's implicit guarantee: the device is idle before the block can be handed to another stream.
class DevicePool {
 public:
  static DevicePool& get() {
    // (never destroyed: objects that outlive static destruction -- a handle closed by an exit handler -- must still find it, and the
    //  blocks it holds at process exit are left to the runtime's own teardown)
    static DevicePool* pool = new DevicePool;
    return *pool;
  }
  // a block of at least `bytes` on the current device; *got = its size
  void* take(size_t bytes, size_t* got) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const size_t want = round_size(bytes);
    if (enabled_) {
      std::lock_guard<std::mutex> lock(mu_);
      auto& blocks = free_[dev];
      auto it = blocks.lower_bound(want);
      if (it != blocks.end() && it->first <= want + want / 2) {
        void* p = it->second;
        *got = it->first;
        held_ -= it->first;
        blocks.erase(it);
        return p;
      }
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {  // give back what the pool holds, then once more
      (void)hipGetLastError();
      trim();
      e = hipMalloc(&p, want);
    }
    if (e != hipSuccess)
      throw Error(MOE_ERR_RUNTIME, std::string("HIP error: ") + hipGetErrorString(e) + " (hipMalloc of " + std::to_string(want) + " bytes)");
    *got = want;
    return p;
  }
  void give(void* p, size_t bytes) {
    if (p == nullptr) return;
    if (enabled_) {
      hipPointerAttribute_t attr;
      int dev = 0;
      if (hipPointerGetAttributes(&attr, p) == hipSuccess) {
        dev = attr.device;
      } else {  // (whose block is this? filed under a guessed device it could be handed to another device's request: not pooled)
        (void)hipGetLastError();
        (void)hipFree(p);
        return;
      }
      int cur = dev;
      (void)hipGetDevice(&cur);
      if (cur != dev) (void)hipSetDevice(dev);  // (a block of another device than the calling thread's current one)
      (void)hipDeviceSynchronize();             // (what hipFree would have waited for)
      if (poison_) (void)hipMemset(p, 0xFF, bytes);  // (debugging: every double a NaN, every int -1)
      if (cur != dev) (void)hipSetDevice(cur);
      std::lock_guard<std::mutex> lock(mu_);
      if (held_ + bytes <= max_held_) {
        free_[dev].emplace(bytes, p);
        held_ += bytes;
        return;
      }
    }
    (void)hipFree(p);
  }
  // Streams and the device's CU count the same way: hipStreamCreate / hipStreamDestroy / hipGetDeviceProperties together cost a GP
  // constructor more than a millisecond (N = 8000 build: 13.8 ms in the constructor, 11.9 of them in the build itself).
  hipStream_t take_stream(int dev) {
    {
      std::lock_guard<std::mutex> lock(mu_);
      auto& v = streams_[dev];
      if (!v.empty()) {
        hipStream_t s = v.back();
        v.pop_back();
        return s;
      }
    }
    hipStream_t s = nullptr;
    hipError_t e = hipStreamCreate(&s);
    if (e != hipSuccess) throw Error(MOE_ERR_RUNTIME, std::string("HIP error: ") + hipGetErrorString(e) + " (hipStreamCreate)");
    return s;
  }
  void give_stream(int dev, hipStream_t s) {
    if (s == nullptr) return;
    (void)hipStreamSynchronize(s);
    if (enabled_) {
      std::lock_guard<std::mutex> lock(mu_);
      auto& v = streams_[dev];
      if (v.size() < 64) {
        v.push_back(s);
        return;
      }
    }
    (void)hipStreamDestroy(s);
  }
  // pinned host staging buffers (hipHostMalloc: 0.1 ms and more each; a GP holds two and a KG workspace several)
  void* take_host(size_t bytes, size_t* got) {
    const size_t want = round_size(bytes);
    if (enabled_) {
      std::lock_guard<std::mutex> lock(mu_);
      auto it = host_.lower_bound(want);
      if (it != host_.end() && it->first <= want + want / 2) {
        void* p = it->second;
        *got = it->first;
        host_held_ -= it->first;
        host_.erase(it);
        return p;
      }
    }
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e != hipSuccess)
      throw Error(MOE_ERR_RUNTIME, std::string("HIP error: ") + hipGetErrorString(e) + " (hipHostMalloc of " + std::to_string(want) + " bytes)");
    *got = want;
    return p;
  }
  // (the owner has synchronised the stream its copies ran on: every entry point ends with one)
  void give_host(void* p, size_t bytes) {
    if (p == nullptr) return;
    if (enabled_) {
      if (poison_) std::memset(p, 0xFF, bytes);
      std::lock_guard<std::mutex> lock(mu_);
      if (host_held_ + bytes <= kMaxHostHeld) {
        host_.emplace(bytes, p);
        host_held_ += bytes;
        return;
      }
    }
    (void)hipHostFree(p);
  }
  int num_cu(int dev) {
    {
      std::lock_guard<std::mutex> lock(mu_);
      auto it = num_cu_.find(dev);
      if (it != num_cu_.end()) return it->second;
    }
    hipDeviceProp_t prop;
    int cu = 0;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cu = prop.multiProcessorCount;
    std::lock_guard<std::mutex> lock(mu_);
    num_cu_[dev] = cu;
    return cu;
  }
  void trim() {
    std::lock_guard<std::mutex> lock(mu_);
    for (auto& dev : free_)
      for (auto& b : dev.second) (void)hipFree(b.second);
    free_.clear();
    held_ = 0;
    for (auto& b : host_) (void)hipHostFree(b.second);
    host_.clear();
    host_held_ = 0;
  }
  size_t held() {
    std::lock_guard<std::mutex> lock(mu_);
    return held_;
  }

 private:
  DevicePool() {
    const char* on = std::getenv("MOE_POOL");
    enabled_ = !(on != nullptr && std::atoi(on) == 0);
    const char* gb = std::getenv("MOE_POOL_MAX_GB");
    // (what the pool holds is invisible to the other allocators of the process -- PyTorch, RCCL: 16 GB by default, the two factors of an
    //  N = 26 000 GP and a call's workspaces; only a failed hipMalloc of THIS library trims it, so a co-resident framework that runs
    //  short calls moe_pool_trim(); a negative or unparsable value means 0)
    double cap_gb = gb != nullptr ? std::atof(gb) : 16.0;
    if (!(cap_gb >= 0.0)) cap_gb = 0.0;
    if (cap_gb > 1.0e6) cap_gb = 1.0e6;
    max_held_ = (size_t)(cap_gb * 1e9);
    // MOE_POOL_POISON=1 (tests): a released block is filled with 0xFF bytes before it is pooled -- whoever reads memory it has not
    // written finds NaNs / -1 instead of a plausible zero (a fresh hipMalloc is usually zero, a recycled block is not)
    const char* poison = std::getenv("MOE_POOL_POISON");
    poison_ = poison != nullptr && std::atoi(poison) != 0;
  }
  // sizes in classes, so that shapes a few rows apart share blocks: 256 B granules up to 64 KB, 1/16 of the leading power of two above
  static size_t round_size(size_t bytes) {
    if (bytes <= 65536) return (bytes + 255) / 256 * 256 + (bytes == 0 ? 256 : 0);
    size_t pow2 = 65536;
    while (pow2 * 2 <= bytes) pow2 *= 2;
    const size_t gran = pow2 / 16;
    return (bytes + gran - 1) / gran * gran;
  }
  std::mutex mu_;
  std::map<int, std::multimap<size_t, void*>> free_;
  std::map<int, std::vector<hipStream_t>> streams_;
  std::multimap<size_t, void*> host_;
  size_t host_held_ = 0;
  static constexpr size_t kMaxHostHeld = (size_t)4 << 30;
  std::map<int, int> num_cu_;
  size_t held_ = 0, max_held_ = 0;
  bool enabled_ = true, poison_ = false;
};

// Device buffer of doubles/ints with RAII; grows on demand, never shrinks (state objects are reused across calls).  Its memory comes
// from and returns to the DevicePool.
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;
  size_t bytes = 0;  // size of the pool block behind p
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { DevicePool::get().give(p, bytes); }
  void reserve(size_t n) {
    if (n <= cap) return;
    // (while launches are being recorded, ops already recorded may name the old block: it goes back to the pool after their replay)
    if (Recorder* r = Recorder::current()) {
      if (p != nullptr) r->retired_dev.emplace_back(p, bytes);
    } else {
      DevicePool::get().give(p, bytes);
    }
    p = nullptr;
    cap = 0;
    bytes = 0;
    p = static_cast<T*>(DevicePool::get().take(n * sizeof(T), &bytes));
    cap = n;
  }
  void upload(const T* host, size_t n, hipStream_t s, bool host_pinned = false) {
    reserve(n);
    copy_async(p, host, n * sizeof(T), hipMemcpyHostToDevice, s, host_pinned);
  }
  void download(T* host, size_t n, hipStream_t s) const { copy_async(host, p, n * sizeof(T), hipMemcpyDeviceToHost, s); }
  void swap(DevBuf& o) {
    std::swap(p, o.p);
    std::swap(cap, o.cap);
    std::swap(bytes, o.bytes);
  }
};

// Pinned (page-locked) host staging buffer: small H2D / D2H copies from pageable memory cost 10-20 us each and block the
// host; from pinned memory they are truly asynchronous and a few microseconds.
template <typename T>
struct PinnedBuf {
  T* p = nullptr;
  size_t cap = 0;
  size_t bytes = 0;  // size of the pool block behind p
  PinnedBuf() = default;
  PinnedBuf(const PinnedBuf&) = delete;
  PinnedBuf& operator=(const PinnedBuf&) = delete;
  ~PinnedBuf() { DevicePool::get().give_host(p, bytes); }
  void reserve(size_t n) {
    if (n <= cap) return;
    if (Recorder* r = Recorder::current()) {
      if (p != nullptr) r->retired_host.emplace_back(p, bytes);
    } else {
      DevicePool::get().give_host(p, bytes);
    }
    p = nullptr;
    cap = 0;
    bytes = 0;
    p = static_cast<T*>(DevicePool::get().take_host(n * sizeof(T), &bytes));
    cap = n;
  }
};

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
// Padded dimension the kernels are instantiated for: multiples of 4 up to 16, then 24 and 32 (padded coordinates are 0 and carry
// inverse length 0, so they drop out of every distance and gradient).
inline int padded_dim(int d) { return d <= 16 ? round_up(d, 4) : round_up(d, 8); }

}  // namespace moe
