// cornell_moe_amd/csrc/common.hpp -- shared host-side helpers for libmoe_hip.so (gfx950 only; no CPU fallback).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
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

// Device buffer of doubles/ints with RAII; grows on demand, never shrinks (state objects are reused across calls).
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  void reserve(size_t n) {
    if (n <= cap) return;
    if (p) MOE_HIP_CHECK(hipFree(p));
    p = nullptr;
    MOE_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(T)));
    cap = n;
  }
  void upload(const T* host, size_t n, hipStream_t s) {
    reserve(n);
    if (n) MOE_HIP_CHECK(hipMemcpyAsync(p, host, n * sizeof(T), hipMemcpyHostToDevice, s));
  }
  void download(T* host, size_t n, hipStream_t s) const {
    if (n) MOE_HIP_CHECK(hipMemcpyAsync(host, p, n * sizeof(T), hipMemcpyDeviceToHost, s));
  }
  void swap(DevBuf& o) {
    std::swap(p, o.p);
    std::swap(cap, o.cap);
  }
};

// Pinned (page-locked) host staging buffer: small H2D / D2H copies from pageable memory cost 10-20 us each and block the
// host; from pinned memory they are truly asynchronous and a few microseconds.
template <typename T>
struct PinnedBuf {
  T* p = nullptr;
  size_t cap = 0;
  PinnedBuf() = default;
  PinnedBuf(const PinnedBuf&) = delete;
  PinnedBuf& operator=(const PinnedBuf&) = delete;
  ~PinnedBuf() {
    if (p) (void)hipHostFree(p);
  }
  void reserve(size_t n) {
    if (n <= cap) return;
    if (p) MOE_HIP_CHECK(hipHostFree(p));
    p = nullptr;
    MOE_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&p), n * sizeof(T), hipHostMallocDefault));
    cap = n;
  }
};

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
// Padded dimension the kernels are instantiated for: multiples of 4 up to 16, then 24 and 32 (padded coordinates are 0 and carry
// inverse length 0, so they drop out of every distance and gradient).
inline int padded_dim(int d) { return d <= 16 ? round_up(d, 4) : round_up(d, 8); }

}  // namespace moe
