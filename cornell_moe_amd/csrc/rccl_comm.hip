// cornell_moe_amd/csrc/rccl_comm.hip -- round 6: the exchange of the multi-rank outer optimisers carried by RCCL itself.
//
// moe_comm_t (include/moe_hip.h) is an all-gather of `count` doubles per rank that the caller hands in; until r5 the only provider
// was cornell_moe_amd/dist.py -- a Python callback out of the C++ optimiser loop into torch.distributed.  Here the library provides
// it natively: ncclAllGather / ncclAllReduce on a stream of its own, device staging buffers it owns, one pinned copy in and one out,
// one stream wait per exchange; the optimiser loop (csrc/multistart.hip: sharded_items) never leaves C++.  It replaces the merge the
// reference does under `omp critical` (gpp_optimization.hpp:1537-1545) when the restarts are dealt to processes, one per GPU.
// librccl is resolved at RUN time (dlopen): the library has no link-time dependency on it, and on a box without RCCL the entry
// points return MOE_ERR_RUNTIME.  The unique id is made on rank 0 (moe_rccl_unique_id) and carried to the other ranks by whatever the
// host already has (torch.distributed's gloo group in dist.py, MPI, a file).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstring>
#include <mutex>
#include <string>

#include "common.hpp"

struct moe_rccl {
  int rank = 0, world = 1, device = 0;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  double* dev = nullptr;     // [send (cap) | recv (cap world)]
  double* host = nullptr;    // pinned, same layout
  size_t cap = 0;            // doubles per rank the buffers hold
  long long calls = 0, bytes = 0;
  double seconds = 0.0;
  std::mutex mu;
};

namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string why;
};

RcclApi& api() {
  static RcclApi a;
  static std::once_flag once;
  std::call_once(once, [] {
    // an RCCL the process has loaded already (torch.distributed's) is reused; otherwise the ROCm installation's
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
      a.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
      if (a.handle) break;
    }
    if (!a.handle) {
      const char* env = std::getenv("MOE_RCCL_LIB");
      if (env && *env) a.handle = dlopen(env, RTLD_NOW | RTLD_LOCAL);
      for (size_t i = 0; !a.handle && i < sizeof(names) / sizeof(names[0]); ++i) a.handle = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    }
    if (!a.handle) {
      const char* e = dlerror();
      a.why = std::string("librccl could not be loaded (") + (e ? e : "no message") + ")";
      return;
    }
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(a.handle, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(a.handle, "ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.handle, "ncclCommDestroy"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(a.handle, "ncclAllGather"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(a.handle, "ncclAllReduce"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(a.handle, "ncclGetErrorString"));
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather || !a.AllReduce) {
      a.why = "librccl lacks an expected symbol";
      a.handle = nullptr;
    }
  });
  return a;
}

void need_api() {
  if (!api().handle) throw moe::Error(MOE_ERR_RUNTIME, "native RCCL exchange unavailable: " + api().why);
}

void check(ncclResult_t r, const char* what) {
  if (r != ncclSuccess)
    throw moe::Error(MOE_ERR_RUNTIME, std::string("RCCL error in ") + what + ": " +
                                          (api().GetErrorString ? api().GetErrorString(r) : "code " + std::to_string((int)r)));
}

int fill(moe_error_t* err, int code, const std::string& msg) {
  if (err) {
    err->code = code;
    std::strncpy(err->message, msg.c_str(), sizeof(err->message) - 1);
    err->message[sizeof(err->message) - 1] = 0;
    err->payload[0] = err->payload[1] = err->payload[2] = 0.0;
  }
  return code;
}

template <class F>
int guarded(moe_error_t* err, F&& f) {
  try {
    f();
    return fill(err, MOE_OK, "");
  } catch (const moe::Error& e) {
    return fill(err, e.code, e.what());
  } catch (const std::exception& e) {
    return fill(err, MOE_ERR_RUNTIME, e.what());
  } catch (...) {
    return fill(err, MOE_ERR_RUNTIME, "unknown exception");
  }
}

void reserve(moe_rccl* r, size_t count) {
  if (count <= r->cap) return;
  size_t cap = 1024;
  while (cap < count) cap *= 2;
  if (r->dev) (void)hipFree(r->dev);
  if (r->host) (void)hipHostFree(r->host);
  r->dev = nullptr;
  r->host = nullptr;
  r->cap = 0;
  MOE_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&r->dev), sizeof(double) * cap * (1 + (size_t)r->world)));
  MOE_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&r->host), sizeof(double) * cap * (1 + (size_t)r->world), hipHostMallocDefault));
  r->cap = cap;
}

// moe_allgather_fn: recv[world][count] in rank order; 0 = ok (nothing may propagate into the caller's frames)
int rccl_allgather(void* ctx, const double* send, double* recv, int count) {
  moe_rccl* r = static_cast<moe_rccl*>(ctx);
  try {
    std::lock_guard<std::mutex> lock(r->mu);
    const auto t0 = std::chrono::steady_clock::now();
    MOE_HIP_CHECK(hipSetDevice(r->device));
    reserve(r, (size_t)count);
    const size_t n = (size_t)count, W = (size_t)r->world;
    std::memcpy(r->host, send, sizeof(double) * n);
    MOE_HIP_CHECK(hipMemcpyAsync(r->dev, r->host, sizeof(double) * n, hipMemcpyHostToDevice, r->stream));
    check(api().AllGather(r->dev, r->dev + r->cap, n, ncclDouble, r->comm, r->stream), "ncclAllGather");
    MOE_HIP_CHECK(hipMemcpyAsync(r->host + r->cap, r->dev + r->cap, sizeof(double) * n * W, hipMemcpyDeviceToHost, r->stream));
    MOE_HIP_CHECK(hipStreamSynchronize(r->stream));
    std::memcpy(recv, r->host + r->cap, sizeof(double) * n * W);
    r->calls += 1;
    r->bytes += (long long)(sizeof(double) * n * W);
    r->seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return 0;
  } catch (...) {
    return 1;
  }
}

}  // namespace

extern "C" {

int moe_rccl_unique_id(char* id, moe_error_t* err) {
  return guarded(err, [&] {
    if (id == nullptr) throw moe::Error(MOE_ERR_INVALID_VALUE, "NULL id");
    need_api();
    static_assert(sizeof(ncclUniqueId) == MOE_RCCL_ID_BYTES, "unique id size");
    ncclUniqueId u;
    check(api().GetUniqueId(&u), "ncclGetUniqueId");
    std::memcpy(id, &u, sizeof(u));
  });
}

int moe_rccl_create(const char* id, int rank, int world, int device, moe_rccl_t** out, moe_error_t* err) {
  return guarded(err, [&] {
    if (id == nullptr || out == nullptr) throw moe::Error(MOE_ERR_INVALID_VALUE, "NULL argument");
    if (world < 1 || rank < 0 || rank >= world) throw moe::Error(MOE_ERR_BOUNDS, "rank out of range", rank, 0, world - 1);
    need_api();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
      (void)hipGetLastError();
      throw moe::Error(MOE_ERR_RUNTIME, "native RCCL exchange: no such device");
    }
    MOE_HIP_CHECK(hipSetDevice(device));
    moe_rccl* r = new moe_rccl;
    r->rank = rank;
    r->world = world;
    r->device = device;
    try {
      MOE_HIP_CHECK(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking));
      ncclUniqueId u;
      std::memcpy(&u, id, sizeof(u));
      check(api().CommInitRank(&r->comm, world, u, rank), "ncclCommInitRank");
      reserve(r, 1024);
    } catch (...) {
      moe_rccl_destroy(r);
      throw;
    }
    *out = r;
  });
}

int moe_rccl_comm(moe_rccl_t* r, moe_comm_t* out) {
  if (r == nullptr || out == nullptr) return MOE_ERR_INVALID_VALUE;
  out->rank = r->rank;
  out->world = r->world;
  out->allgather = &rccl_allgather;
  out->ctx = r;
  return MOE_OK;
}

int moe_rccl_allreduce_sum(moe_rccl_t* r, double* inout, int count, moe_error_t* err) {
  return guarded(err, [&] {
    if (r == nullptr || inout == nullptr || count < 0) throw moe::Error(MOE_ERR_INVALID_VALUE, "bad argument");
    std::lock_guard<std::mutex> lock(r->mu);
    const auto t0 = std::chrono::steady_clock::now();
    MOE_HIP_CHECK(hipSetDevice(r->device));
    reserve(r, (size_t)count);
    const size_t n = (size_t)count;
    std::memcpy(r->host, inout, sizeof(double) * n);
    MOE_HIP_CHECK(hipMemcpyAsync(r->dev, r->host, sizeof(double) * n, hipMemcpyHostToDevice, r->stream));
    check(api().AllReduce(r->dev, r->dev, n, ncclDouble, ncclSum, r->comm, r->stream), "ncclAllReduce");
    MOE_HIP_CHECK(hipMemcpyAsync(r->host, r->dev, sizeof(double) * n, hipMemcpyDeviceToHost, r->stream));
    MOE_HIP_CHECK(hipStreamSynchronize(r->stream));
    std::memcpy(inout, r->host, sizeof(double) * n);
    r->calls += 1;
    r->bytes += (long long)(sizeof(double) * n);
    r->seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  });
}

int moe_rccl_stats(const moe_rccl_t* r, long long* calls, long long* bytes, double* seconds) {
  if (r == nullptr) return MOE_ERR_INVALID_VALUE;
  if (calls) *calls = r->calls;
  if (bytes) *bytes = r->bytes;
  if (seconds) *seconds = r->seconds;
  return MOE_OK;
}

void moe_rccl_destroy(moe_rccl_t* r) {
  if (r == nullptr) return;
  (void)hipSetDevice(r->device);
  if (r->stream) (void)hipStreamSynchronize(r->stream);
  if (r->comm && api().CommDestroy) (void)api().CommDestroy(r->comm);
  if (r->dev) (void)hipFree(r->dev);
  if (r->host) (void)hipHostFree(r->host);
  if (r->stream) (void)hipStreamDestroy(r->stream);
  delete r;
}

}  // extern "C"
