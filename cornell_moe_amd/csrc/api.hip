// cornell_moe_amd/csrc/api.hip -- the C ABI of libmoe_hip.so (include/moe_hip.h).  Host code only.
#include <memory>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <random>
#include <condition_variable>
#include <thread>

#include "gp.hpp"
#include "kg.hpp"

struct moe_gp {
  moe::GpDev dev;
  // One call at a time per handle: every entry point works in the handle's own device workspaces, pinned staging buffers and
  // stream (the reference's GaussianProcess is genuinely const; this one is const in what it represents only).  ctypes / cgo
  // callers release their runtime's lock around a call, so two host threads sharing a handle would otherwise race.
  std::mutex mu;
  moe_gp(const double* hyper, int cov_type, const double* X, const double* y, const double* noise, const int* derivs, int g,
         int d, int n, int device)
      : dev(hyper, cov_type, X, y, noise, derivs, g, d, n, device) {}
};

namespace {

void set_err(moe_error_t* err, int code, const char* msg, const double* payload) {
  if (!err) return;
  err->code = code;
  std::strncpy(err->message, msg ? msg : "", sizeof(err->message) - 1);
  err->message[sizeof(err->message) - 1] = '\0';
  for (int i = 0; i < 3; ++i) err->payload[i] = payload ? payload[i] : 0.0;
}

template <typename F>
int guarded(moe_error_t* err, F&& f) {
  try {
    if (err) set_err(err, MOE_OK, "", nullptr);
    f();
    return MOE_OK;
  } catch (const moe::Error& e) {
    set_err(err, e.code, e.what(), e.payload);
    return e.code;
  } catch (const std::exception& e) {
    set_err(err, MOE_ERR_RUNTIME, e.what(), nullptr);
    return MOE_ERR_RUNTIME;
  }
}

void require(bool cond, const char* what) {
  if (!cond) throw moe::Error(MOE_ERR_RUNTIME, what);
}

// The device GP behind a handle, with the handle locked for the caller's scope (NULL handles are an error, not a crash).
moe::GpDev& lock_gp(const moe_gp_t* gp_c, std::unique_lock<std::mutex>& lk) {
  require(gp_c != nullptr, "NULL GP handle");
  moe_gp_t* gp = const_cast<moe_gp_t*>(gp_c);
  lk = std::unique_lock<std::mutex>(gp->mu);
  return gp->dev;
}

moe::DerivList no_derivs() {
  moe::DerivList d;
  d.g = 0;
  for (int i = 0; i < moe::kMaxDerivs; ++i) d.idx[i] = 0;
  return d;
}

}  // namespace

// moe_ll_t: the data of a log-likelihood problem + a lazily built device GP that is re-factored per hyper-parameter set.
struct moe_ll {
  int cov_type, g, d, n, device;
  std::vector<double> X, y;
  std::vector<int> derivs;
  std::mutex mu;                   // calls on one handle are serialised, like moe_gp_t's
  std::unique_ptr<moe::GpDev> gp;  // moe_ll_grad's factorisation (with the inverse factor)
  // moe_ll_evaluate: batches of bordered factorisations (kernels.hpp launch_cholesky_batch)
  hipStream_t stream = nullptr;
  moe::DevBuf<double> dX, dYc, dA, dLinv, dNoise, dOut, dScratch;
  moe::DevBuf<int> dInfo;
  ~moe_ll() {
    if (stream) {
      (void)hipSetDevice(device);
      (void)hipStreamDestroy(stream);
    }
  }
};

extern "C" {

const char* moe_version(void) { return "cornell_moe_amd 0.3 (gfx950)"; }

int moe_set_reference_quirks(int on) {
  moe::set_reference_quirks(on);
  return MOE_OK;
}
int moe_get_reference_quirks(void) { return moe::reference_quirks() ? 1 : 0; }
int moe_set_ensemble_launches(int on) {
  moe::set_ensemble_launches(on);
  return MOE_OK;
}
int moe_ensemble_launch_stats(long long* out4) {
  if (out4 == nullptr) return MOE_ERR_INVALID_VALUE;
  moe::ensemble_launch_stats(out4);
  return MOE_OK;
}

int moe_device_count(int* count) {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
  if (count) *count = c;
  return MOE_OK;
}

int moe_device_arch(int device, char* name, int name_len) {
  hipDeviceProp_t prop;
  if (name == nullptr || name_len <= 0) return MOE_ERR_BOUNDS;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return MOE_ERR_RUNTIME;
  std::strncpy(name, prop.gcnArchName, name_len - 1);
  name[name_len - 1] = '\0';
  return MOE_OK;
}

int moe_gp_create(const double* hyperparameters, int cov_type, const double* points_sampled,
                  const double* points_sampled_value, const double* noise_variance, const int* derivatives,
                  int num_derivatives, int dim, int num_sampled, int device, moe_gp_t** gp_out, moe_error_t* err) {
  return guarded(err, [&] {
    require(gp_out != nullptr, "gp_out is NULL");
    *gp_out = nullptr;
    *gp_out = new moe_gp(hyperparameters, cov_type, points_sampled, points_sampled_value, noise_variance, derivatives,
                         num_derivatives, dim, num_sampled, device);
  });
}

int moe_gp_destroy(moe_gp_t* gp) {
  delete gp;
  return MOE_OK;
}

int moe_gp_dim(const moe_gp_t* gp) { return gp ? gp->dev.d : -1; }
int moe_gp_num_sampled(const moe_gp_t* gp) { return gp ? gp->dev.n : -1; }
int moe_gp_num_derivatives(const moe_gp_t* gp) { return gp ? gp->dev.g : -1; }

int moe_gp_add_points(moe_gp_t* gp_c, const double* new_points, const double* new_values, int num_new, moe_error_t* err) {
  return guarded(err, [&] {
    std::unique_lock<std::mutex> lk;
    moe::GpDev& gp = lock_gp(gp_c, lk);
    require(num_new <= 0 || (new_points != nullptr && new_values != nullptr), "NULL argument");
    gp.add_points(new_points, new_values, num_new);
  });
}

int moe_gp_get_factor(const moe_gp_t* gp_c, double* K_chol, double* K_inv_y, double* mean, moe_error_t* err) {
  return guarded(err, [&] {
    std::unique_lock<std::mutex> lk;
    moe::GpDev& gp = lock_gp(gp_c, lk);
    gp.use_device();
    if (K_chol)
      MOE_HIP_CHECK(hipMemcpy2DAsync(K_chol, sizeof(double) * gp.N, gp.dL.p, sizeof(double) * gp.ldL, sizeof(double) * gp.N,
                                     gp.N, hipMemcpyDeviceToHost, gp.stream));
    if (K_inv_y) gp.dKinvY.download(K_inv_y, gp.N, gp.stream);
    MOE_HIP_CHECK(hipStreamSynchronize(gp.stream));
    if (mean) *mean = gp.mean;
  });
}

int moe_gp_mean(const moe_gp_t* gp_c, const double* pts, int num_pts, double* out, moe_error_t* err) {
  return guarded(err, [&] {
    std::unique_lock<std::mutex> lk;
    moe::GpDev& gp = lock_gp(gp_c, lk);
    gp.mean_of_points(pts, num_pts, out, nullptr);
  });
}

int moe_gp_additional_mean(const moe_gp_t* gp, const double* pts, int num_pts, double* out, moe_error_t* err) {
  return moe_gp_mean(gp, pts, num_pts, out, err);  // same quantity; the reference differs only in its temporaries
}

int moe_gp_grad_mean(const moe_gp_t* gp_c, const double* pts, int num_pts, double* out, moe_error_t* err) {
  return guarded(err, [&] {
    std::unique_lock<std::mutex> lk;
    moe::GpDev& gp = lock_gp(gp_c, lk);
    moe::StateHost h;
    moe::compute_state(gp, pts, num_pts, gp.derivs, num_pts, nullptr, 0, false, nullptr, &h);
    moe::host_grad_mean(h, out);
  });
}

int moe_gp_variance(const moe_gp_t* gp_c, const double* pts, int num_pts, double* out, moe_error_t* err) {
  return guarded(err, [&] {
    std::unique_lock<std::mutex> lk;
    moe::GpDev& gp = lock_gp(gp_c, lk);
    if (num_pts * (1 + gp.g) >= moe::device_variance_min_m(false)) {  // r5: hundreds of query points -- the m x m algebra on the device too
      moe::variance_on_device(gp, pts, num_pts, false, out);
      return;
    }
    moe::StateHost h;
    moe::compute_state(gp, pts, num_pts, gp.derivs, 0, nullptr, 0, false, nullptr, &h);
    moe::host_variance(h, out);
  });
}

int moe_gp_cholesky_variance(const moe_gp_t* gp_c, const double* pts, int num_pts, double* out, moe_error_t* err) {
  return guarded(err, [&] {
    std::unique_lock<std::mutex> lk;
    moe::GpDev& gp = lock_gp(gp_c, lk);
    if (num_pts * (1 + gp.g) >= moe::device_variance_min_m(false)) {  // (the factor itself on the device from device_variance_min_m(true) rows)
      moe::variance_on_device(gp, pts, num_pts, true, out);
      return;
    }
    moe::StateHost h;
    moe::compute_state(gp, pts, num_pts, gp.derivs, 0, nullptr, 0, false, nullptr, &h);
    moe::host_variance(h, out);
    const int m = h.lay.m;
    const int lm = moe::host_cholesky(m, out);
    if (lm != 0)
      throw moe::Error(MOE_ERR_SINGULAR,
                       "GP-Variance matrix singular. Check for duplicate points_to_sample or points_to_sample "
                       "duplicating points_sampled with 0 noise.",
                       m, lm);
  });
}

int moe_gp_grad_variance(const moe_gp_t* gp_c, const double* pts, int num_pts, int num_derivs, double* out,
                         moe_error_t* err) {
  return guarded(err, [&] {
    std::unique_lock<std::mutex> lk;
    moe::GpDev& gp = lock_gp(gp_c, lk);
    require(num_derivs >= 0 && num_derivs <= num_pts, "num_derivs must be in [0, num_pts]");
    moe::grad_variance_on_device(gp, pts, num_pts, num_derivs, false, out);  // r6: the m x m x d algebra on the device (query_grad.hip)
  });
}

int moe_gp_grad_cholesky_variance(const moe_gp_t* gp_c, const double* pts, int num_pts, int num_derivs, double* out,
                                  moe_error_t* err) {
  return guarded(err, [&] {
    std::unique_lock<std::mutex> lk;
    moe::GpDev& gp = lock_gp(gp_c, lk);
    require(num_derivs >= 0 && num_derivs <= num_pts, "num_derivs must be in [0, num_pts]");
    moe::grad_variance_on_device(gp, pts, num_pts, num_derivs, true, out);
  });
}

int moe_posterior_mean(const moe_gp_t* gp_c, int num_fidelity, const double* point, double* value, double* grad,
                       moe_error_t* err) {
  return guarded(err, [&] {
    std::unique_lock<std::mutex> lk;
    moe::GpDev& gp = lock_gp(gp_c, lk);
    require(num_fidelity >= 0 && num_fidelity < gp.d, "num_fidelity out of range");
    std::vector<double> pt(gp.d, 1.0);  // fidelity coordinates pinned to 1 (gpp_knowledge_gradient_optimization.cpp:353-357)
    for (int i = 0; i < gp.d - num_fidelity; ++i) pt[i] = point[i];
    double mu;
    std::vector<double> g(gp.d);
    gp.mean_of_points(pt.data(), 1, &mu, grad ? g.data() : nullptr);
    if (value) *value = -mu;
    if (grad)
      for (int i = 0; i < gp.d - num_fidelity; ++i) grad[i] = -g[i];
  });
}

int moe_normal_draws(unsigned int seed, long long count, double* out) {
  // NormalRNG (gpp_random.hpp:204-303) = boost::mt19937 + boost::normal_distribution<double>.  The engine is bit-identical to
  // std::mt19937; the normal algorithm is Boost-version dependent and the reference pins no draws.  What IS
  // reproduced, draw for draw, is the reference as it builds in this repository (oracle/_ref: std-backed Boost shim, i.e.
  // libstdc++'s std::normal_distribution -- Marsaglia's polar method on two generate_canonical<double, 53> uniforms, the
  // second variate saved for the next call); tests/test_oracle.py pins this stream to oracle/_ref's NormalRNG.  Written out
  // here so that the stream does not depend on the C++ library this file is compiled against.
  if (count > 0 && out == nullptr) return MOE_ERR_RUNTIME;
  std::mt19937 eng(seed);
  auto canonical = [&eng]() {
    // std::generate_canonical<double, 53>(mt19937): two 32-bit words, sum and scaling in double, as libstdc++ evaluates them
    double sum = static_cast<double>(eng());
    sum += static_cast<double>(eng()) * 4294967296.0;
    double ret = sum / 18446744073709551616.0;
    if (ret >= 1.0) ret = std::nextafter(1.0, 0.0);
    return ret;
  };
  bool saved_available = false;
  double saved = 0.0;
  for (long long i = 0; i < count; ++i) {
    if (saved_available) {
      saved_available = false;
      out[i] = saved;
      continue;
    }
    double x, y, r2;
    do {
      x = 2.0 * canonical() - 1.0;
      y = 2.0 * canonical() - 1.0;
      r2 = x * x + y * y;
    } while (r2 > 1.0 || r2 == 0.0);
    const double mult = std::sqrt(-2.0 * std::log(r2) / r2);
    saved = x * mult;
    saved_available = true;
    out[i] = y * mult;
  }
  return MOE_OK;
}

int moe_gp_mix_covariance(const moe_gp_t* gp_c, const double* pts, int num_pts, const int* derivs2, int g2, double* out,
                          moe_error_t* err) {
  return guarded(err, [&] {
    std::unique_lock<std::mutex> lk;
    moe::GpDev& gp = lock_gp(gp_c, lk);
    gp.use_device();
    moe::DerivList d2 = no_derivs();
    require(g2 >= 0 && g2 <= moe::kMaxDerivs, "g2 out of range");
    d2.g = g2;
    for (int i = 0; i < g2; ++i) d2.idx[i] = derivs2[i];
    const std::vector<double> P = gp.padded(pts, num_pts);
    gp.dPts.upload(P.data(), P.size(), gp.stream);
    const size_t total = (size_t)gp.N * num_pts * (1 + g2);
    gp.dE.reserve(total);
    moe::launch_cov_build(gp.cp, gp.dX.p, gp.n, gp.derivs, gp.dPts.p, num_pts, d2, nullptr, gp.dE.p, gp.N, 0, gp.stream);
    gp.dE.download(out, total, gp.stream);
    MOE_HIP_CHECK(hipStreamSynchronize(gp.stream));
  });
}

int moe_cov_build_probe(const moe_gp_t* gp_c, const double* pts, int num_pts, int repeat, double* avg_ms,
                        double* bytes_per_launch, moe_error_t* err) {
  return guarded(err, [&] {
    std::unique_lock<std::mutex> lk;
    moe::GpDev& gp = lock_gp(gp_c, lk);
    gp.use_device();
    const std::vector<double> P = gp.padded(pts, num_pts);
    moe::DevBuf<double> dP, dOut;
    dP.upload(P.data(), P.size(), gp.stream);
    dOut.reserve((size_t)gp.N * num_pts);
    hipEvent_t e0, e1;
    MOE_HIP_CHECK(hipEventCreate(&e0));
    MOE_HIP_CHECK(hipEventCreate(&e1));
    moe::launch_cov_build(gp.cp, gp.dX.p, gp.n, gp.derivs, dP.p, num_pts, no_derivs(), nullptr, dOut.p, gp.N, 0, gp.stream, true);
    MOE_HIP_CHECK(hipEventRecord(e0, gp.stream));
    for (int r = 0; r < repeat; ++r)
      moe::launch_cov_build(gp.cp, gp.dX.p, gp.n, gp.derivs, dP.p, num_pts, no_derivs(), nullptr, dOut.p, gp.N, 0, gp.stream, true);
    MOE_HIP_CHECK(hipEventRecord(e1, gp.stream));
    MOE_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    MOE_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (avg_ms) *avg_ms = ms / std::max(repeat, 1);
    if (bytes_per_launch)
      *bytes_per_launch = 8.0 * ((double)gp.n * gp.d + (double)num_pts * gp.d + (double)gp.N * num_pts);
  });
}

int moe_kxx_build_probe(const moe_gp_t* gp_c, int repeat, double* avg_ms, double* bytes_per_launch, moe_error_t* err) {
  return guarded(err, [&] {
    std::unique_lock<std::mutex> lk;
    moe::GpDev& gp = lock_gp(gp_c, lk);
    gp.use_device();
    moe::DevBuf<double> dOut;
    dOut.reserve((size_t)gp.ldL * gp.N);
    hipEvent_t e0, e1;
    MOE_HIP_CHECK(hipEventCreate(&e0));
    MOE_HIP_CHECK(hipEventCreate(&e1));
    auto launch = [&] {  // exactly the call of GpDev::rebuild (gp.hip): K(X, X) + noise on the diagonal, leading dimension ldL
      moe::launch_cov_build(gp.cp, gp.dX.p, gp.n, gp.derivs, gp.dX.p, gp.n, gp.derivs, gp.dNoise.p, dOut.p, gp.ldL, 0, gp.stream,
                            false, true);
    };
    launch();
    MOE_HIP_CHECK(hipEventRecord(e0, gp.stream));
    for (int r = 0; r < repeat; ++r) launch();
    MOE_HIP_CHECK(hipEventRecord(e1, gp.stream));
    MOE_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    MOE_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (avg_ms) *avg_ms = ms / std::max(repeat, 1);
    // SURVEY 8(d), symmetric case: 8 [n d + N (N + 1) / 2] -- the points once, the lower triangle once (r4: what the launch writes)
    if (bytes_per_launch) *bytes_per_launch = 8.0 * ((double)gp.n * gp.d + 0.5 * (double)gp.N * ((double)gp.N + 1.0));
  });
}

namespace {
// Sustained FP64 FMA rate of the whole chip: 8 independent dependent-FMA chains per lane, 16 wavefronts per CU.
__global__ __launch_bounds__(256) void fp64_rate_kernel(double* __restrict__ out, double a, double b, int iters) {
  double x0 = a + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      x0 = fma(x0, a, b);
      x1 = fma(x1, a, b);
      x2 = fma(x2, a, b);
      x3 = fma(x3, a, b);
      x4 = fma(x4, a, b);
      x5 = fma(x5, a, b);
      x6 = fma(x6, a, b);
      x7 = fma(x7, a, b);
    }
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7));
}
}  // namespace

int moe_debug_fp64_rate(int device, double* tflops, moe_error_t* err) {
  return guarded(err, [&] {
    require(tflops != nullptr, "NULL argument");
    MOE_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    MOE_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    const int blocks = prop.multiProcessorCount * 4, iters = 20000;
    moe::DevBuf<double> dOut;
    dOut.reserve((size_t)blocks * 256);
    hipStream_t s = nullptr;
    MOE_HIP_CHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    MOE_HIP_CHECK(hipEventCreate(&e0));
    MOE_HIP_CHECK(hipEventCreate(&e1));
    MOE_LAUNCH_NOW(fp64_rate_kernel, dim3(blocks), dim3(256), 0, s, dOut.p, 0.999999, 1.0e-6, 2000);  // warm-up, clocks up
    MOE_HIP_CHECK(hipEventRecord(e0, s));
    MOE_LAUNCH_NOW(fp64_rate_kernel, dim3(blocks), dim3(256), 0, s, dOut.p, 0.999999, 1.0e-6, iters);
    MOE_HIP_CHECK(hipEventRecord(e1, s));
    MOE_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    MOE_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipStreamDestroy(s);
    *tflops = 2.0 * 64.0 * (double)iters * (double)blocks * 256.0 / (ms * 1e-3) / 1e12;
  });
}

int moe_debug_cholesky(int n, const double* a, int device, double* chol, double* chol_inv, int* info, moe_error_t* err) {
  return guarded(err, [&] {
    require(n > 0, "n must be positive");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
      throw moe::Error(MOE_ERR_RUNTIME, "no HIP device visible: libmoe_hip has no CPU fallback");
    MOE_HIP_CHECK(hipSetDevice(device));
    moe::DevBuf<double> dA, dLinv, dWork;
    moe::DevBuf<int> dInfo;
    hipStream_t s = nullptr;
    dA.upload(a, (size_t)n * n, s);
    dLinv.reserve((size_t)n * n);
    dWork.reserve(moe::cholesky_work_doubles(n));
    dInfo.reserve(1);
    moe::launch_cholesky_and_inverse(n, dA.p, n, dLinv.p, n, dWork.p, dInfo.p, s);
    int inf = 0;
    dInfo.download(&inf, 1, s);
    if (chol) dA.download(chol, (size_t)n * n, s);
    if (chol_inv) dLinv.download(chol_inv, (size_t)n * n, s);
    MOE_HIP_CHECK(hipStreamSynchronize(s));
    if (info) *info = inf;
    if (inf != 0) throw moe::Error(MOE_ERR_SINGULAR, "matrix singular in device Cholesky", n, inf);
  });
}

int moe_debug_math(int n, const double* x, int device, double* exp_neg, double* sqrt_out, moe_error_t* err) {
  return guarded(err, [&] {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
      throw moe::Error(MOE_ERR_RUNTIME, "no HIP device visible: libmoe_hip has no CPU fallback");
    MOE_HIP_CHECK(hipSetDevice(device));
    moe::DevBuf<double> dx, de, dr;
    dx.upload(x, n, nullptr);
    de.reserve(n);
    dr.reserve(n);
    moe::launch_debug_math(dx.p, n, de.p, dr.p, nullptr);
    de.download(exp_neg, n, nullptr);
    dr.download(sqrt_out, n, nullptr);
    MOE_HIP_CHECK(hipStreamSynchronize(nullptr));
  });
}

int moe_last_kernel_ms(const moe_gp_t* gp, double* out5) {
  if (gp == nullptr || out5 == nullptr) return MOE_ERR_RUNTIME;
  std::lock_guard<std::mutex> lk(const_cast<moe_gp_t*>(gp)->mu);
  for (int i = 0; i < 5; ++i) out5[i] = gp->dev.last_ms[i];
  return MOE_OK;
}

int moe_last_kernel_info(const moe_gp_t* gp, int* out8) {
  if (gp == nullptr || out8 == nullptr) return MOE_ERR_RUNTIME;
  std::lock_guard<std::mutex> lk(const_cast<moe_gp_t*>(gp)->mu);
  for (int i = 0; i < 8; ++i) out8[i] = gp->dev.last_info[i];
  return MOE_OK;
}

int moe_ei(const moe_gp_t* gp_c, const double* points_to_sample, const double* points_being_sampled, int num_to_sample,
           int num_being_sampled, int num_mc, double best_so_far, const double* normals, double* ei, double* grad_ei,
           moe_error_t* err) {
  return guarded(err, [&] {
    std::unique_lock<std::mutex> lk;
    moe::GpDev& gp = lock_gp(gp_c, lk);
    moe::ei_evaluate(gp, points_to_sample, points_being_sampled, num_to_sample, num_being_sampled, num_mc, best_so_far,
                     normals, ei, grad_ei);
  });
}

int moe_ei_batch(const moe_gp_t* gp_c, const double* points_to_sample_all, int num_evals, const double* points_being_sampled,
                 int num_to_sample, int num_being_sampled, int num_mc, double best_so_far, const double* normals,
                 double* ei, double* grad_ei, moe_error_t* err) {
  return guarded(err, [&] {
    std::unique_lock<std::mutex> lk;
    moe::GpDev& gp = lock_gp(gp_c, lk);
    moe::ei_evaluate_batch(gp, points_to_sample_all, num_evals, points_being_sampled, num_to_sample, num_being_sampled, num_mc,
                           best_so_far, normals, ei, grad_ei);
  });
}

int moe_ei_analytic_batch(const moe_gp_t* gp_c, const double* points, int num_evals, double best_so_far, double* ei,
                          double* grad_ei, moe_error_t* err) {
  return guarded(err, [&] {
    std::unique_lock<std::mutex> lk;
    moe::GpDev& gp = lock_gp(gp_c, lk);
    require(points != nullptr, "NULL argument");
    moe::ei_analytic_batch(gp, points, num_evals, best_so_far, ei, grad_ei);
  });
}

int moe_ei_multistart(const moe_gp_t* gp_c, const moe_gd_params_t* outer_params, const double* domain_bounds,
                      const double* start_points, int num_starts, const double* points_being_sampled, int num_to_sample,
                      int num_being_sampled, int num_mc, double best_so_far, const double* normals, int do_gradient_ascent,
                      double* best_points, double* best_ei, int* found, moe_error_t* err) {
  return guarded(err, [&] {
    std::unique_lock<std::mutex> lk;
    moe::GpDev& gp = lock_gp(gp_c, lk);
    require(outer_params && domain_bounds && start_points && best_points && best_ei && found, "NULL argument");
    moe::ei_multistart(gp, *outer_params, domain_bounds, start_points, num_starts, points_being_sampled, num_to_sample,
                       num_being_sampled, num_mc, best_so_far, normals, do_gradient_ascent, best_points, best_ei, found);
  });
}

int moe_kg_batch(const moe_gp_t* gp_c, int num_fidelity, const moe_gd_params_t* inner_params, const double* domain_bounds,
                 const double* discrete_pts, int num_pts, const double* points_to_sample_all, int num_evals,
                 const double* points_being_sampled, int num_to_sample, int num_being_sampled, int num_mc,
                 double best_so_far, const double* normals, int first_sample, int num_local, int want_grad,
                 double* kg_sum, double* grad_sum, moe_kg_stats_t* stats, moe_error_t* err) {
  return guarded(err, [&] {
    std::unique_lock<std::mutex> lk;
    moe::GpDev& gp = lock_gp(gp_c, lk);
    require(inner_params && domain_bounds && points_to_sample_all && normals && kg_sum, "NULL argument");
    moe::kg_evaluate_batch(gp, num_fidelity, *inner_params, domain_bounds, discrete_pts, num_pts, points_to_sample_all,
                           num_evals, points_being_sampled, num_to_sample, num_being_sampled, num_mc, best_so_far, normals,
                           first_sample, num_local, want_grad != 0, kg_sum, grad_sum, nullptr, stats);
  });
}

int moe_kg_multistart(const moe_gp_t* gp_c, int num_fidelity, const moe_gd_params_t* outer_params,
                      const moe_gd_params_t* inner_params, const double* domain_bounds, const double* discrete_pts, int num_pts,
                      const double* start_points, int num_starts, const double* points_being_sampled, int num_to_sample,
                      int num_being_sampled, int num_mc, double best_so_far, const double* normals, int do_gradient_ascent,
                      double* best_points, double* best_kg, int* found, moe_error_t* err) {
  return guarded(err, [&] {
    std::unique_lock<std::mutex> lk;
    moe::GpDev& gp = lock_gp(gp_c, lk);
    require(outer_params && inner_params && best_points && best_kg && found, "NULL argument");
    moe::kg_multistart(gp, num_fidelity, *outer_params, *inner_params, domain_bounds, discrete_pts, num_pts, start_points,
                       num_starts, points_being_sampled, num_to_sample, num_being_sampled, num_mc, best_so_far, normals,
                       do_gradient_ascent, best_points, best_kg, found);
  });
}

namespace {
std::vector<moe::GpDev*> ensemble(const moe_gp_t* const* gps, int num_mcmc) {
  require(gps != nullptr && num_mcmc > 0, "empty MCMC ensemble");
  std::vector<moe::GpDev*> v(num_mcmc);
  for (int i = 0; i < num_mcmc; ++i) {
    require(gps[i] != nullptr, "NULL GP handle in the MCMC ensemble");
    v[i] = &const_cast<moe_gp_t*>(gps[i])->dev;
  }
  return v;
}

// Every member of an ensemble locked for the caller's scope, in address order (two calls sharing members cannot deadlock).
std::vector<std::unique_lock<std::mutex>> lock_ensemble(const moe_gp_t* const* gps, int num_mcmc) {
  std::vector<moe_gp_t*> hs;
  for (int i = 0; gps != nullptr && i < num_mcmc; ++i)
    if (gps[i] != nullptr) hs.push_back(const_cast<moe_gp_t*>(gps[i]));
  std::sort(hs.begin(), hs.end());
  hs.erase(std::unique(hs.begin(), hs.end()), hs.end());
  std::vector<std::unique_lock<std::mutex>> locks;
  for (moe_gp_t* h : hs) locks.emplace_back(h->mu);
  return locks;
}
}  // namespace

// ---- one node, several devices, no torch: SURVEY 8b "multistart drivers taking num_devices" ----
// gps[num_devices] are handles of the SAME GP built on different devices (moe_gp_create(..., device = k, ...)); one host
// thread per handle drives its device.  shard_mode 0 (restarts): evaluation e runs whole on handle e % num_devices
// (round-robin, like omp schedule(static,1) over the starts, gpp_optimization.hpp:1481-1535) -- results are those of
// moe_kg_batch, bit for bit.  shard_mode 1 (MC samples): every handle evaluates every point set on its contiguous
// EVEN-ALIGNED slice of the samples (antithetic pairs stay together) and the per-handle sums are added on the host in handle
// order -- a fixed-order reduction of num_evals x (1 + q d) doubles, the all_reduce of the multi-process path
// (cornell_moe_amd/dist.py) done in shared memory.
int moe_kg_batch_multi(const moe_gp_t* const* gps, int num_devices, int shard_mode, int num_fidelity,
                       const moe_gd_params_t* inner_params, const double* domain_bounds, const double* discrete_pts, int num_pts,
                       const double* points_to_sample_all, int num_evals, const double* points_being_sampled,
                       int num_to_sample, int num_being_sampled, int num_mc, double best_so_far, const double* normals,
                       int want_grad, double* kg_sum, double* grad_sum, moe_kg_stats_t* stats, moe_error_t* err) {
  return guarded(err, [&] {
    require(gps != nullptr && num_devices > 0, "need at least one GP handle");
    require(inner_params && domain_bounds && points_to_sample_all && normals && kg_sum, "NULL argument");
    require(!want_grad || grad_sum != nullptr, "grad_sum is NULL");
    require(shard_mode == 0 || shard_mode == 1, "shard_mode must be 0 (restarts) or 1 (MC samples)");
    require(num_evals > 0, "num_evals must be positive");
    const auto locks = lock_ensemble(gps, num_devices);
    require((int)locks.size() == num_devices, "the handles must be distinct and non-NULL");
    const int W = num_devices, E = num_evals;
    const int d = gps[0]->dev.d, qd = num_to_sample * d;
    for (int k = 1; k < W; ++k)
      require(gps[k]->dev.d == d && gps[k]->dev.n == gps[0]->dev.n && gps[k]->dev.g == gps[0]->dev.g,
              "the handles must hold the same GP");
    // work lists
    std::vector<std::vector<int>> mine(W);
    std::vector<int> first(W, 0), count(W, num_mc);
    if (shard_mode == 0) {
      for (int e = 0; e < E; ++e) mine[e % W].push_back(e);
    } else {
      const int pairs = (num_mc + 1) / 2, base = pairs / W, rem = pairs % W;
      for (int k = 0; k < W; ++k) {
        const int p0 = k * base + std::min(k, rem), p1 = p0 + base + (k < rem ? 1 : 0);
        first[k] = 2 * p0;
        count[k] = std::max(std::min(2 * p1, num_mc) - first[k], 0);
        for (int e = 0; e < E; ++e) mine[k].push_back(e);
      }
    }
    std::vector<std::vector<double>> ks(W), gs(W), xs(W);
    std::vector<moe_kg_stats_t> st(W);
    std::vector<moe::Error> errors(W, moe::Error(MOE_OK, ""));
    std::vector<char> failed(W, 0);
    std::vector<std::thread> threads;
    threads.reserve(W);
    // (a thread that cannot be started -- std::system_error under thread exhaustion -- must not unwind past joinable threads:
    //  that would be std::terminate; the started ones are joined first and the error is returned as a status code)
    struct JoinAll {
      std::vector<std::thread>& t;
      ~JoinAll() {
        for (std::thread& th : t)
          if (th.joinable()) th.join();
      }
    } join_all{threads};
    for (int k = 0; k < W; ++k) {
      const int ne = (int)mine[k].size();
      st[k] = moe_kg_stats_t{};
      if (ne == 0 || count[k] == 0) continue;
      ks[k].assign(ne, 0.0);
      gs[k].assign(want_grad ? (size_t)ne * qd : 0, 0.0);
      xs[k].resize((size_t)ne * qd);
      for (int j = 0; j < ne; ++j)
        std::copy(points_to_sample_all + (size_t)mine[k][j] * qd, points_to_sample_all + (size_t)(mine[k][j] + 1) * qd,
                  &xs[k][(size_t)j * qd]);
      threads.emplace_back([&, k, ne] {
        try {
          moe::kg_evaluate_batch(const_cast<moe_gp_t*>(gps[k])->dev, num_fidelity, *inner_params, domain_bounds, discrete_pts,
                                 num_pts, xs[k].data(), ne, points_being_sampled, num_to_sample, num_being_sampled, num_mc,
                                 best_so_far, normals, first[k], count[k], want_grad != 0, ks[k].data(),
                                 want_grad ? gs[k].data() : nullptr, nullptr, &st[k]);
        } catch (const moe::Error& e) {
          errors[k] = e;
          failed[k] = 1;
        } catch (const std::exception& e) {
          errors[k] = moe::Error(MOE_ERR_RUNTIME, e.what());
          failed[k] = 1;
        }
      });
    }
    for (std::thread& t : threads) t.join();
    for (int k = 0; k < W; ++k)
      if (failed[k]) throw errors[k];
    std::fill(kg_sum, kg_sum + E, 0.0);
    if (want_grad) std::fill(grad_sum, grad_sum + (size_t)E * qd, 0.0);
    moe_kg_stats_t total{};
    for (int k = 0; k < W; ++k) {  // handle order: a fixed-order reduction
      for (size_t j = 0; j < ks[k].size(); ++j) {
        const int e = mine[k][j];
        kg_sum[e] += ks[k][j];
        if (want_grad)
          for (int c = 0; c < qd; ++c) grad_sum[(size_t)e * qd + c] += gs[k][j * qd + c];
      }
      total.posterior_mean_evals += st[k].posterior_mean_evals;
      total.posterior_grad_evals += st[k].posterior_grad_evals;
      total.ms_state = std::max(total.ms_state, st[k].ms_state);  // the devices run side by side: the slowest one counts
      total.ms_mc = std::max(total.ms_mc, st[k].ms_mc);
      total.ms_tail = std::max(total.ms_tail, st[k].ms_tail);
    }
    if (stats) *stats = total;
  });
}

int moe_kg_mcmc_batch(const moe_gp_t* const* gps, int num_mcmc, int num_fidelity, const moe_gd_params_t* inner_params,
                      const double* domain_bounds, const double* discrete_pts_all, int num_pts,
                      const double* points_to_sample_all, int num_evals, const double* points_being_sampled, int num_to_sample,
                      int num_being_sampled, int num_mc, const double* best_so_far, const double* normals, int finalize,
                      int total_num_mcmc, double* kg, double* grad_kg, moe_error_t* err) {
  return guarded(err, [&] {
    require(inner_params && domain_bounds && discrete_pts_all && points_to_sample_all && best_so_far && normals && kg,
            "NULL argument");
    require(num_evals > 0, "num_evals must be positive");
    const auto locks = lock_ensemble(gps, num_mcmc);
    const std::vector<moe::GpDev*> v = ensemble(gps, num_mcmc);
    moe::kg_mcmc_sums(v, num_fidelity, *inner_params, domain_bounds, discrete_pts_all, num_pts, points_to_sample_all, num_evals,
                      points_being_sampled, num_to_sample, num_being_sampled, num_mc, best_so_far, normals, grad_kg != nullptr,
                      kg, grad_kg);
    if (finalize)
      moe::kg_mcmc_finalize(kg, grad_kg, points_to_sample_all, num_evals, num_to_sample, v[0]->d, num_fidelity,
                            total_num_mcmc > 0 ? total_num_mcmc : num_mcmc);
  });
}

int moe_kg_mcmc_finalize(double* kg, double* grad_kg, const double* points_to_sample_all, int num_evals, int num_to_sample,
                         int dim, int num_fidelity, int total_num_mcmc) {
  if (!kg || !points_to_sample_all || num_evals <= 0 || num_to_sample <= 0 || dim <= 0 || num_fidelity < 0 ||
      num_fidelity >= dim || total_num_mcmc <= 0)
    return MOE_ERR_BOUNDS;
  moe::kg_mcmc_finalize(kg, grad_kg, points_to_sample_all, num_evals, num_to_sample, dim, num_fidelity, total_num_mcmc);
  return MOE_OK;
}

int moe_ei_mcmc_batch(const moe_gp_t* const* gps, int num_mcmc, const double* points_to_sample_all, int num_evals,
                      const double* points_being_sampled, int num_to_sample, int num_being_sampled, int num_mc,
                      const double* best_so_far, const double* normals, int analytic, double* ei, double* grad_ei,
                      moe_error_t* err) {
  return guarded(err, [&] {
    require(points_to_sample_all && best_so_far && (analytic || normals), "NULL argument");
    require(num_evals > 0, "num_evals must be positive");
    const auto locks = lock_ensemble(gps, num_mcmc);
    const std::vector<moe::GpDev*> v = ensemble(gps, num_mcmc);
    moe::ei_mcmc_batch(v, points_to_sample_all, num_evals, points_being_sampled, num_to_sample, num_being_sampled, num_mc,
                       best_so_far, normals, analytic != 0, ei, grad_ei);
  });
}

int moe_kg_mcmc_multistart(const moe_gp_t* const* gps, int num_mcmc, int num_fidelity, const moe_gd_params_t* outer_params,
                           const moe_gd_params_t* inner_params, const double* domain_bounds, const double* discrete_pts_all,
                           int num_pts, const double* start_points, int num_starts, const double* points_being_sampled,
                           int num_to_sample, int num_being_sampled, int num_mc, const double* best_so_far,
                           const double* normals, int do_gradient_ascent, double* best_points, double* best_kg, int* found,
                           moe_error_t* err) {
  return guarded(err, [&] {
    require(outer_params && inner_params && domain_bounds && discrete_pts_all && start_points && best_so_far && normals &&
                best_points && best_kg && found,
            "NULL argument");
    const auto locks = lock_ensemble(gps, num_mcmc);
    const std::vector<moe::GpDev*> v = ensemble(gps, num_mcmc);
    moe::kg_mcmc_multistart(v, num_fidelity, *outer_params, *inner_params, domain_bounds, discrete_pts_all, num_pts, start_points,
                            num_starts, points_being_sampled, num_to_sample, num_being_sampled, num_mc, best_so_far, normals,
                            do_gradient_ascent, best_points, best_kg, found);
  });
}

// ---- r5: the outer optimisers on several ranks (include/moe_hip.h: moe_comm_t) ----
namespace {

moe::Comm make_comm(const moe_comm_t* c) {
  require(c != nullptr, "NULL moe_comm_t");
  require(c->world >= 1 && c->rank >= 0 && c->rank < c->world, "moe_comm_t: rank must lie in [0, world)");
  require(c->world == 1 || c->allgather != nullptr, "moe_comm_t: world > 1 needs an allgather function");
  moe::Comm comm;
  comm.rank = c->rank;
  comm.world = c->world;
  moe_allgather_fn fn = c->allgather;
  void* ctx = c->ctx;
  comm.allgather = [fn, ctx](const double* send, double* recv, int count) {
    if (fn(ctx, send, recv, count) != 0) throw moe::Error(MOE_ERR_RUNTIME, "the caller's allgather function reported a failure");
  };
  return comm;
}

// The exchange between the host threads of ONE process that drive several devices: an all-gather through a shared buffer.  Two
// buffers: a rank may enter the next round as soon as it has left this one, but that round cannot complete before every rank has.
struct LocalExchange {
  explicit LocalExchange(int world) : W(world) {}
  int W;
  std::mutex m;
  std::condition_variable cv;
  int arrived = 0, count = 0;
  long generation = 0;
  bool aborted = false;
  std::vector<double> filling, ready;
  void abort() {
    std::lock_guard<std::mutex> lk(m);
    aborted = true;
    cv.notify_all();
  }
  void allgather(int rank, const double* send, double* recv, int n) {
    std::unique_lock<std::mutex> lk(m);
    if (aborted) throw moe::Error(MOE_ERR_RUNTIME, "another worker of the multi-device optimisation failed");
    if (arrived == 0) {
      count = n;
      filling.assign((size_t)W * n, 0.0);
    }
    if (n != count) {
      aborted = true;
      cv.notify_all();
      throw moe::Error(MOE_ERR_RUNTIME, "multi-device optimisation: the workers disagree on the size of an exchange", n, count, 0);
    }
    std::copy(send, send + n, filling.begin() + (size_t)rank * n);
    const long mine = generation;
    if (++arrived == W) {
      arrived = 0;
      ready.swap(filling);
      ++generation;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return generation != mine || aborted; });
      if (generation == mine) throw moe::Error(MOE_ERR_RUNTIME, "another worker of the multi-device optimisation failed");
    }
    std::copy(ready.begin(), ready.begin() + (size_t)W * n, recv);
  }
};

// One host thread per worker, each running `body(rank, comm)`; the first failure (by rank) is rethrown after all have been joined.
void run_workers(int W, const std::function<void(int, const moe::Comm&)>& body) {
  LocalExchange ex(W);
  std::vector<moe::Error> errors(W, moe::Error(MOE_OK, ""));
  std::vector<char> failed(W, 0);
  std::vector<std::thread> threads;
  threads.reserve(W);
  struct JoinAll {
    std::vector<std::thread>& t;
    LocalExchange& ex;
    ~JoinAll() {
      bool pending = false;
      for (std::thread& th : t) pending = pending || th.joinable();
      if (pending) ex.abort();  // (only reached when a thread could not be started: release the ones already waiting)
      for (std::thread& th : t)
        if (th.joinable()) th.join();
    }
  } join_all{threads, ex};
  for (int k = 0; k < W; ++k) {
    threads.emplace_back([&, k] {
      moe::Comm comm;
      comm.rank = k;
      comm.world = W;
      comm.allgather = [&ex, k](const double* send, double* recv, int n) { ex.allgather(k, send, recv, n); };
      try {
        body(k, comm);
      } catch (const moe::Error& e) {
        errors[k] = e;
        failed[k] = 1;
        ex.abort();
      } catch (const std::exception& e) {
        errors[k] = moe::Error(MOE_ERR_RUNTIME, e.what());
        failed[k] = 1;
        ex.abort();
      }
    });
  }
  for (std::thread& t : threads) t.join();
  // (a worker released by another's failure reports "another worker ... failed": prefer the original error)
  int first = -1;
  for (int k = 0; k < W; ++k)
    if (failed[k] && (first < 0 || (std::string(errors[first].what()).find("another worker") != std::string::npos &&
                                    std::string(errors[k].what()).find("another worker") == std::string::npos)))
      first = k;
  if (first >= 0) throw errors[first];
}

}  // namespace

long long moe_pool_held_bytes(void) { return (long long)moe::DevicePool::get().held(); }

int moe_pool_trim(void) {
  return guarded(nullptr, [&] { moe::DevicePool::get().trim(); });
}

int moe_multistart_trace(double* out, int cap) { return moe::multistart_trace_get(out, cap); }

// The deal-and-exchange step of the multi-rank optimisers on synthetic items -- item i's result is width copies of
// seed + i + j / 1000 -- with no device work: what the CPU tests drive over gloo.  fail_item >= 0: the rank that owns that item
// throws (MOE_ERR_SINGULAR), and every rank must come back with that code.
int moe_debug_sharded_items(const moe_comm_t* comm_c, int n, int width, double seed, int fail_item, double* out, moe_error_t* err) {
  return guarded(err, [&] {
    require(out != nullptr && n >= 0 && width >= 1, "bad argument");
    const moe::Comm comm = make_comm(comm_c);
    moe::sharded_items(comm, n, width, [&](const std::vector<int>& idx, double* out_local) {
      for (size_t k = 0; k < idx.size(); ++k) {
        if (idx[k] == fail_item) throw moe::Error(MOE_ERR_SINGULAR, "synthetic failure", (double)fail_item, 1.0, 2.0);
        for (int j = 0; j < width; ++j) out_local[k * (size_t)width + j] = seed + idx[k] + j / 1000.0;
      }
    }, out);
  });
}

int moe_kg_multistart_comm(const moe_gp_t* gp_c, const moe_comm_t* comm_c, int num_fidelity, const moe_gd_params_t* outer_params,
                           const moe_gd_params_t* inner_params, const double* domain_bounds, const double* discrete_pts,
                           int num_pts, const double* start_points, int num_starts, const double* points_being_sampled,
                           int num_to_sample, int num_being_sampled, int num_mc, double best_so_far, const double* normals,
                           int do_gradient_ascent, double* best_points, double* best_kg, int* found, moe_error_t* err) {
  return guarded(err, [&] {
    std::unique_lock<std::mutex> lk;
    moe::GpDev& gp = lock_gp(gp_c, lk);
    require(outer_params && inner_params && best_points && best_kg && found, "NULL argument");
    const moe::Comm comm = make_comm(comm_c);
    moe::kg_multistart(gp, num_fidelity, *outer_params, *inner_params, domain_bounds, discrete_pts, num_pts, start_points,
                       num_starts, points_being_sampled, num_to_sample, num_being_sampled, num_mc, best_so_far, normals,
                       do_gradient_ascent, best_points, best_kg, found, &comm);
  });
}

int moe_kg_mcmc_multistart_comm(const moe_gp_t* const* local_gps, int num_local, int total_num_mcmc, const moe_comm_t* comm_c,
                                int num_fidelity, const moe_gd_params_t* outer_params, const moe_gd_params_t* inner_params,
                                const double* domain_bounds, const double* discrete_pts_local, int num_pts,
                                const double* start_points, int num_starts, const double* points_being_sampled,
                                int num_to_sample, int num_being_sampled, int num_mc, const double* best_so_far_local,
                                const double* normals, int do_gradient_ascent, double* best_points, double* best_kg, int* found,
                                moe_error_t* err) {
  return guarded(err, [&] {
    require(outer_params && inner_params && domain_bounds && discrete_pts_local && start_points && best_so_far_local && normals &&
                best_points && best_kg && found,
            "NULL argument");
    const auto locks = lock_ensemble(local_gps, num_local);
    const std::vector<moe::GpDev*> v = ensemble(local_gps, num_local);
    const moe::Comm comm = make_comm(comm_c);
    moe::kg_mcmc_multistart(v, num_fidelity, *outer_params, *inner_params, domain_bounds, discrete_pts_local, num_pts, start_points,
                            num_starts, points_being_sampled, num_to_sample, num_being_sampled, num_mc, best_so_far_local, normals,
                            do_gradient_ascent, best_points, best_kg, found, total_num_mcmc, &comm);
  });
}

int moe_kg_multistart_multi(const moe_gp_t* const* gps, int num_devices, int num_fidelity, const moe_gd_params_t* outer_params,
                            const moe_gd_params_t* inner_params, const double* domain_bounds, const double* discrete_pts,
                            int num_pts, const double* start_points, int num_starts, const double* points_being_sampled,
                            int num_to_sample, int num_being_sampled, int num_mc, double best_so_far, const double* normals,
                            int do_gradient_ascent, double* best_points, double* best_kg, int* found, moe_error_t* err) {
  return guarded(err, [&] {
    require(gps != nullptr && num_devices > 0, "need at least one GP handle");
    require(outer_params && inner_params && best_points && best_kg && found, "NULL argument");
    const auto locks = lock_ensemble(gps, num_devices);
    require((int)locks.size() == num_devices, "the handles must be distinct and non-NULL");
    const int W = num_devices, qd = num_to_sample * gps[0]->dev.d;
    for (int k = 1; k < W; ++k)
      require(gps[k]->dev.d == gps[0]->dev.d && gps[k]->dev.n == gps[0]->dev.n && gps[k]->dev.g == gps[0]->dev.g,
              "the handles must hold the same GP");
    std::vector<std::vector<double>> pts(W, std::vector<double>((size_t)std::max(qd, 1)));
    std::vector<double> val(W, 0.0);
    std::vector<int> fnd(W, 0);
    run_workers(W, [&](int k, const moe::Comm& comm) {
      moe::kg_multistart(const_cast<moe_gp_t*>(gps[k])->dev, num_fidelity, *outer_params, *inner_params, domain_bounds, discrete_pts,
                         num_pts, start_points, num_starts, points_being_sampled, num_to_sample, num_being_sampled, num_mc,
                         best_so_far, normals, do_gradient_ascent, pts[k].data(), &val[k], &fnd[k], &comm);
    });
    std::copy(pts[0].begin(), pts[0].begin() + qd, best_points);  // (every worker holds the same answer)
    *best_kg = val[0];
    *found = fnd[0];
  });
}

int moe_kg_mcmc_multistart_multi(const moe_gp_t* const* gps, int num_mcmc, int num_workers, int num_fidelity,
                                 const moe_gd_params_t* outer_params, const moe_gd_params_t* inner_params,
                                 const double* domain_bounds, const double* discrete_pts_all, int num_pts,
                                 const double* start_points, int num_starts, const double* points_being_sampled,
                                 int num_to_sample, int num_being_sampled, int num_mc, const double* best_so_far,
                                 const double* normals, int do_gradient_ascent, double* best_points, double* best_kg, int* found,
                                 moe_error_t* err) {
  return guarded(err, [&] {
    require(outer_params && inner_params && domain_bounds && discrete_pts_all && start_points && best_so_far && normals &&
                best_points && best_kg && found,
            "NULL argument");
    require(num_workers >= 1 && num_workers <= num_mcmc, "num_workers must lie in [1, num_mcmc]");
    const auto locks = lock_ensemble(gps, num_mcmc);
    const std::vector<moe::GpDev*> all = ensemble(gps, num_mcmc);
    const int W = num_workers, d = all[0]->d, qd = num_to_sample * d;
    require(num_fidelity >= 0 && num_fidelity < d, "num_fidelity out of range");
    const size_t disc_stride = (size_t)num_pts * (d - num_fidelity);
    // worker k holds members k, k + W, ...: its rows of the per-member arrays, gathered
    std::vector<std::vector<moe::GpDev*>> mem(W);
    std::vector<std::vector<double>> disc(W), best(W), pts(W, std::vector<double>((size_t)std::max(qd, 1)));
    for (int g = 0; g < num_mcmc; ++g) {
      mem[g % W].push_back(all[g]);
      disc[g % W].insert(disc[g % W].end(), discrete_pts_all + g * disc_stride, discrete_pts_all + (g + 1) * disc_stride);
      best[g % W].push_back(best_so_far[g]);
    }
    std::vector<double> val(W, 0.0);
    std::vector<int> fnd(W, 0);
    run_workers(W, [&](int k, const moe::Comm& comm) {
      moe::kg_mcmc_multistart(mem[k], num_fidelity, *outer_params, *inner_params, domain_bounds, disc[k].data(), num_pts, start_points,
                              num_starts, points_being_sampled, num_to_sample, num_being_sampled, num_mc, best[k].data(), normals,
                              do_gradient_ascent, pts[k].data(), &val[k], &fnd[k], num_mcmc, &comm);
    });
    std::copy(pts[0].begin(), pts[0].begin() + qd, best_points);
    *best_kg = val[0];
    *found = fnd[0];
  });
}

int moe_ei_mcmc_multistart(const moe_gp_t* const* gps, int num_mcmc, const moe_gd_params_t* outer_params,
                           const double* domain_bounds, const double* start_points, int num_starts,
                           const double* points_being_sampled, int num_to_sample, int num_being_sampled, int num_mc,
                           const double* best_so_far, const double* normals, int do_gradient_ascent, double* best_points,
                           double* best_ei, int* found, moe_error_t* err) {
  return guarded(err, [&] {
    require(outer_params && domain_bounds && start_points && best_so_far && best_points && best_ei && found, "NULL argument");
    const auto locks = lock_ensemble(gps, num_mcmc);
    const std::vector<moe::GpDev*> v = ensemble(gps, num_mcmc);
    moe::ei_mcmc_multistart(v, *outer_params, domain_bounds, start_points, num_starts, points_being_sampled, num_to_sample,
                            num_being_sampled, num_mc, best_so_far, normals, do_gradient_ascent, best_points, best_ei, found);
  });
}

int moe_ll_create(int cov_type, const double* points_sampled, const double* points_sampled_value, const int* derivatives,
                  int num_derivatives, int dim, int num_sampled, int device, moe_ll_t** ll_out, moe_error_t* err) {
  return guarded(err, [&] {
    require(ll_out != nullptr && points_sampled != nullptr && points_sampled_value != nullptr, "NULL argument");
    *ll_out = nullptr;
    if (dim <= 0 || dim > moe::kMaxDimPadded) throw moe::Error(MOE_ERR_BOUNDS, "dim out of range", dim, 1, moe::kMaxDimPadded);
    if (num_derivatives < 0 || num_derivatives > moe::kMaxDerivs)
      throw moe::Error(MOE_ERR_BOUNDS, "num_derivatives out of range", num_derivatives, 0, moe::kMaxDerivs);
    if (num_sampled <= 0) throw moe::Error(MOE_ERR_BOUNDS, "num_sampled must be positive", num_sampled, 1, 1e9);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
      throw moe::Error(MOE_ERR_RUNTIME, "no HIP device visible: libmoe_hip has no CPU fallback");
    auto ll = std::make_unique<moe_ll>();
    ll->cov_type = cov_type;
    ll->g = num_derivatives;
    ll->d = dim;
    ll->n = num_sampled;
    ll->device = device;
    ll->X.assign(points_sampled, points_sampled + (size_t)num_sampled * dim);
    ll->y.assign(points_sampled_value, points_sampled_value + (size_t)num_sampled * (1 + num_derivatives));
    if (num_derivatives > 0) ll->derivs.assign(derivatives, derivatives + num_derivatives);
    *ll_out = ll.release();
  });
}

int moe_ll_destroy(moe_ll_t* ll) {
  delete ll;
  return MOE_OK;
}

}  // extern "C"

namespace {
void ll_evaluate_locked(moe_ll_t* ll, const double* hyperparameters_all, int num_sets, double* values);
void ll_grad_locked(moe_ll_t* ll, const double* hyperparameters, double* grad);
}  // namespace

extern "C" {

int moe_ll_evaluate(moe_ll_t* ll, const double* hyperparameters_all, int num_sets, double* values, moe_error_t* err) {
  return guarded(err, [&] {
    require(ll != nullptr && hyperparameters_all != nullptr && values != nullptr, "NULL argument");
    if (num_sets <= 0) return;
    std::lock_guard<std::mutex> lk(ll->mu);
    ll_evaluate_locked(ll, hyperparameters_all, num_sets, values);
  });
}

}  // extern "C"

namespace {
void ll_evaluate_locked(moe_ll_t* ll, const double* hyperparameters_all, int num_sets, double* values) {
  {
    MOE_HIP_CHECK(hipSetDevice(ll->device));
    const int g1 = 1 + ll->g, d = ll->d, n = ll->n, N = n * g1, stride = 1 + d + g1;
    const int dp = moe::padded_dim(d);
    if (!ll->stream) {
      MOE_HIP_CHECK(hipStreamCreate(&ll->stream));
      std::vector<double> Xp((size_t)n * dp, 0.0), yc(ll->y);
      for (int i = 0; i < n; ++i)
        for (int k = 0; k < d; ++k) Xp[(size_t)i * dp + k] = ll->X[(size_t)i * d + k];
      double mean = 0.0;  // centred on the mean of the function values (gpp_model_selection.cpp:555-563)
      for (int i = 0; i < n; ++i) mean += ll->y[(size_t)i * g1];
      mean /= n;
      for (int i = 0; i < n; ++i) yc[(size_t)i * g1] -= mean;
      ll->dX.upload(Xp.data(), Xp.size(), ll->stream);
      ll->dYc.upload(yc.data(), yc.size(), ll->stream);
      MOE_HIP_CHECK(hipStreamSynchronize(ll->stream));
    }
    hipStream_t s = ll->stream;
    moe::DerivList dl;
    dl.g = ll->g;
    for (int i = 0; i < moe::kMaxDerivs; ++i) dl.idx[i] = (i < ll->g) ? ll->derivs[i] : 0;
    // every set is a bordered (N + 1) x (N + 1) factorisation; as many at a time as fit ~2 GB, at most 64
    const int Np = N + 1;
    const long lda = ((long)Np + 15) / 16 * 16;
    const long mat = lda * Np;
    const int chunk = (int)std::max<long>(1, std::min<long>(64, (long)(2.0e9 / (16.0 * (double)mat))));
    const int B = std::min(chunk, num_sets);
    ll->dA.reserve((size_t)mat * B);
    ll->dLinv.reserve((size_t)mat * B);
    ll->dNoise.reserve((size_t)g1 * B);
    ll->dOut.reserve((size_t)2 * B);
    ll->dInfo.reserve(B);
    std::vector<double> noise((size_t)g1 * B), out((size_t)2 * B);
    std::vector<int> info(B);
    std::vector<moe::CovParams> cps(B);
    for (int i0 = 0; i0 < num_sets; i0 += B) {
      const int nb = std::min(B, num_sets - i0);
      for (int b = 0; b < nb; ++b) {
        const double* h = hyperparameters_all + (size_t)(i0 + b) * stride;
        moe::fill_cov_params(cps[b], ll->cov_type, d, h);
        for (int a = 0; a < g1; ++a) noise[(size_t)b * g1 + a] = h[1 + d + a] + 1.0e-6;  // gpp_model_selection.cpp:546-549
      }
      ll->dNoise.upload(noise.data(), (size_t)g1 * nb, s);
      for (int b = 0; b < nb; ++b)
        moe::launch_cov_build(cps[b], ll->dX.p, n, dl, ll->dX.p, n, dl, ll->dNoise.p + (size_t)b * g1,
                              ll->dA.p + (size_t)b * mat, lda, 0, s);
      moe::launch_ll_border(ll->dA.p, lda, mat, N, ll->dYc.p, nb, s);
      ll->dScratch.reserve(moe::chol_scratch_doubles(Np));
      moe::launch_cholesky_batch(Np, ll->dA.p, lda, mat, ll->dLinv.p, lda, mat, ll->dInfo.p, nb, s, ll->dScratch.p);
      moe::launch_ll_terms_batch(ll->dA.p, lda, mat, N, ll->dOut.p, nb, s);
      ll->dOut.download(out.data(), (size_t)2 * nb, s);
      ll->dInfo.download(info.data(), nb, s);
      MOE_HIP_CHECK(hipStreamSynchronize(s));
      for (int b = 0; b < nb; ++b)
        values[i0 + b] = (info[b] != 0) ? -INFINITY
                                        : -0.5 * out[2 * b + 1] - out[2 * b] - 0.5 * (double)N * 1.8378770664093454835607;
    }
  }
}

void ll_grad_locked(moe_ll_t* ll, const double* hyperparameters, double* grad) {
  {
    const int g1 = 1 + ll->g;
    std::vector<double> noise(g1);
    for (int a = 0; a < g1; ++a) noise[a] = hyperparameters[1 + ll->d + a] + 1.0e-6;  // gpp_model_selection.cpp:546-549
    if (!ll->gp)
      ll->gp.reset(new moe::GpDev(hyperparameters, ll->cov_type, ll->X.data(), ll->y.data(), noise.data(),
                                  ll->derivs.empty() ? nullptr : ll->derivs.data(), ll->g, ll->d, ll->n, ll->device));
    else
      ll->gp->set_hyperparameters(hyperparameters, noise.data());
    ll->gp->grad_log_marginal_likelihood(grad);
  }
}
}  // namespace

extern "C" {

int moe_ll_grad(moe_ll_t* ll, const double* hyperparameters, double* grad, moe_error_t* err) {
  return guarded(err, [&] {
    require(ll != nullptr && hyperparameters != nullptr && grad != nullptr, "NULL argument");
    std::lock_guard<std::mutex> lk(ll->mu);
    ll_grad_locked(ll, hyperparameters, grad);
  });
}

// RestartedGradientDescentHyperparameterOptimizationTensor (gpp_model_selection.hpp:989-1012): where the restarted ascent from x0 ENDS
// (the reference reads the state's current point back; it does not compare it with the start).
int moe_ll_ascend(moe_ll_t* ll, const moe_gd_params_t* gd, const double* domain_log10, const double* x0, double* end_point,
                  moe_error_t* err) {
  return guarded(err, [&] {
    require(ll != nullptr && gd != nullptr && domain_log10 != nullptr && x0 != nullptr && end_point != nullptr, "NULL argument");
    std::lock_guard<std::mutex> lk(ll->mu);
    const int nh = 1 + ll->d + 1 + ll->g;
    std::vector<double> lin(2 * (size_t)nh);
    for (int j = 0; j < 2 * nh; ++j) lin[j] = std::pow(10.0, domain_log10[j]);
    moe::BatchObjective f;
    f.values = [&](const double* x_all, int n, double* values) { ll_evaluate_locked(ll, x_all, n, values); };
    f.grads = [&](const double* x_all, int n, double* grads) {
      for (int i = 0; i < n; ++i) ll_grad_locked(ll, x_all + (size_t)i * nh, grads + (size_t)i * nh);
    };
    std::copy(x0, x0 + nh, end_point);
    moe_gd_params_t g = *gd;
    g.domain_type = MOE_DOMAIN_TENSOR_PRODUCT;
    moe::gradient_ascent_batch(f, g, lin.data(), nh, nh, end_point, 1);
  });
}

// MultistartGradientDescentHyperparameterOptimization / RestartedGradientDescentHyperparameterOptimizationTensor
// (gpp_model_selection.hpp:967-1103) from caller-supplied initial guesses: see include/moe_hip.h.
int moe_ll_multistart(moe_ll_t* ll, const moe_gd_params_t* gd, const double* domain_log10, const double* initial_guesses, int num_starts,
                      double* best_hyperparameters, double* best_value, int* found, moe_error_t* err) {
  return guarded(err, [&] {
    require(ll != nullptr && gd != nullptr && domain_log10 != nullptr && initial_guesses != nullptr && best_hyperparameters != nullptr,
            "NULL argument");
    if (num_starts <= 0) throw moe::Error(MOE_ERR_BOUNDS, "num_multistarts must be > 1", num_starts, 1, 1e9);
    std::lock_guard<std::mutex> lk(ll->mu);
    const int nh = 1 + ll->d + 1 + ll->g;
    std::vector<double> lin(2 * (size_t)nh);
    for (int j = 0; j < 2 * nh; ++j) lin[j] = std::pow(10.0, domain_log10[j]);  // ConvertFromLogToLinearDomain (:862-869)
    moe::BatchObjective f;
    f.values = [&](const double* x_all, int n, double* values) { ll_evaluate_locked(ll, x_all, n, values); };
    f.grads = [&](const double* x_all, int n, double* grads) {
      for (int i = 0; i < n; ++i) ll_grad_locked(ll, x_all + (size_t)i * nh, grads + (size_t)i * nh);
    };
    // InitializeBestKnownPoint (:911-930): the best of the initial guesses seeds the result (found stays false)
    std::vector<double> v0(num_starts);
    f.values(initial_guesses, num_starts, v0.data());
    double best = -INFINITY;
    std::copy(initial_guesses, initial_guesses + nh, best_hyperparameters);
    for (int i = 0; i < num_starts; ++i)
      if (best < v0[i]) {
        best = v0[i];
        std::copy(initial_guesses + (size_t)i * nh, initial_guesses + (size_t)(i + 1) * nh, best_hyperparameters);
      }
    int fnd = 0;
    // MultistartOptimizer (gpp_optimization.hpp:1472-1546): restarted gradient ascent from EVERY guess -- all of them stepped together,
    // one batched pass per step -- the end points compared in start order with the strict test of :1512
    std::vector<double> ends(initial_guesses, initial_guesses + (size_t)num_starts * nh), ve(num_starts);
    moe_gd_params_t g = *gd;
    g.domain_type = MOE_DOMAIN_TENSOR_PRODUCT;
    moe::gradient_ascent_batch(f, g, lin.data(), nh, nh, ends.data(), num_starts);
    f.values(ends.data(), num_starts, ve.data());
    for (int i = 0; i < num_starts; ++i)
      if (ve[i] > best) {
        best = ve[i];
        std::copy(&ends[(size_t)i * nh], &ends[(size_t)(i + 1) * nh], best_hyperparameters);
        fnd = 1;
      }
    if (best_value) *best_value = best;
    if (found) *found = fnd;
  });
}

int moe_posterior_mean_optimize(const moe_gp_t* gp_c, int num_fidelity, const moe_gd_params_t* params,
                                const double* domain_bounds, const double* initial_guess, double* best_point,
                                double* best_value, moe_error_t* err) {
  return guarded(err, [&] {
    std::unique_lock<std::mutex> lk;
    moe::GpDev& gp = lock_gp(gp_c, lk);
    require(params && best_point, "NULL argument");
    moe::posterior_mean_optimize(gp, num_fidelity, *params, domain_bounds, initial_guess, best_point, best_value);
  });
}

int moe_latin_hypercube(unsigned int seed, const double* domain_bounds, int dim, int num_points, double* out) {
  if (dim <= 0 || num_points <= 0 || !domain_bounds || !out) return MOE_ERR_BOUNDS;
  moe::latin_hypercube(seed, domain_bounds, dim, num_points, out);
  return MOE_OK;
}

int moe_kg(const moe_gp_t* gp_c, int num_fidelity, const moe_gd_params_t* inner_params, const double* domain_bounds,
           const double* discrete_pts, int num_pts, const double* points_to_sample, const double* points_being_sampled,
           int num_to_sample, int num_being_sampled, int num_mc, double best_so_far, const double* normals,
           int first_sample, int num_local, int want_grad, double* kg_sum, double* grad_sum, double* best_points,
           moe_kg_stats_t* stats, moe_error_t* err) {
  return guarded(err, [&] {
    std::unique_lock<std::mutex> lk;
    moe::GpDev& gp = lock_gp(gp_c, lk);
    require(inner_params && domain_bounds && points_to_sample && normals && kg_sum, "NULL argument");
    moe::kg_evaluate_batch(gp, num_fidelity, *inner_params, domain_bounds, discrete_pts, num_pts, points_to_sample, 1,
                           points_being_sampled, num_to_sample, num_being_sampled, num_mc, best_so_far, normals,
                           first_sample, num_local, want_grad != 0, kg_sum, grad_sum, best_points, stats);
  });
}

}  // extern "C"
