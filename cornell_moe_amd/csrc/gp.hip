// cornell_moe_amd/csrc/gp.hip -- see gp.hpp.
#include <chrono>
#include "gp.hpp"

#include <algorithm>

#include "device_cov.hpp"

#include <cstdlib>
#include <cmath>
#include <cstring>

namespace moe {

namespace {
constexpr int kAppendHeadroom = 128;  // rows of room left behind a rebuilt factorisation for add_points
}

GpDev::GpDev(const double* hyper, int cov_type, const double* X_in, const double* y_in, const double* noise_in,
             const int* derivs_in, int g_in, int d_in, int n_in, int device_in)
    : device(device_in), d(d_in), n(n_in), g(g_in) {
  if (d <= 0 || d > kMaxDimPadded) throw Error(MOE_ERR_BOUNDS, "dim out of range", d, 1, kMaxDimPadded);
  if (g < 0 || g > kMaxDerivs) throw Error(MOE_ERR_BOUNDS, "num_derivatives out of range", g, 0, kMaxDerivs);
  if (n <= 0) throw Error(MOE_ERR_BOUNDS, "num_sampled must be positive", n, 1, 1e9);
  if (cov_type != MOE_COV_SQUARE_EXPONENTIAL && cov_type != MOE_COV_MATERN_NU_2P5)
    throw Error(MOE_ERR_INVALID_VALUE, "unknown covariance type", cov_type, 0, 1);
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    throw Error(MOE_ERR_RUNTIME, "no HIP device visible: libmoe_hip has no CPU fallback");
  if (device < 0 || device >= count) throw Error(MOE_ERR_BOUNDS, "device index out of range", device, 0, count - 1);
  dp = padded_dim(d);
  N = n * (1 + g);
  cp.type = cov_type;
  cp.dim = d;
  cp.dp = dp;
  set_covariance(hyper);
  derivs.g = g;
  for (int i = 0; i < kMaxDerivs; ++i) derivs.idx[i] = 0;
  for (int i = 0; i < g; ++i) {
    if (derivs_in[i] < 0 || derivs_in[i] >= d) throw Error(MOE_ERR_BOUNDS, "derivative index out of range", derivs_in[i], 0, d - 1);
    derivs.idx[i] = derivs_in[i];
  }
  X.assign(X_in, X_in + (size_t)n * d);
  y.assign(y_in, y_in + (size_t)N);
  noise.assign(noise_in, noise_in + (1 + g));
  use_device();
  stream = DevicePool::get().take_stream(device);
  {
    const int cu = DevicePool::get().num_cu(device);
    if (cu > 0) num_cu = cu;
  }
  try {
    rebuild();
  } catch (...) {  // (a singular K: no destructor runs for a constructor that throws)
    DevicePool::get().give_stream(device, stream);
    release_side_streams();
    stream = nullptr;
    throw;
  }
}

void fill_cov_params(CovParams& cp, int cov_type, int d, const double* hyper) {
  cp.type = cov_type;
  cp.dim = d;
  cp.dp = padded_dim(d);
  cp.alpha = hyper[0];
  if (!(cp.alpha > 0.0)) throw Error(MOE_ERR_BOUNDS, "alpha must be positive", cp.alpha, 0.0, INFINITY);
  for (int k = 0; k < kMaxDimPadded; ++k) {
    cp.inv_l2[k] = 0.0;
    cp.inv_l[k] = 0.0;
    cp.center[k] = 0.0;
  }
  for (int k = 0; k < d; ++k) {
    const double l = hyper[1 + k];
    if (!(l > 0.0)) throw Error(MOE_ERR_BOUNDS, "length scale must be positive", l, 0.0, INFINITY);  // gpp_covariance.cpp:85-92
    cp.inv_l2[k] = 1.0 / (l * l);
    cp.inv_l[k] = 1.0 / l;
  }
}

void GpDev::set_covariance(const double* hyper) { fill_cov_params(cp, cp.type, d, hyper); }

void GpDev::set_hyperparameters(const double* hyper, const double* noise_in) {
  set_covariance(hyper);
  noise.assign(noise_in, noise_in + (1 + g));
  rebuild();
}

namespace {
// out[0] = sum_i log L_ii, out[1] = yc . K^-1 yc; one workgroup (N is a few thousand at most: two strided passes).
__global__ __launch_bounds__(256) void ll_terms_kernel(const double* __restrict__ L, long ldl, int N,
                                                       const double* __restrict__ yc, const double* __restrict__ KinvY,
                                                       double* __restrict__ out) {
  __shared__ double red[2][256];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < N; i += 256) {
    a += log(L[(long)i + (long)i * ldl]);
    b = fma(yc[i], KinvY[i], b);
  }
  red[0][threadIdx.x] = a;
  red[1][threadIdx.x] = b;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
      red[0][threadIdx.x] += red[0][threadIdx.x + w];
      red[1][threadIdx.x] += red[1][threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = red[0][0];
    out[1] = red[1][0];
  }
}
}  // namespace

namespace {
// Hyper-parameter gradient of the log marginal likelihood, covariance part: with w_ij = alpha_i alpha_j - K^-1_ij on the
// function-value rows,
//   part[block][0]     = sum w_ij  dK_ij/d alpha          (= base / alpha)
//   part[block][1 + k] = sum w_ij  first_ij (x_ik - x_jk)^2   (dK_ij/d l_k = first * diff_k^2 / l_k^3; host scales)
// over the block's 256 rows i x 64 columns j.  Row per thread (K^-1 reads coalesced), the 64 x_j broadcast from LDS.
template <int DP>
__global__ __launch_bounds__(256) void ll_grad_kernel(CovParams cp, const double* __restrict__ X, int n, int g1,
                                                      const double* __restrict__ alpha, const double* __restrict__ Kinv,
                                                      long ldk, double* __restrict__ part) {
  __shared__ double Xj[64][DP];
  __shared__ double aj[64];
  __shared__ double red[4][1 + DP];
  const int j0 = blockIdx.y * 64, nj = min(64, n - j0);
  for (int t = threadIdx.x; t < nj * DP; t += 256) Xj[t / DP][t % DP] = X[(long)(j0 + t / DP) * DP + t % DP];
  if ((int)threadIdx.x < nj) aj[threadIdx.x] = alpha[(long)(j0 + threadIdx.x) * g1];
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  double acc[1 + DP];
#pragma unroll
  for (int k = 0; k <= DP; ++k) acc[k] = 0.0;
  if (i < n) {
    double xi[DP];
#pragma unroll
    for (int k = 0; k < DP; ++k) xi[k] = X[(long)i * DP + k];
    const double ai = alpha[(long)i * g1];
    const double* Krow = Kinv + i;  // (Kinv: the n x n function-value rows / columns of K^-1)
    for (int jj = 0; jj < nj; ++jj) {
      double diff2[DP];
      double r2 = 0.0;
#pragma unroll
      for (int k = 0; k < DP; ++k) {
        const double dlt = xi[k] - Xj[jj][k];
        diff2[k] = dlt * dlt;
        r2 = fma(diff2[k], cp.inv_l2[k], r2);
      }
      const Radial rd = radial_scalars(cp.type, 1.0, r2);  // alpha = 1: base IS dK/d alpha
      const double w = fma(ai, aj[jj], -Krow[(long)(j0 + jj) * ldk]);
      acc[0] = fma(w, rd.base, acc[0]);
      const double wf = w * rd.first;
#pragma unroll
      for (int k = 0; k < DP; ++k) acc[1 + k] = fma(wf, diff2[k], acc[1 + k]);
    }
  }
#pragma unroll
  for (int k = 0; k <= DP; ++k) {
    double v = acc[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if ((int)threadIdx.x <= DP) {
    const int k = threadIdx.x;
    part[((long)blockIdx.y * gridDim.x + blockIdx.x) * (1 + DP) + k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
  }
}

// out[k] = sum over blocks of part[block][k] (k < width), then out[width + a] = sum_i alpha_{i,a}^2 - K^-1_{(i,a),(i,a)}
// (the noise-variance gradients: dK/d sigma_a is the indicator of the rows of observation kind a; kdiag = diag K^-1).  One workgroup.
__global__ __launch_bounds__(256) void ll_grad_finish_kernel(const double* __restrict__ part, int nblocks, int width, int n,
                                                             int g1, const double* __restrict__ alpha,
                                                             const double* __restrict__ kdiag,
                                                             double* __restrict__ out) {
  __shared__ double red[256];
  for (int k = 0; k < width + g1; ++k) {
    double v = 0.0;
    if (k < width) {
      for (int b = threadIdx.x; b < nblocks; b += 256) v += part[(long)b * width + k];
    } else {
      const int a = k - width;
      for (int i = threadIdx.x; i < n; i += 256) {
        const long r = (long)i * g1 + a;
        v += fma(alpha[r], alpha[r], -kdiag[r]);
      }
    }
    red[threadIdx.x] = v;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
      __syncthreads();
    }
    if (threadIdx.x == 0) out[k] = red[0];
    __syncthreads();
  }
}

template <int DP>
void launch_ll_grad(const CovParams& cp, const double* X, int n, int g1, const double* alpha, const double* Kinv, long ldk,
                    const double* kdiag, double* part, double* out, hipStream_t s) {
  dim3 grid((n + 255) / 256, (n + 63) / 64);
  MOE_LAUNCH((ll_grad_kernel<DP>), grid, dim3(256), 0, s, cp, X, n, g1, alpha, Kinv, ldk, part);
  MOE_LAUNCH(ll_grad_finish_kernel, dim3(1), dim3(256), 0, s, (const double*)part, (int)(grid.x * grid.y), 1 + DP, n,
                     g1, alpha, kdiag, out);
  MOE_HIP_CHECK(hipGetLastError());
}
}  // namespace

void GpDev::grad_log_marginal_likelihood(double* grad) {
  use_device();
  if (cp.type == MOE_COV_SQUARE_EXPONENTIAL && g > 0)
    throw Error(MOE_ERR_INVALID_VALUE,
                "hyper-parameter gradient with derivative observations is provided for the Matern-5/2 kernel only "
                "(the kernel the reference's Python boundary builds)");
  // tr(K^-1 dK/d theta) reads K^-1 = L^-T L^-1 on the function-value rows / columns (every (1 + g)-th: the length-scale and
  // amplitude gradients) and on the diagonal (the noise gradients): a Gram matrix of n of L^-1's N columns -- at g = 3 a
  // sixteenth of the entries the full product formed (r4: 9.8 -> 0.7 ms at N = 8000) -- and the N column norms
  const int g1 = 1 + g;
  dVE.reserve((size_t)n * n + (size_t)N);
  double* kdiag = dVE.p + (size_t)n * n;
  launch_tri_gram_strided(N, n, g1, dLinv.p, ldL, dVE.p, n, kdiag, stream);
  const size_t nblocks = (size_t)((n + 255) / 256) * ((n + 63) / 64);
  dE.reserve(nblocks * (1 + dp) + (size_t)(1 + dp + g1));
  double* out = dE.p + nblocks * (1 + dp);
  switch (dp) {
    case 4: launch_ll_grad<4>(cp, dX.p, n, g1, dKinvY.p, dVE.p, n, kdiag, dE.p, out, stream); break;
    case 8: launch_ll_grad<8>(cp, dX.p, n, g1, dKinvY.p, dVE.p, n, kdiag, dE.p, out, stream); break;
    case 12: launch_ll_grad<12>(cp, dX.p, n, g1, dKinvY.p, dVE.p, n, kdiag, dE.p, out, stream); break;
    case 16: launch_ll_grad<16>(cp, dX.p, n, g1, dKinvY.p, dVE.p, n, kdiag, dE.p, out, stream); break;
    case 24: launch_ll_grad<24>(cp, dX.p, n, g1, dKinvY.p, dVE.p, n, kdiag, dE.p, out, stream); break;
    case 32: launch_ll_grad<32>(cp, dX.p, n, g1, dKinvY.p, dVE.p, n, kdiag, dE.p, out, stream); break;
    default: throw Error(MOE_ERR_BOUNDS, "unsupported padded dimension", dp, 4, 16);
  }
  std::vector<double> h((size_t)(1 + dp + g1));
  MOE_HIP_CHECK(hipMemcpyAsync(h.data(), out, sizeof(double) * h.size(), hipMemcpyDeviceToHost, stream));
  MOE_HIP_CHECK(hipStreamSynchronize(stream));
  // 0.5 alpha^T dK alpha - 0.5 tr(K^-1 dK)   (ComputeGradLogLikelihood, gpp_model_selection.cpp:629-677)
  grad[0] = 0.5 * h[0];
  for (int k = 0; k < d; ++k) grad[1 + k] = 0.5 * cp.alpha * h[1 + k] * cp.inv_l2[k] * cp.inv_l[k];
  for (int a = 0; a < g1; ++a) grad[1 + d + a] = 0.5 * h[1 + dp + a];
}

double GpDev::log_marginal_likelihood() {
  use_device();
  // dTmp[0, N) still holds yc = y - mean on the value rows (rebuild), dTmp[N, 2N) is free scratch
  MOE_LAUNCH(ll_terms_kernel, dim3(1), dim3(256), 0, stream, dL.p, ldL, N, dTmp.p, dKinvY.p, dTmp.p + N);
  MOE_HIP_CHECK(hipGetLastError());
  double terms[2] = {0.0, 0.0};
  MOE_HIP_CHECK(hipMemcpyAsync(terms, dTmp.p + N, sizeof(terms), hipMemcpyDeviceToHost, stream));
  MOE_HIP_CHECK(hipStreamSynchronize(stream));
  return -0.5 * terms[1] - terms[0] - 0.5 * (double)N * 1.8378770664093454835607;  // kLog2Pi, gpp_common.hpp:747
}

GpDev::~GpDev() {
  if (stream) {
    (void)hipSetDevice(device);
    DevicePool::get().give_stream(device, stream);
    release_side_streams();
  }
}

void GpDev::use_device() const { MOE_HIP_CHECK(hipSetDevice(device)); }

void GpDev::release_side_streams() {
  DevicePool::get().give_stream(device, side_stream);
  side_stream = nullptr;
}

std::vector<double> GpDev::padded(const double* pts, int k) const {
  std::vector<double> out((size_t)k * dp, 0.0);
  for (int i = 0; i < k; ++i)
    for (int j = 0; j < d; ++j) out[(size_t)i * dp + j] = pts[(size_t)i * d + j];
  return out;
}

void GpDev::refresh_extent() {
  for (int k = 0; k < d; ++k) {
    double c = 0.0, ext = 0.0;
    for (int j = 0; j < n; ++j) c += X[(size_t)j * d + k];
    c /= (double)std::max(n, 1);
    for (int j = 0; j < n; ++j) ext = std::max(ext, std::fabs(X[(size_t)j * d + k] - c));
    x_mean[k] = c;
    x_ext[k] = ext;
  }
}

void GpDev::rebuild() {
  use_device();
  // MOE_BUILD_TRACE=1: host-clock split of a build on stderr (buffers | launches queued | device done | K^-1 y)
  static const bool trace = std::getenv("MOE_BUILD_TRACE") != nullptr && std::atoi(std::getenv("MOE_BUILD_TRACE")) != 0;
  const auto t0 = std::chrono::steady_clock::now();
  auto ms_since = [&](std::chrono::steady_clock::time_point t) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
  };
  N = n * (1 + g);
  refresh_extent();
  for (int k = 0; k < d; ++k) cp.center[k] = x_mean[k];  // frame centre of the value-only covariance builds: the training-set mean
  const std::vector<double> Xp = padded(X.data(), n);
  dX.upload(Xp.data(), Xp.size(), stream);
  dNoise.upload(noise.data(), noise.size(), stream);
  // head-room for rows appended later (add_points): a multiple of 16 doubles keeps columns 128-byte aligned
  ldL = ((long)N + kAppendHeadroom + 15) / 16 * 16;
  dL.reserve((size_t)ldL * ldL);
  dLinv.reserve((size_t)ldL * ldL);
  dInfo.reserve(1);
  dKinvY.reserve(N);
  dTmp.reserve((size_t)2 * N);
  // K(X, X) + noise, lower triangle only (r4: 8 [n d + N (N + 1) / 2] bytes, SURVEY 8(d)'s symmetric count -- the full square and the
  // pass that zeroed its upper half afterwards wrote three times that).  The strict upper triangle of dL is cleared once per buffer
  // SHAPE: neither this build nor the factorisation writes there, so rebuilds in place (hyper-parameter sampling) find it zero.
  const double t_buf = trace ? ms_since(t0) : 0.0;
  if (dL.p != zeroed_L || ldL != zeroed_ld) {
    MOE_HIP_CHECK(hipMemsetAsync(dL.p, 0, sizeof(double) * (size_t)ldL * ldL, stream));
    zeroed_L = dL.p;
    zeroed_ld = ldL;
  }
  launch_cov_build(cp, dX.p, n, derivs, dX.p, n, derivs, dNoise.p, dL.p, ldL, 0, stream, false, true);
  dWE.reserve(cholesky_work_doubles(N));  // the state workspace doubles as scratch of the recursive inversion
  // (r6, measured and NOT kept: the early inverse on a stream restricted to 3/4 of every XCD's CUs -- hipExtStreamCreateWithCUMask -- so
  //  that the trailing half's 64-column steps find a CU at once: 11.1 -> 10.5 ms at N = 8000 while the stream is pooled, but a masked
  //  hardware queue left alive doubles the time of every later burst of small kernels (a KG-MCMC suggestion 0.18 -> 0.37 s), and
  //  created / destroyed per build the second build of a process hung in the runtime.  `profiles/r06_g_chol_lookahead_ab.txt`.)
  if (side_stream == nullptr && early_inverse_split(N) > 0) side_stream = DevicePool::get().take_stream(device);
  launch_cholesky_and_inverse(N, dL.p, ldL, dLinv.p, ldL, dWE.p, dInfo.p, stream, true, side_stream);
  const double t_queued = trace ? ms_since(t0) : 0.0;
  if (trace) MOE_HIP_CHECK(hipStreamSynchronize(stream));
  const double t_dev = trace ? ms_since(t0) : 0.0;
  finish_factorisation();
  if (trace)
    std::fprintf(stderr, "[moe build] N=%d: buffers %.3f ms, launches queued at %.3f, device done at %.3f, K^-1 y done at %.3f\n", N, t_buf,
                 t_queued, t_dev, ms_since(t0));
}

// mean_ and K^-1 (y - mean_) from the inverse factor (two triangular GEMVs), queued behind the factorisation without a host round
// trip (r5: the centred data travel from a pinned buffer); the factorisation's status word is checked once everything has run.
void GpDev::finish_factorisation() {
  // mean_ = average of the function-value column only (gpp_math.cpp:498-504)
  mean = 0.0;
  for (int i = 0; i < n; ++i) mean += y[(size_t)i * (1 + g)];
  mean /= n;
  hYc.reserve((size_t)N + 1);
  for (int i = 0; i < N; ++i) hYc.p[i] = y[i];
  for (int i = 0; i < n; ++i) hYc.p[(size_t)i * (1 + g)] -= mean;
  MOE_HIP_CHECK(hipMemcpyAsync(dTmp.p, hYc.p, sizeof(double) * N, hipMemcpyHostToDevice, stream));
  launch_tri_gemm_skinny('N', N, 1, dLinv.p, ldL, dTmp.p, N, dTmp.p + N, N, stream);
  launch_tri_gemm_skinny('T', N, 1, dLinv.p, ldL, dTmp.p + N, N, dKinvY.p, N, stream);
  int* info = reinterpret_cast<int*>(hYc.p + N);
  *info = 0;
  MOE_HIP_CHECK(hipMemcpyAsync(info, dInfo.p, sizeof(int), hipMemcpyDeviceToHost, stream));
  MOE_HIP_CHECK(hipStreamSynchronize(stream));
  if (*info != 0)
    throw Error(MOE_ERR_SINGULAR,
                "Covariance matrix (K) singular. Check for duplicate points_sampled (with 0 noise) and/or extreme "
                "hyperparameter values.",
                N, *info);
}

void GpDev::mean_of_points(const double* pts, int k, double* mu, double* grad) {
  use_device();
  if (k <= 0) return;
  const bool want_grad = grad != nullptr;
  const int wdt = want_grad ? 1 + dp : 1;
  hStateIn.reserve((size_t)k * dp);
  for (int i = 0; i < k; ++i)
    for (int j = 0; j < dp; ++j) hStateIn.p[(size_t)i * dp + j] = (j < d) ? pts[(size_t)i * d + j] : 0.0;
  hStateOut.reserve((size_t)k * wdt);
  // r6, the latency path (compute_posterior_mean one candidate at a time): the kernel reads the few query points from the pinned
  // staging buffer and writes its results into the pinned result buffer -- pinned host memory is device-visible -- so the call is ONE
  // kernel and one wait instead of copy + kernel + copy (C1: 23.5 -> see profiles/r06_al_*).  Larger queries keep the copies: a
  // workgroup re-reads nothing of its point, but k x (1 + dp) doubles over PCIe in single stores stop paying.  MOE_GP_ZERO_COPY=0: copies.
  static const bool zero_copy_on = !(std::getenv("MOE_GP_ZERO_COPY") && std::getenv("MOE_GP_ZERO_COPY")[0] == '0');
  if (zero_copy_on && k <= 64 && Recorder::current() == nullptr) {
    launch_mean(cp, dX.p, n, derivs, dKinvY.p, hStateIn.p, k, mean, want_grad, hStateOut.p, stream);
  } else {
    dPts.upload(hStateIn.p, (size_t)k * dp, stream);
    dE.reserve((size_t)k * wdt);
    launch_mean(cp, dX.p, n, derivs, dKinvY.p, dPts.p, k, mean, want_grad, dE.p, stream);
    dE.download(hStateOut.p, (size_t)k * wdt, stream);
  }
  MOE_HIP_CHECK(hipStreamSynchronize(stream));
  for (int i = 0; i < k; ++i) {
    mu[i] = hStateOut.p[(size_t)i * wdt];
    if (want_grad)
      for (int j = 0; j < d; ++j) grad[(size_t)i * d + j] = hStateOut.p[(size_t)i * wdt + 1 + j];
  }
}

// On failure (a singular matrix with the new points) the handle is rolled back to the data it held before the call and
// refactorised, so it stays usable and consistent with the caller's own bookkeeping (the reference's AddPointsToGP leaves a
// half-updated object behind; its Python wrapper then holds an unusable GP).
void GpDev::add_points(const double* pts, const double* vals, int k) {
  if (k <= 0) return;
  const size_t x_keep = X.size(), y_keep = y.size();
  const int n_keep = n;
  try {
    add_points_unchecked(pts, vals, k);
  } catch (...) {
    X.resize(x_keep);
    y.resize(y_keep);
    n = n_keep;
    rebuild();  // the old data factorised before: this restores L, L^-1, K^-1 y, N and the device copy of X
    throw;
  }
}

void GpDev::add_points_unchecked(const double* pts, const double* vals, int k) {
  const int n0 = n, N0 = N, kk = k * (1 + g);
  X.insert(X.end(), pts, pts + (size_t)k * d);
  y.insert(y.end(), vals, vals + (size_t)k * (1 + g));
  n += k;
  // A few new points against an existing factorisation: append a block row to L and L^-1 (O(N^2 k)) instead of
  // rebuilding (O(N^3)).  MOE_GP_APPEND=0 forces the rebuild (A/B and tests).
  static const bool allow_append = [] {
    const char* e = std::getenv("MOE_GP_APPEND");
    return !(e && std::atoi(e) == 0);
  }();
  const int N1 = N0 + kk;
  if (!allow_append || n0 == 0 || N1 > ldL) {
    rebuild();
    return;
  }
  use_device();
  refresh_extent();  // (cp.center stays: any point near the data serves the value-only builds)
  bool singular = false;
  {
    // (dX is re-sent whole: a few KB, and its buffer may move when it grows)
    const std::vector<double> Xp = padded(X.data(), n);
    dX.upload(Xp.data(), Xp.size(), stream);
    const double* dNew = dX.p + (size_t)n0 * dp;
    dE.reserve((size_t)N0 * kk);
    dGram.reserve((size_t)kk * kk);
    dVE.reserve(cholesky_append_work_doubles(N0, kk));
    launch_cov_build(cp, dX.p, n0, derivs, dNew, k, derivs, nullptr, dE.p, N0, 0, stream);
    launch_cov_build(cp, dNew, k, derivs, dNew, k, derivs, dNoise.p, dGram.p, kk, 0, stream);
    launch_cholesky_append(N0, kk, dL.p, ldL, dLinv.p, ldL, dE.p, dGram.p, dVE.p, dInfo.p, stream);
    N = N1;
    dKinvY.reserve(N1);
    dTmp.reserve((size_t)2 * N1);
    try {
      finish_factorisation();
    } catch (const Error& e) {
      if (e.code != MOE_ERR_SINGULAR) throw;
      singular = true;
    }
  }
  // the Schur complement lost positive definiteness: report exactly as a full refactorisation does
  if (singular) rebuild();
}

namespace {
// Shared front half of the state set-ups: the padded points of all evaluations in ONE upload, then E = [K* | dK* | K(X, extra)] for the
// whole batch (columns grouped by kind, BatchLayout).  Returns the device pointer of the extra points.
const double* build_state_matrix(GpDev& gp, const double* U_all, int u, const DerivList& dt, int nd, const double* extra_all, int A,
                                 int E, StateLayout* lay_out, BatchLayout* bl_out, const StateAppendix* apx = nullptr) {
  hipStream_t s = gp.stream;
  StateLayout lay;
  lay.d = gp.d;
  lay.u = u;
  lay.gt = dt.g;
  lay.m = u * (1 + dt.g);
  lay.nd = nd;
  lay.A = A;
  const int ngrad = nd * (1 + dt.g) * gp.d;
  const int N = gp.N;
  BatchLayout bl;
  bl.E = E;
  bl.m = lay.m;
  bl.ngrad = ngrad;
  bl.A = A;
  const long ctot = bl.total();
  // padded point upload: all evaluations' union points | the differentiated subset (first nd of each) | the extras,
  // packed into ONE pinned staging buffer and ONE copy
  // (+ the caller's appendix -- records, normal draws -- so that a call's host operands go down in ONE copy, r4)
  const size_t nU = (size_t)E * u * gp.dp, nD = (size_t)E * nd * gp.dp, nX = (size_t)E * A * gp.dp;
  const size_t nApx = apx ? apx->doubles : 0;
  gp.hStateIn.reserve(nU + nD + nX + nApx);
  std::memset(gp.hStateIn.p, 0, sizeof(double) * (nU + nD + nX));
  // (r5: the extras right behind the union points -- without derivative observations the two value builds are one launch)
  double* Up = gp.hStateIn.p;
  double* Ep = Up + nU;
  double* Dp = Ep + nX;
  for (int e = 0; e < E; ++e) {
    for (int i = 0; i < u; ++i)
      for (int k = 0; k < gp.d; ++k) Up[((size_t)e * u + i) * gp.dp + k] = U_all[((size_t)e * u + i) * gp.d + k];
    for (int i = 0; i < nd; ++i)
      for (int k = 0; k < gp.d; ++k) Dp[((size_t)e * nd + i) * gp.dp + k] = U_all[((size_t)e * u + i) * gp.d + k];
    for (int j = 0; j < A; ++j)
      for (int k = 0; k < gp.d; ++k) Ep[((size_t)e * A + j) * gp.dp + k] = extra_all[((size_t)e * A + j) * gp.d + k];
  }
  if (apx) apx->fill(gp.hStateIn.p + nU + nD + nX);
  gp.dStateIn.upload(gp.hStateIn.p, nU + nD + nX + nApx, s, true);  // (hStateIn: pinned)
  double* dUp = gp.dStateIn.p;
  double* dEp = dUp + nU;
  double* dDp = dEp + nX;
  gp.dUnion = dUp;  // (kg.hip / ei.hip read the padded union points back from here)
  gp.dAppendix = dDp + nD;
  gp.dE.reserve((size_t)N * ctot);
  const bool pair = A > 0 && dt.g == 0 && gp.derivs.g == 0;
  if (pair)
    launch_cov_build_pair(gp.cp, gp.dX.p, gp.n, dUp, E * u, bl.col_kstar0(0), E * A, bl.col_extra0(0), gp.dE.p, N, s);
  else
    launch_cov_build(gp.cp, gp.dX.p, gp.n, gp.derivs, dUp, E * u, dt, nullptr, gp.dE.p, N, bl.col_kstar0(0), s);
  if (nd > 0) launch_grad_kstar(gp.cp, gp.dX.p, gp.n, gp.derivs, dDp, E * nd, dt, gp.dE.p, N, bl.col_grad0(0), s);
  if (A > 0 && !pair) {
    DerivList none;
    none.g = 0;
    for (int i = 0; i < kMaxDerivs; ++i) none.idx[i] = 0;
    launch_cov_build(gp.cp, gp.dX.p, gp.n, gp.derivs, dEp, E * A, none, nullptr, gp.dE.p, N, bl.col_extra0(0), s);
  }
  *lay_out = lay;
  *bl_out = bl;
  return dEp;
}
}  // namespace

StateEnqueued enqueue_state_batch(GpDev& gp, const double* U_all, int u, const DerivList& dt, int nd, const double* extra_all,
                                  int A, bool need_W, int num_evals, const StateAppendix* apx, bool consumer_forms_gram) {
  gp.use_device();
  hipStream_t s = gp.stream;
  const int E = num_evals;
  StateLayout lay;
  BatchLayout bl;
  build_state_matrix(gp, U_all, u, dt, nd, extra_all, A, E, &lay, &bl, apx);
  const int c = lay.c();
  const int ngrad = bl.ngrad;
  const int N = gp.N;
  const long ctot = bl.total();
  gp.dVE.reserve((size_t)N * ctot);
  const size_t nG = (size_t)c * c * E;
  gp.dGram.reserve(nG + ctot);  // gram matrices followed by ek: one download
  // (r4: a state of at most 16 columns per evaluation -- every q-EI call -- takes the skinny row-strip kernels for the triangular
  //  product, 18.6 -> ~6 us at C2, chosen from the per-evaluation column count so that a batch does not change an evaluation's bits;
  //  wider states keep the tiled kernels)
  if (c <= 16) {
    launch_tri_gemm_cols('N', N, (int)ctot, c, gp.dLinv.p, gp.ldL, gp.dE.p, N, gp.dVE.p, N, nullptr, s);
  } else {
    launch_tri_gemm('N', N, (int)ctot, gp.dLinv.p, gp.ldL, gp.dE.p, N, gp.dVE.p, N, s);
  }
  if (need_W) {
    const int cw = E * (lay.m + ngrad);
    gp.dWE.reserve((size_t)N * cw);
    launch_tri_gemm('T', N, cw, gp.dLinv.p, gp.ldL, gp.dVE.p, N, gp.dWE.p, N, s);
  }
  const bool fused = consumer_forms_gram && A == 0 && state_fits_lds(N, c);
  if (!fused) {
    gp.dEK.reserve((size_t)E * gram_batch_slices(E, c, N) * c * c);  // partial Grams of the K-sliced kernel
    launch_gram_batch(E, lay.m, ngrad, A, N, gp.dVE.p, N, gp.dGram.p, gp.dEK.p, s);
    launch_gemm_tn((int)ctot, 1, N, gp.dE.p, N, gp.dKinvY.p, N, gp.dGram.p + nG, (int)ctot, s);
  }
  StateEnqueued se;
  se.fused = fused;
  se.bl = bl;
  se.lay = lay;
  se.nG = nG;
  se.ctot = ctot;
  return se;
}

// The KG state (r4): L^-1 and L^-T are applied to the m columns of K* only --
//   V = L^-1 K*,  W = L^-T V = K^-1 K*,  gkk = V^T V,  gx = [dK* | K(X, discretised set)]^T W,  ek = E^T K^-1 (y - mean)
// -- which is how the reference forms the gradient of the variance (grad_K_star^T K_inv_times_K_star, gpp_math.cpp:1277-1290) and what
// its covariance with other points reduces to once K^-1 K* is at hand (gpp_math.cpp:815-821).  Until r3 both triangular products ran
// over all m + ngrad + A columns of the state matrix (474 per evaluation at C5, 32 of them K*: 2 x 3e10 flop per evaluation against 4e9).
KgStateEnqueued enqueue_kg_state_batch(GpDev& gp, const double* U_all, int u, int nd, const double* extra_all, int A, int num_evals,
                                       const StateAppendix* apx) {
  gp.use_device();
  hipStream_t s = gp.stream;
  const int E = num_evals;
  StateLayout lay;
  BatchLayout bl;
  const double* dEp = build_state_matrix(gp, U_all, u, gp.derivs, nd, extra_all, A, E, &lay, &bl, apx);
  const int m = lay.m, ng = bl.ngrad, R = ng + A, N = gp.N;
  const long ctot = bl.total();
  const long cm = (long)E * m;
  gp.dVE.reserve((size_t)N * cm * 2);  // (the second half: workspace of the gradient tail's K^-1 TB, kg.hip)
  gp.dWE.reserve((size_t)N * cm);
  // (one workspace for the split-K partials of the triangular products -- here and in the gradient tail -- and of the Gram kernels)
  // (the two Gram kernels' partials side by side: with m <= 8 they stay there for kg_state.hip to add up)
  const size_t part_kk = (size_t)E * gram_batch_slices(E, m, N) * m * m, part_x = (size_t)E * gram_cross_slices(m, ng, A, N) * R * m;
  gp.dEK.reserve(std::max(part_kk + part_x, tri_cols_work_doubles(N, (int)cm)));
  launch_tri_gemm_cols('N', N, (int)cm, m, gp.dLinv.p, gp.ldL, gp.dE.p + bl.col_kstar0(0) * N, N, gp.dVE.p, N, gp.dEK.p, s);
  launch_tri_gemm_cols('T', N, (int)cm, m, gp.dLinv.p, gp.ldL, gp.dVE.p, N, gp.dWE.p, N, gp.dEK.p, s);
  const size_t n_kk = (size_t)E * m * m, n_x = (size_t)E * R * m;
  gp.dGram.reserve(n_kk + n_x + ctot);
  static const bool defer_env = !(std::getenv("MOE_KG_GRAM_IN_STATE") != nullptr && std::atoi(std::getenv("MOE_KG_GRAM_IN_STATE")) == 0);
  const bool defer = defer_env && m <= 8;  // (the latency path: q-KG with a handful of points)
  const int s_kk = launch_gram_batch(E, m, 0, 0, N, gp.dVE.p, N, gp.dGram.p, gp.dEK.p, s, defer);
  const int s_x = launch_gram_cross_batch(E, m, ng, A, N, gp.dE.p, N, gp.dWE.p, N, gp.dGram.p + n_kk, gp.dEK.p + part_kk, s, defer);
  launch_gemm_tn((int)ctot, 1, N, gp.dE.p, N, gp.dKinvY.p, N, gp.dGram.p + n_kk + n_x, (int)ctot, s);
  KgStateEnqueued se;
  se.bl = bl;
  se.gkk = (defer && s_kk > 1) ? gp.dEK.p : gp.dGram.p;
  se.gx = (defer && s_x > 1) ? gp.dEK.p + part_kk : gp.dGram.p + n_kk;
  se.gkk_slices = defer ? s_kk : 1;
  se.gx_slices = defer ? s_x : 1;
  se.ek = gp.dGram.p + n_kk + n_x;
  se.U = gp.dUnion;
  se.extra = dEp;
  return se;
}

namespace {
// var = Kss - gram in place (Kss in `var`), col-major m x m
__global__ __launch_bounds__(256) void var_sub_kernel(double* __restrict__ var, const double* __restrict__ gram, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) var[i] = var[i] - gram[i];
}
// out = lower triangle (diagonal included) of `chol`, strict upper triangle of `var`
__global__ __launch_bounds__(256) void chol_merge_kernel(double* __restrict__ out, const double* __restrict__ chol,
                                                         const double* __restrict__ var, int m) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < (long)m * m) {
    const int row = (int)(i % m), col = (int)(i / m);
    out[i] = (row >= col) ? chol[i] : var[i];
  }
}
}  // namespace

// ComputeVarianceOfPoints (gpp_math.cpp:924-970) and its Cholesky factor (ComputeCholeskyFactorL, gpp_linear_algebra.cpp:109-148, pivot
// rule 1e-16) with the m x m algebra ON THE DEVICE (r5): Kss by the covariance-assembly kernel on the query points
// themselves, the subtraction of the Gram matrix, the factorisation by the GP's own blocked kernels -- for query sets of hundreds of
// points, where the host algebra of host_math.hip costs O(m^2 d) covariance calls and O(m^3) scalar operations.  out [m x m] col-major:
// the variance, or (cholesky) its factor in the lower triangle with the variance's entries left above the diagonal, as the host path
// leaves them (the reference's ZeroUpperTriangle call is the Python boundary's, GPP.py).
void variance_on_device(GpDev& gp, const double* pts, int k, bool cholesky, double* out) {
  gp.use_device();
  hipStream_t s = gp.stream;
  const StateEnqueued se = enqueue_state_batch(gp, pts, k, gp.derivs, 0, nullptr, 0, false, 1);
  const int m = se.lay.m;
  const size_t mm = (size_t)m * m;
  const size_t work = cholesky ? cholesky_work_doubles(m) : 0;
  // [var | factor workspace | inverse factor (unused by the caller) | scratch | merged output]
  gp.dVarWork.reserve(mm * (cholesky ? 4 : 1) + work);
  double* dVar = gp.dVarWork.p;
  launch_cov_build(gp.cp, gp.dUnion, k, gp.derivs, gp.dUnion, k, gp.derivs, nullptr, dVar, m, 0, s);
  MOE_LAUNCH_NOW(var_sub_kernel, dim3((unsigned)((mm + 255) / 256)), dim3(256), 0, s, dVar, gp.dGram.p, (long)mm);
  MOE_HIP_CHECK(hipGetLastError());
  gp.hStateOut.reserve(mm + 1);
  const bool factor_on_device = cholesky && m >= device_variance_min_m(true);
  if (!factor_on_device) {
    MOE_HIP_CHECK(hipMemcpyAsync(gp.hStateOut.p, dVar, sizeof(double) * mm, hipMemcpyDeviceToHost, s));
    MOE_HIP_CHECK(hipStreamSynchronize(s));
    std::copy(gp.hStateOut.p, gp.hStateOut.p + mm, out);
    if (cholesky) {  // r6: the variance from the device, its factor by the host's column sweep (a few us at these sizes; the blocked device
                     // factorisation costs 70 us before it does anything): the layout the callers know -- factor below, variance above
      const int lm = host_cholesky(m, out);
      if (lm != 0)
        throw Error(MOE_ERR_SINGULAR,
                    "GP-Variance matrix singular. Check for duplicate points_to_sample or points_to_sample "
                    "duplicating points_sampled with 0 noise.",
                    m, lm);
    }
    return;
  }
  double* dChol = dVar + mm;
  double* dInv = dChol + mm;
  double* dOut = dInv + mm;
  double* dWork = dOut + mm;
  MOE_HIP_CHECK(hipMemcpyAsync(dChol, dVar, sizeof(double) * mm, hipMemcpyDeviceToDevice, s));
  gp.dInfo.reserve(1);
  launch_cholesky_and_inverse(m, dChol, m, dInv, m, dWork, gp.dInfo.p, s);
  MOE_LAUNCH(chol_merge_kernel, dim3((unsigned)((mm + 255) / 256)), dim3(256), 0, s, dOut, dChol, dVar, m);
  MOE_HIP_CHECK(hipGetLastError());
  int info = 0;
  gp.dInfo.download(&info, 1, s);
  MOE_HIP_CHECK(hipMemcpyAsync(gp.hStateOut.p, dOut, sizeof(double) * mm, hipMemcpyDeviceToHost, s));
  MOE_HIP_CHECK(hipStreamSynchronize(s));
  if (info != 0)
    throw Error(MOE_ERR_SINGULAR,
                "GP-Variance matrix singular. Check for duplicate points_to_sample or points_to_sample "
                "duplicating points_sampled with 0 noise.",
                m, info);
  std::copy(gp.hStateOut.p, gp.hStateOut.p + mm, out);
}

void compute_state_batch(GpDev& gp, const double* U_all, int u, const DerivList& dt, int nd, const double* extra_all, int A,
                         bool need_W, int num_evals, BatchLayout* blay, std::vector<StateHost>* hosts) {
  const StateEnqueued se = enqueue_state_batch(gp, U_all, u, dt, nd, extra_all, A, need_W, num_evals);
  hipStream_t s = gp.stream;
  const int E = num_evals;
  const BatchLayout& bl = se.bl;
  const StateLayout& lay = se.lay;
  const int c = lay.c();
  const int ngrad = nd * (1 + dt.g) * gp.d;
  const size_t nG = se.nG;
  const long ctot = se.ctot;
  gp.hStateOut.reserve(nG + ctot);
  gp.dGram.download(gp.hStateOut.p, nG + ctot, s);
  MOE_HIP_CHECK(hipStreamSynchronize(s));
  const double* gram_all = gp.hStateOut.p;
  const double* ek_all = gp.hStateOut.p + nG;
  hosts->resize(E);
  for (int e = 0; e < E; ++e) {
    StateHost& h = (*hosts)[e];
    h.lay = lay;
    h.cp = gp.cp;
    h.dt = dt;
    h.U.assign(U_all + (size_t)e * u * gp.d, U_all + (size_t)(e + 1) * u * gp.d);
    if (A > 0)
      h.extra.assign(extra_all + (size_t)e * A * gp.d, extra_all + (size_t)(e + 1) * A * gp.d);
    else
      h.extra.clear();
    h.gram.assign(gram_all + (size_t)e * c * c, gram_all + (size_t)(e + 1) * c * c);
    h.ek.resize(c);
    for (int l = 0; l < lay.m; ++l) h.ek[l] = ek_all[bl.col_kstar0(e) + l];
    for (int l = 0; l < ngrad; ++l) h.ek[lay.m + l] = ek_all[bl.col_grad0(e) + l];
    for (int l = 0; l < A; ++l) h.ek[lay.m + ngrad + l] = ek_all[bl.col_extra0(e) + l];
    h.mean = gp.mean;
  }
  if (blay) *blay = bl;
}

void compute_state(GpDev& gp, const double* U, int u, const DerivList& dt, int nd, const double* extra, int A, bool need_W,
                   StateDev* dev, StateHost* host) {
  std::vector<StateHost> hosts;
  compute_state_batch(gp, U, u, dt, nd, extra, A, need_W, 1, nullptr, &hosts);
  *host = std::move(hosts[0]);
  if (dev) {
    dev->lay = host->lay;
    dev->have_W = need_W;
  }
}

}  // namespace moe
