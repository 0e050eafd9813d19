// cornell_moe_amd/csrc/gp.hpp -- device-resident Gaussian process (the object behind moe_gp_t) and the per-call
// "points state" set-up shared by the posterior queries, q-EI and q-KG.
#pragma once
#include <cstdlib>
#include <functional>
#include <memory>
#include <vector>

#include "common.hpp"
#include "host_math.hpp"
#include "kernels.hpp"

namespace moe {

// Validated covariance parameters from [alpha, lengths[d]] (throws MOE_ERR_BOUNDS on non-positive entries).
void fill_cov_params(CovParams& cp, int cov_type, int d, const double* hyper);

// HBM-resident state of one GP (replaces GaussianProcess' K_chol_/K_inv_y_ members, gpp_math.hpp:840-868):
//   dX     [n][DP]   padded training points
//   dL     [N][N]    Cholesky factor of K + noise (lower; strict upper zero), column-major with leading dimension ldL
//   dLinv  [N][N]    its explicit inverse (lower), same layout
//   dKinvY [N]       K^-1 (y - mean)
struct GpDev {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t side_stream = nullptr;  // early-inverse schedule of the build (kernels.hpp: launch_cholesky_and_inverse); taken on first need
  void release_side_streams();
  int d = 0, dp = 0, n = 0, g = 0, N = 0;
  long ldL = 0;  // leading dimension of dL / dLinv: N at the last rebuild + head-room for appended rows
  const double* zeroed_L = nullptr;  // the buffer / leading dimension dL's strict upper triangle was last cleared for (rebuild)
  long zeroed_ld = 0;
  CovParams cp;
  DerivList derivs;
  std::vector<double> X, y, noise;  // host copies (reference keeps them too, gpp_math.hpp:846-856)
  double mean = 0.0;
  DevBuf<double> dX, dL, dLinv, dKinvY, dNoise, dTmp;
  DevBuf<double> dVarWork;  // variance_on_device (dTmp keeps y - mean for the log likelihood)
  DevBuf<int> dInfo;
  // reusable workspaces for states
  DevBuf<double> dPts, dPtsGrad, dExtra, dE, dVE, dWE, dGram, dEK, dStateIn;
  // inside dStateIn after a state set-up: the padded union points of the batch [E][u][dp], and the caller's appendix (StateAppendix)
  const double* dUnion = nullptr;
  const double* dAppendix = nullptr;
  PinnedBuf<double> hStateIn, hStateOut, hKgIn, hKgOut;  // pinned staging for the per-call operands / results
  // the argument tables of ensemble-wide launches (mcmc.hip: replay_ensemble; held by the ensemble's first member)
  PinnedBuf<unsigned char> hEns;
  DevBuf<unsigned char> dEns;
  PinnedBuf<double> hYc;  // y - mean on its way to the device, and the factorisation's status word on its way back
  // reusable workspaces of the KG evaluator (kg.hip)
  DevBuf<double> kBlob, kNormals, kTab, kBestPoint, kBestValue, kBeta, kT, kC, kTB, kOut, kSW, kSWpart, kZcPart, kV;
  DevBuf<unsigned long long> kCounters;
  DevBuf<unsigned int> kEiTicket;  // arrival counters of ei_mc_kernel's fused final sum (ei.hip)
  bool ei_ticket_dirty = false;  // a launch of ei_mc_kernel was enqueued and its completion not yet seen: an aborted launch leaves
                                  // the arrival counters non-zero, so the next call clears them first
  DevBuf<int> kBestJ, kStateI;   // kStateI: singular flags | winners of a KG batch (kg_state.hip)
  DevBuf<double> kStateD;        // grad mu | d chol / d Xq (packed) | final [kg_sum | grad] per evaluation
  int num_cu = 256;
  // per-dimension mean and max |x - mean| of the training points (refreshed by rebuild() and by the append path of add_points):
  // the frame centre of the KG coordinate tables and the extent the kernel selection needs, so that an evaluation does not
  // re-walk the n x d coordinates on the host (r3, ADVICE)
  double x_mean[kMaxDimPadded] = {0}, x_ext[kMaxDimPadded] = {0};
  void refresh_extent();
  // timing of the last KG call (ms): mc, cov-build, tail contraction, state, total
  double last_ms[5] = {0, 0, 0, 0, 0};
  // which MC kernel the last KG launch took (moe_last_kernel_info): variant (0 wave-per-sample, 1 workgroup-per-sample) |
  // coordinate table in LDS | wavefronts per workgroup | register tiles per wavefront (variant 1) | streamed per-sample weight
  // table | T-free gradient tail | workgroups | sample pre-pass
  int last_info[8] = {0, 0, 0, 0, 0, 0, 0, 0};

  GpDev(const double* hyper, int cov_type, const double* X_in, const double* y_in, const double* noise_in,
        const int* derivs_in, int g_in, int d_in, int n_in, int device_in);
  ~GpDev();
  GpDev(const GpDev&) = delete;
  GpDev& operator=(const GpDev&) = delete;

  void use_device() const;
  void set_covariance(const double* hyper);  // alpha, lengths -> cp (validated)
  void rebuild();  // K assembly + Cholesky + inverse + K^-1 (y - mean)  (RecomputeDerivedVariables, gpp_math.cpp:481-511)
  // Appends k observations: a rank-k(1+g) block-row append to L and L^-1 while the head-room lasts (launch_cholesky_append),
  // the full rebuild otherwise.  (AddPointsToGP, gpp_math.cpp:1699-1737, always refactorises.)
  void add_points(const double* pts, const double* vals, int k);
  void add_points_unchecked(const double* pts, const double* vals, int k);
  void finish_factorisation();
  // New covariance hyper-parameters [alpha, lengths...] and noise [1 + g] on the same data: rebuild in place (buffers and
  // stream are kept) -- the inner step of hyper-parameter sampling (LogMarginalLikelihoodState::SetHyperparameters,
  // gpp_model_selection.cpp:798-811).
  void set_hyperparameters(const double* hyper, const double* noise_in);
  // log p(y | X, theta) of the current factorisation = -1/2 yc^T K^-1 yc - sum log L_ii - N/2 log 2 pi
  // (LogMarginalLikelihoodEvaluator::ComputeLogLikelihood, gpp_model_selection.cpp:593-612).
  double log_marginal_likelihood();
  // d log p / d [alpha, lengths[d], noise variances[1 + g]] of the current factorisation
  // (LogMarginalLikelihoodEvaluator::ComputeGradLogLikelihood, gpp_model_selection.cpp:629-677; with the reference's
  // Matern-5/2 convention that only the function-value block depends on the covariance hyper-parameters,
  // gpp_covariance.cpp:461-487).
  void grad_log_marginal_likelihood(double* grad);
  std::vector<double> padded(const double* pts, int k) const;  // [k][d] -> [k][DP]
  // Posterior mean of the function value at k points (one kernel, one copy each way): mu[k]; grad (may be NULL) [k][d].
  void mean_of_points(const double* pts, int k, double* mu, double* grad);
};

// Device-side product of a state set-up.  Columns of E (ld = N): see StateLayout.
struct StateDev {
  StateLayout lay;
  bool have_W = false;  // dWE = K^-1 E was computed (needed by the KG kernels)
};

// Builds E = [K*(X,U) | dK*/dU (first nd points) | K(X, extra)], VE = L^-1 E, optionally WE = L^-T VE, then
// gram = VE^T VE and ek = E^T K^-1(y-mean); downloads gram/ek into `host` (synchronises the GP's stream).
void compute_state(GpDev& gp, const double* U, int u, const DerivList& dt, int nd, const double* extra, int A, bool need_W,
                   StateDev* dev, StateHost* host);

// The same for `num_evals` independent point sets in one pass (the multistart axis): U_all[e][u][d], extra_all[e][A][d].
// Columns of E / VE / WE (ld = N) are grouped by kind so that every group is one launch:
//   [ K* of eval 0 | K* of eval 1 | ... | dK* of eval 0 | ... | extra of eval 0 | ... ]
// i.e. W_e = WE + col_kstar0(e) * N (m columns) and K^-1 dK*_e = WE + col_grad0(e) * N (ngrad columns).
struct BatchLayout {
  int E = 1, m = 0, ngrad = 0, A = 0;
  long col_kstar0(int e) const { return (long)e * m; }
  long col_grad0(int e) const { return (long)E * m + (long)e * ngrad; }
  long col_extra0(int e) const { return (long)E * (m + ngrad) + (long)e * A; }
  long total() const { return (long)E * (m + ngrad + A); }
};
// The device half of compute_state_batch -- uploads and kernels on gp.stream, NO sync: Gram matrices [E][c][c] followed by ek [ctot]
// are left in gp.dGram, the padded union points at gp.dUnion.  For callers that keep going on the device (ei.hip).
// Host operands of the CALLER that ride along in the state set-up's single host->device copy (r4: a q-EI evaluation made five
// copies of ~3 us each, a q-KG one seven): `doubles` doubles written by fill() into the pinned staging buffer behind the points;
// their device address afterwards is GpDev::dAppendix.
struct StateAppendix {
  size_t doubles = 0;
  std::function<void(double*)> fill;
};
struct StateEnqueued {
  BatchLayout bl;
  StateLayout lay;
  size_t nG = 0;   // doubles of Gram matrices in front of ek
  long ctot = 0;
  bool fused = false;  // small state, caller asked for it: neither the Gram matrices nor ek were formed -- V = L^-1 E (gp.dVE) and E
                       // (gp.dE) are left for a consumer that forms them itself (ei.hip: ei_state_kernel)
};
// whether a state of c columns over N rows is "small": its V and E columns fit the LDS of the workgroup that consumes them
inline bool state_fits_lds(int N, int c) { return c <= 48 && sizeof(double) * ((size_t)c * c + c + N + 2 * (size_t)c * N) <= 144 * 1024; }
StateEnqueued enqueue_state_batch(GpDev& gp, const double* U_all, int u, const DerivList& dt, int nd, const double* extra_all,
                                  int A, bool need_W, int num_evals, const StateAppendix* apx = nullptr,
                                  bool consumer_forms_gram = false);
// The KG evaluator's state set-up (r4), everything left on the device for kg_state.hip: see gp.hip.
struct KgStateEnqueued {
  BatchLayout bl;
  const double* gkk = nullptr;    // [E][m x m]              -- or, gkk_slices > 1: [E][gkk_slices][m x m] partial sums over K slices that
  const double* gx = nullptr;     // [E][(ngrad + A) x m]    --     the consumer adds up in slice order (r5, m <= 8: two launches less)
  int gkk_slices = 1, gx_slices = 1;
  const double* ek = nullptr;     // [E m | E ngrad | E A]
  const double* U = nullptr;      // [E][u][dp]
  const double* extra = nullptr;  // [E][A][dp]
};
KgStateEnqueued enqueue_kg_state_batch(GpDev& gp, const double* U_all, int u, int nd, const double* extra_all, int A, int num_evals,
                                       const StateAppendix* apx = nullptr);
void compute_state_batch(GpDev& gp, const double* U_all, int u, const DerivList& dt, int nd, const double* extra_all, int A,
                         bool need_W, int num_evals, BatchLayout* blay, std::vector<StateHost>* hosts);
// Variance (or its Cholesky factor) of k query points with the m x m algebra on the device: gp.hip.  For large query sets; small ones
// keep the host algebra.  Two thresholds in rows of the m x m variance: from kDeviceVarianceMinMPlain the variance itself is assembled
// on the device (both endpoints), from kDeviceVarianceMinM its factor too (below that the host's column sweep factors the downloaded
// variance).
constexpr int kDeviceVarianceMinM = 224;      // the factorisation on the device (r6: was 33; the blocked kernels cost 70 us before they do anything --
                                              // host sweep 287 us at 192 rows against 357, 536 at 256 against 277: profiles/r06_an_cholesky_variance_threshold.txt)
constexpr int kDeviceVarianceMinMPlain = 16;  // the variance on the device (r6: measured break-even, profiles/r06_am_variance_threshold.txt: at 32 rows 64 against 96 us)
// (MOE_GP_VARIANCE_DEVICE_MIN / MOE_GP_CHOL_VARIANCE_DEVICE_MIN: A/B runs of the thresholds, read per call; 1 = the device at every size)
inline int device_variance_min_m(bool cholesky) {
  const char* v = std::getenv(cholesky ? "MOE_GP_CHOL_VARIANCE_DEVICE_MIN" : "MOE_GP_VARIANCE_DEVICE_MIN");
  return (v && *v) ? std::atoi(v) : (cholesky ? kDeviceVarianceMinM : kDeviceVarianceMinMPlain);
}
void variance_on_device(GpDev& gp, const double* pts, int k, bool cholesky, double* out);
// r6 (query_grad.hip): ComputeGradVarianceOfPoints / ComputeGradCholeskyVarianceOfPoints (gpp_math.cpp:1267-1474) for the first
// `num_derivs` of the `num_pts` points, the m x m x d algebra on the device: out[num_derivs][d m m] in the reference's layout.
void grad_variance_on_device(GpDev& gp, const double* pts, int num_pts, int num_derivs, bool cholesky, double* out);

}  // namespace moe
