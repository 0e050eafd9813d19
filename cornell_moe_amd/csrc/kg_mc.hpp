// cornell_moe_amd/csrc/kg_mc.hpp -- the q-KG / d-KG Monte-Carlo inner-optimisation kernel (gfx950), shared by the
// per-dimension translation units kg_mc_dp*.hip (one instantiation set per padded dimension, compiled in parallel).
//
// What the reference does per MC sample i (gpp_knowledge_gradient_optimization.cpp:170-196): draw z_i, set the fantasy
// observations y_i = mu(Xu) + L z_i, RE-SOLVE K_after^-1 (y - mean) with two O((N+m)^2) triangular sweeps
// (gpp_math.cpp:531-551), then maximise -mu_after,i(x) from the best discretised start with a back-tracking
// line-search gradient descent (.cpp:420-472, gpp_optimization.hpp:708-828).
//
// What this kernel does instead -- same mathematics, no N^2 work per sample:
//   K_after^-1 (y_i - mean) = [ K^-1(y - mean) - W beta_i ; beta_i ],   W = K^-1 K*(X,Xu),  beta_i = L^-T z_i,
// (block elimination of the (N+m) system; the reference's own gradient tail relies on the same identity, .cpp:199-209),
// so  mu_after,i(x) = mean + sum over the n + u points p of  [ w_p0 base(x,p) + first(x,p) sum_a w_pa (p - x)_{d_a} ]
// with a per-sample weight block w (1 + g values per point: the function-value weight and one weight per observed
// partial derivative) that costs N*m flops to form.
//
// Mapping: ONE WAVEFRONT owns one MC sample at a time and pulls the next sample of its evaluation from an atomic
// counter when it finishes (persistent waves: no workgroup-level tail while a slow line search finishes).  Lanes stride
// over the n + u points; point coordinates are staged once per workgroup in LDS, tile-major [tile][dim][64 lanes] so that
// every ds_read_b64 is conflict free and its address is base + immediate; the wave's weights live in its own LDS slab with
// the same tiling; each posterior-mean (or mean + gradient) evaluation ends in a 64-lane butterfly reduction.  Control
// flow of the line search is wave-uniform: there is no intra-wave divergence.
#pragma once
#include <hip/hip_runtime.h>

#include "fastmath.hpp"
#include "kernels.hpp"

namespace moe {

// Phase timing of the workgroup-per-sample kernel (s_memtime cycles summed over samples by wave 0 into counters[...]):
// build with -DMOE_BLOCK_PROF=1 (tools only; off in the product build).
#ifndef MOE_BLOCK_PROF
#define MOE_BLOCK_PROF 0
#endif
// Gradient passes with dot-product distances (eval_loop GDOT): 5 fewer instructions per point, but the gradient then carries the
// cancellation of sum coef x - q sum coef; where the inner optimiser is run to convergence (100 steps x 10 restarts, the
// reference's own ping-test settings) the end points drift to 1.4e-6 from the reference's instead of 1e-7.  Off.
#ifndef MOE_KG_GRAD_DOT
#define MOE_KG_GRAD_DOT 0
#endif
// radial3's square root: 1 = seed + one Heron step (<= 36 ulp, two instructions less per covariance entry), 0 = correctly rounded
#ifndef MOE_KG_FAST_SQRT
#define MOE_KG_FAST_SQRT 1
#endif
#if MOE_BLOCK_PROF
#define MOE_PROF_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define MOE_PROF_ADD(dst, a, b) dst += (b) - (a)
#else
#define MOE_PROF_T(var)
#define MOE_PROF_ADD(dst, a, b)
#endif
constexpr int kTicketStride = 32;  // unsigned ints between the sample-ticket counters of consecutive evaluations (128 B)
constexpr int kMaxM = 64;   // m = (q + p)(1 + g) limit of the wave-per-sample kernel and of every "one lane per component" routine
constexpr int kMaxMB = 128;  // limit of the workgroup-per-sample kernel (r2: q = 8 with all 12 derivatives of C5 observed is
                             // m = 104): beta of the current sample lives at zb[kMaxMB ...), filled from the sample pre-pass
constexpr int kExpTabLen = 64;  // 2^(j/64) table at the start of the MC kernel's LDS (fastmath.hpp exp_nonpos_tab)

struct KgRec {  // offsets (doubles) of one evaluation's small operands inside the blob; identical for every evaluation
  int L;        // [m x m] col-major lower Cholesky factor of Var(Xu) + noise
  int mu_disc;  // [A]      mu_n at the discretised points
  int C_disc;   // [A][m]   L^-1 cov_n(Xu, x_j)
  int disc;     // [A][size] discretised points (unscaled, fidelity dims dropped)
  int XuP;      // [u][dp]  union points, padded (unscaled)
  int Mk;       // unused by the MC kernel
  int stride;   // record length
};

struct KgMcParams {
  int cov_type, dim;
  double alpha;
  double center[kMaxDimPadded];  // training-set mean of table row r's coordinate: the tables and the queries are centred on it
                                 // BEFORE scaling ((x - c) / l is exact to rounding wherever the domain sits)
  double inv_lp[kMaxDimPadded];  // frame scale of table row r (row r holds original dimension perm[r]): 1 / length for the squared
                                 // exponential, sqrt(5) / length for Matern-5/2 (radial3 then needs no sqrt(5)); 0 in pad rows
  int perm[kMaxDimPadded];       // the GP's observed-derivative dimensions come first: perm[a] = derivatives[a], a < g
  int inv_perm[kMaxDimPadded];   // table row of original dimension j (the simplex update walks the coordinates in the reference's order)
  int simplex;                   // inner domain: 0 = tensor product, 1 = its intersection with the unit simplex (line_search_lds only:
                                 // kg_launch keeps such evaluations off the LDS-slab wave-per-sample kernel)
  double inv_sqrt_size;          // 1 / sqrt(dim - f): the components of the diagonal face's unit normal
  int n, g, N, u, m, f, A, ntiles, E;
  int multi_trial;  // 0: one Armijo trial per pass only (point sets spanning > 100 length scales: the projected distances of the
                    // multi-trial passes lose absolute accuracy with the square of the trial's offset)
  double mean;
  const double* XsTab;  // [E][ntiles][DP][64] scaled coordinates of X then Xu_e, zero padded
                        // (pair_rows: rows in pairs, [E][ntiles][DP / 2][64][2] -- WideEval: the streamed wave-per-sample kernel and d > 16)
  int wide_lds_tiles;   // streamed wave-per-sample kernel: leading tiles of the table copied to LDS per workgroup
  long tab_stride;
  const double* KinvY;  // [N], entry (j, a) at j (1 + g) + a
  const double* W;      // K^-1 K*: evaluation e at W + e * w_stride, [N x m], ld N
  long w_stride;
  const double* blob;
  KgRec rec;
  const double* bounds;   // [2 kMaxDimPadded] domain bounds in TABLE-ROW order (row r = original dimension perm[r])
  unsigned int free_mask; // bit r set: table row r is an optimised coordinate (a real, non-fidelity dimension)
  const double* normals;  // [ceil(M/2)][m]
  int first_sample, num_local;
  int max_num_steps, max_num_restarts;
  double gamma, pre_mult, max_relative_change, tolerance;
  double* best_point;  // [E][num_local][DP] (unscaled, fidelity coords = 1, pads = 0)
  double* best_value;  // [E][num_local]
  double* beta;        // [E][num_local][m]
  unsigned long long* counters;  // [E][2]: value passes, value + gradient passes
  unsigned int* next_sample;     // [E] work counters (zeroed before launch)
  unsigned long long* prof;      // 16 spare words behind the counters (MOE_BLOCK_PROF builds)
  const double* V;               // [E][num_local][v_stride] per-sample weights alpha-scaled (kg_sample_weights_kernel, kg.hip), or NULL
  long v_stride;                 // N (workgroup-per-sample kernel) or ntiles * 64 * v_slots1: the fantasy points' weights and the zero
                                 // padding included (streamed-weights kernel, kg_mc_stream_kernel)
  int v_slots1;                  // weights per point in the table: 1 + g, or 1 + G of the streamed-weights instantiation when it has
                                 // more derivative slots than the GP observes (g = 5 .. 7 -> 8, 9 .. 11 -> 12: the extra slots hold 0)
  const int* best_j;             // [E][num_local] start point of every sample's line search, precomputed with beta by
                                 // kg_sample_prep_kernel (kg.hip); NULL = each sample computes them itself
};

// Launchers (one translation unit per padded dimension).  `waves` = wavefronts per workgroup, `shm` = dynamic LDS bytes.
void launch_kg_mc_dp4(const KgMcParams& P, int G, bool xlds, int blocks, int waves, size_t shm, hipStream_t s);
void launch_kg_mc_dp8(const KgMcParams& P, int G, bool xlds, int blocks, int waves, size_t shm, hipStream_t s);
void launch_kg_mc_dp12(const KgMcParams& P, int G, bool xlds, int blocks, int waves, size_t shm, hipStream_t s);
void launch_kg_mc_dp16(const KgMcParams& P, int G, bool xlds, int blocks, int waves, size_t shm, hipStream_t s);
// padded dimensions 24 and 32 (d = 17 .. 32) are built for a reduced set: derivative slots {0, 4} (wave-per-sample kernel,
// coordinates streamed from L2 only) and {0, 4, 8, 12} with all tiles in LDS (workgroup-per-sample kernel)
void launch_kg_mc_dp24(const KgMcParams& P, int G, bool xlds, int blocks, int waves, size_t shm, hipStream_t s);
void launch_kg_mc_dp32(const KgMcParams& P, int G, bool xlds, int blocks, int waves, size_t shm, hipStream_t s);

// Lane-parked wave-per-sample kernel (r5, kg_mc_lane.hpp): the LDS-table kernel above with the line search's vectors one row per lane and
// the evaluation's record head ([L | mu_disc | C_disc | disc]: `rec_head` doubles) in LDS; 8 wavefronts, padded dimension <= 16.
void launch_kg_mc_lane_dp4(const KgMcParams& P, int G, int rec_head, int blocks, int waves, size_t shm, hipStream_t s);
void launch_kg_mc_lane_dp8(const KgMcParams& P, int G, int rec_head, int blocks, int waves, size_t shm, hipStream_t s);
void launch_kg_mc_lane_dp12(const KgMcParams& P, int G, int rec_head, int blocks, int waves, size_t shm, hipStream_t s);
void launch_kg_mc_lane_dp16(const KgMcParams& P, int G, int rec_head, int blocks, int waves, size_t shm, hipStream_t s);
size_t kg_mc_lane_fixed_bytes(int dp, int rec_head);  // LDS in front of the coordinate table

// Streamed-weights wave-per-sample kernel (kg_mc_stream_kernel): weights from the table P.V, P.wide_lds_tiles tiles of coordinates in LDS.
void launch_kg_mc_stream_dp4(const KgMcParams& P, int G, int blocks, int waves, size_t shm, hipStream_t s);
void launch_kg_mc_stream_dp8(const KgMcParams& P, int G, int blocks, int waves, size_t shm, hipStream_t s);
void launch_kg_mc_stream_dp12(const KgMcParams& P, int G, int blocks, int waves, size_t shm, hipStream_t s);
void launch_kg_mc_stream_dp16(const KgMcParams& P, int G, int blocks, int waves, size_t shm, hipStream_t s);
void launch_kg_mc_stream_dp24(const KgMcParams& P, int G, int blocks, int waves, size_t shm, hipStream_t s);
void launch_kg_mc_stream_dp32(const KgMcParams& P, int G, int blocks, int waves, size_t shm, hipStream_t s);

// Workgroup-per-sample variant: the first `num_lds_tiles` tiles of 64 points in LDS, `tr` (0, 2 or 4) register tiles per
// wavefront for the rest; bytes of dynamic LDS = kg_mc_block_lds_bytes(...).
void launch_kg_mc_block_dp4(const KgMcParams& P, int G, int tr, int num_lds_tiles, int blocks, int waves, hipStream_t s);
void launch_kg_mc_block_dp8(const KgMcParams& P, int G, int tr, int num_lds_tiles, int blocks, int waves, hipStream_t s);
void launch_kg_mc_block_dp12(const KgMcParams& P, int G, int tr, int num_lds_tiles, int blocks, int waves, hipStream_t s);
void launch_kg_mc_block_dp16(const KgMcParams& P, int G, int tr, int num_lds_tiles, int blocks, int waves, hipStream_t s);
void launch_kg_mc_block_dp24(const KgMcParams& P, int G, int tr, int num_lds_tiles, int blocks, int waves, hipStream_t s);
void launch_kg_mc_block_dp32(const KgMcParams& P, int G, int tr, int num_lds_tiles, int blocks, int waves, hipStream_t s);
size_t kg_mc_block_lds_bytes(int dp, int G, int num_lds_tiles);

#if defined(__HIPCC__)
namespace mc {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// One DPP data movement of a double (two 32-bit v_mov_b32 dpp).  FULL = every lane has a valid source (the in-row
// permutations): the destination's previous contents are irrelevant, so the source itself is passed as `old` and no
// zero-fill is emitted.  Otherwise lanes masked off by ROW_MASK read 0.
template <int CTRL, int ROW_MASK, bool FULL>
__device__ __forceinline__ double dpp_move(double v) {
  const int slo = __double2loint(v), shi = __double2hiint(v);
  int lo, hi;
  if (FULL) {  // every destination lane is written: no `old` operand, hence no copy to set it up
    lo = __builtin_amdgcn_mov_dpp(slo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_mov_dpp(shi, CTRL, ROW_MASK, 0xf, false);
  } else {
    lo = __builtin_amdgcn_update_dpp(0, slo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, shi, CTRL, ROW_MASK, 0xf, false);
  }
  return __hiloint2double(hi, lo);
}

// Sum over the 64 lanes, returned wave-uniform (in SGPRs).  All-VALU: four in-row DPP steps (quad xor 1, quad xor 2,
// half-row mirror, row mirror), row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3, then v_readlane of lane 63.
// ~20 instructions with ALU latencies, instead of six dependent ds_bpermute round trips (~100 cycles each) per sum.
// The summation tree is fixed, so results are deterministic.  Requires all 64 lanes active.
__device__ __forceinline__ double wave_sum_uniform(double v) {
  v += dpp_move<0xB1, 0xf, true>(v);    // quad_perm [1,0,3,2]
  v += dpp_move<0x4E, 0xf, true>(v);    // quad_perm [2,3,0,1]
  v += dpp_move<0x141, 0xf, true>(v);   // row_half_mirror
  v += dpp_move<0x140, 0xf, true>(v);   // row_mirror
  v += dpp_move<0x142, 0xa, false>(v);  // row_bcast15 -> rows 1, 3
  v += dpp_move<0x143, 0xc, false>(v);  // row_bcast31 -> rows 2, 3
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

// ---- packed wave reductions (gradient passes) ----
// v_permlane32_swap / v_permlane16_swap (gfx950) exchange half-waves / 16-lane rows between TWO registers in one VALU
// instruction, so two per-lane partial sums can be folded into one register with 3 instructions (2 word swaps + 1 add): the
// lower / even part keeps summing value a, the upper / odd part value b.  Eight sums then cost 12 + 6 folding instructions and
// TWO in-row DPP reductions instead of eight full ones (58 instead of 176 VALU instructions with the read-outs).
__device__ __forceinline__ double fold32(double a, double b) {  // lanes 0-31: a.lo + a.hi, lanes 32-63: b.lo + b.hi
  const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double fold16(double a, double b) {  // rows: a.r0 + a.r1 | b.r0 + b.r1 | a.r2 + a.r3 | b.r2 + b.r3
  const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double row_sum(double v) {  // every lane: the sum over its 16-lane row (fixed tree)
  v += dpp_move<0xB1, 0xf, true>(v);    // quad_perm [1,0,3,2]
  v += dpp_move<0x4E, 0xf, true>(v);    // quad_perm [2,3,0,1]
  v += dpp_move<0x141, 0xf, true>(v);   // row_half_mirror
  v += dpp_move<0x140, 0xf, true>(v);   // row_mirror
  return v;
}
__device__ __forceinline__ double lane_value(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
// out[i] = sum over the 64 lanes of v[i], wave-uniform, for four / eight values at a time
__device__ __forceinline__ void wave_sum4_uniform(double v0, double v1, double v2, double v3, double (&out)[4]) {
  const double q = row_sum(fold16(fold32(v0, v1), fold32(v2, v3)));  // rows hold v0 | v2 | v1 | v3
  out[0] = lane_value(q, 0);
  out[2] = lane_value(q, 16);
  out[1] = lane_value(q, 32);
  out[3] = lane_value(q, 48);
}
__device__ __forceinline__ void wave_sum2_uniform(double a, double b, double& sa, double& sb) {
  const double p = fold32(a, b);
  const double q = row_sum(fold16(p, p));
  sa = lane_value(q, 0);
  sb = lane_value(q, 32);
}
// The same with the results handed back through per-wave LDS scratch (N doubles at `scr`) as broadcast reads, i.e. wave-uniform
// values in VGPRs: read out with v_readlane they land in SGPRs, and in the MC kernel -- whose scalar register file is full --
// every one of them was spilled to a VGPR lane (v_writelane) and fetched back (v_readlane), two VALU slots each way.
template <int N>
__device__ __forceinline__ void wave_sum_packed_lds(const double (&v)[N], double* __restrict__ scr, int lane, double (&out)[N]) {
  static_assert(N % 4 == 0, "packed reductions work on multiples of four values");
  volatile __attribute__((address_space(3))) double* S = (volatile __attribute__((address_space(3))) double*)scr;
  const int row = lane >> 4;
  const int slot = ((row & 1) << 1) | (row >> 1);  // rows hold v0 | v2 | v1 | v3
#pragma unroll
  for (int i = 0; i < N; i += 4) {
    const double q = row_sum(fold16(fold32(v[i], v[i + 1]), fold32(v[i + 2], v[i + 3])));
    if ((lane & 15) == 0) S[i + slot] = q;
  }
#pragma unroll
  for (int i = 0; i < N; ++i) out[i] = S[i];  // (LDS operations of one wave complete in order)
}

template <int N>
__device__ __forceinline__ void wave_sum_packed(const double (&v)[N], double (&out)[N]) {
  static_assert(N % 4 == 0, "packed reductions work on multiples of four values");
#pragma unroll
  for (int i = 0; i < N; i += 4) {
    double o[4];
    wave_sum4_uniform(v[i], v[i + 1], v[i + 2], v[i + 3], o);
    out[i] = o[0];
    out[i + 1] = o[1];
    out[i + 2] = o[2];
    out[i + 3] = o[3];
  }
}

// dst[i] = sum over the 64 lanes of v[i], i < N (N padded to a multiple of four with zeros): the packed folds above, the four row
// sums of a group written straight to memory by the first lane of each row -- for callers that only need the sums in LDS (the
// workgroup-per-sample kernel's per-wave partial slots: 16 sums cost 4 folds + 4 in-row reductions instead of 16 full ones).
template <int N>
__device__ __forceinline__ void wave_sum_packed_store(const double (&v)[N], double* __restrict__ dst, int lane) {
  constexpr int NP = (N + 3) / 4 * 4;
  const int row = lane >> 4;
  const int slot = ((row & 1) << 1) | (row >> 1);  // rows hold v0 | v2 | v1 | v3
#pragma unroll
  for (int i = 0; i < NP; i += 4) {
    const double a = v[i], b = (i + 1 < N) ? v[i + 1 < N ? i + 1 : 0] : 0.0, c = (i + 2 < N) ? v[i + 2 < N ? i + 2 : 0] : 0.0,
                 d = (i + 3 < N) ? v[i + 3 < N ? i + 3 : 0] : 0.0;
    const double q = row_sum(fold16(fold32(a, b), fold32(c, d)));
    if ((lane & 15) == 0 && i + slot < N) dst[i + slot] = q;
  }
}

__device__ __forceinline__ double uniform(double v) {
  // all lanes hold the same bits after a butterfly; tell the compiler so (value moves to SGPRs, branches become scalar)
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

// A coordinate in the frame of the tables: centred, then scaled.
__device__ __forceinline__ double to_frame(const KgMcParams& P, double v, int r) { return (v - P.center[r]) * P.inv_lp[r]; }

// Armijo trial points are not limited to the domain and, with small length scales (step ~ gradient ~ 1 / length), land
// millions of length scales away, where sqrt(5 r2) * 64 / ln2 leaves the 32-bit exponent arithmetic of exp_nonpos_tab.
// Every tabulated point lies within kTableExtent = 1e5 length scales of the centre per coordinate (validated on the host),
// so a query beyond kFarRadius from the centre has covariance exactly 0 with all of them: the wave-per-sample kernel skips
// such a pass (eval_loop), the workgroup-per-sample kernel pulls the query back to kQueryClamp per coordinate (same zeros).
// One wave-uniform test per pass either way.
constexpr double kTableExtent = 1.0e5;
constexpr double kFarRadius = 5.657e5 + 400.0;  // sqrt(kMaxDimPadded = 32) * kTableExtent + 400
constexpr double kQueryClamp = 1.0e6;
template <int DP>
__device__ __forceinline__ void clamp_query(double (&xq)[DP]) {
  double qq = 0.0;
#pragma unroll
  for (int k = 0; k < DP; ++k) qq = fma(xq[k], xq[k], qq);
  if (!(uniform(qq) <= kQueryClamp * kQueryClamp)) {
#pragma unroll
    for (int k = 0; k < DP; ++k) xq[k] = fmin(fmax(xq[k], -kQueryClamp), kQueryClamp);
  }
}

// Radial scalars divided by alpha (alpha is folded into the weights): base = cov[0,0], first = first-derivative
// coefficient, second = Hessian-product coefficient (device_cov.hpp), all in the FRAME of the tables.  For the Matern kernel
// the frame's scale is sqrt(5) / length (KgMcParams::inv_lp carries the sqrt(5)), so r2 here is 5 r^2 and a = sqrt(r2) needs
// no multiplication; with differences that are sqrt(5) times larger the derivative coefficients shrink by 5 and 25:
//   d base / d q'_k = (1/3) e^-a (1 + a) (x'_k - q'_k),   d first / d (x' - q')_k = -(1/3) e^-a (x' - q')_k.
template <int COV, bool NEED_FIRST, bool NEED_SECOND>
__device__ __forceinline__ void radial3(double r2, const double* __restrict__ etab, double& base, double& first,
                                        double& second) {
  // r2 >= 1e-300 by construction (the distance accumulation starts from 1e-300, see eval_loop)
  if (COV == MOE_COV_SQUARE_EXPONENTIAL) {
    base = exp_nonpos_tab(fmax(-0.5 * r2, -1000.0), etab);  // (r2 can reach 1e13 here: keep the table exp in range)
    first = base;
    second = base;
  } else {
#if MOE_KG_FAST_SQRT
    const double a = sqrt_pos_fast(r2);
#else
    const double a = sqrt_pos(r2);
#endif
    const double e = exp_nonpos_tab(-a, etab);
    base = e * fma(a, fma(a, 1.0 / 3.0, 1.0), 1.0);  // e^-a (1 + a + a^2/3)
    first = NEED_FIRST ? (1.0 / 3.0) * (e * (a + 1.0)) : 0.0;
    second = NEED_SECOND ? (1.0 / 3.0) * e : 0.0;
  }
}

// One pass over the n + u points for the wave's sample: returns f = -mu_after(x) and (if WG) grad f in table-row order.
// xq = scaled query coordinates in table-row order (wave-uniform).  xs = coordinate table [tile][DP][64] (LDS, or global
// when it does not fit), aw = this wave's weights [tile][1+G][64] in LDS (zero beyond the real points, so padded lanes
// contribute exactly 0).
// Tile loads from LDS are volatile loads through an explicit LDS pointer: they must stay single ds_read_b64.  Merged into
// ds_read2st64_b64 (what the load/store optimiser makes of two loads 512 B apart) they run at half the LDS rate on gfx950
// -- 126 vs 218 B/clk/CU measured (tools/ldsbench.hip) -- and the tile loop moves 4.6 KB per tile and wavefront.
typedef const volatile __attribute__((address_space(3))) double* lds_tile_ptr;
template <bool LDS>
struct tile_ptr {
  typedef const double* type;
};
template <>
struct tile_ptr<true> {
  typedef lds_tile_ptr type;
};

//
// With the table in LDS (XL) every tile carries one more row, |x_j - c|^2 of the centred scaled coordinates, and the value
// passes -- nine in ten of all passes -- get the squared distance as |x_j|^2 + |q|^2 - 2 x_j . q: DP FMAs, one add and one
// max per point instead of DP subtractions + DP FMAs (6 of 43 VALU instructions per point at DP = 8).  The coordinates
// and the queries are centred on the training-set mean before they are scaled (KgMcParams::center) so that the three terms
// are of the size of the distance itself wherever the domain sits: the rounding error of r2 stays within a small multiple of the direct form's
// (absolute ~1e-15 at unit-box scales; the kernel is smooth at r = 0, so close pairs lose nothing).  Gradient passes keep
// the direct differences (they need them anyway).
// Q2IN (value passes of the frame line search): `xq_in` holds q2 = -2 x (what the dot-product distances multiply the table rows
// with), so a trial point costs DP fmas to set up instead of DP frame conversions + DP scalings.  FRAMEG: the gradient is
// returned with respect to the FRAME coordinates (not multiplied by the frame scale).
template <int DP, int G, bool WG, int COV, bool SMALL, bool XL, bool Q2IN = false, bool FRAMEG = false>
__device__ __forceinline__ double eval_loop(const double* __restrict__ xs, const double* __restrict__ aw,
                                            const double* __restrict__ etab, int ntiles, double mean,
                                            const double (&xq_in)[DP], const double* inv_lp, double (&grad)[DP], int lane,
                                            double* __restrict__ scr = nullptr) {
  // GDOT: the gradient pass in dot-product form too (no derivative observations): r2 from the |x|^2 row, and
  //   grad = sum_j coef_j (x_j - q) = sum_j coef_j x_j - q sum_j coef_j
  // needs no differences at all -- DP fmas + one add per point instead of DP subtractions + 2 DP fmas (in the centred frame
  // the two terms are of the size of the gradient itself: same argument as for the distances).
  constexpr bool GDOT = MOE_KG_GRAD_DOT && WG && XL && G == 0;
  constexpr bool DOT = XL && (!WG || GDOT);  // squared distance from the |x|^2 row
  constexpr int XR = DP + (XL ? 1 : 0);  // rows per coordinate tile
  double q2[DP], xq[DP];
  double qq;
  if (Q2IN) {
    double ss = 0.0;
#pragma unroll
    for (int k = 0; k < DP; ++k) {
      q2[k] = xq_in[k];
      ss = fma(q2[k], q2[k], ss);
      if (!DOT || G > 0) xq[k] = -0.5 * q2[k];
    }
    qq = fma(ss, 0.25, 1.0e-300);
  } else {
    qq = 1.0e-300;
#pragma unroll
    for (int k = 0; k < DP; ++k) {
      xq[k] = xq_in[k];
      qq = fma(xq[k], xq[k], qq);
      q2[k] = -2.0 * xq[k];
    }
  }
  // A trial point more than kFarRadius length scales from the centre is > 400 length scales from every tabulated point
  // (all inside the ball of radius sqrt(32) * kTableExtent): every covariance underflows to exactly 0 and the posterior mean
  // IS the prior mean -- the pass is skipped.  Closer than that, sqrt(r2) * 64 / ln2 < 2^31: exp_nonpos_tab is in range.
  if (!WG && !(uniform(qq) <= kFarRadius * kFarRadius)) return -mean;  // (uniform: a scalar branch, and the callers' Armijo decisions stay scalar)
  double accf = 0.0, accs = 0.0;
  double accg[DP];
  double accd[G > 0 ? G : 1];
#pragma unroll
  for (int k = 0; k < DP; ++k) accg[k] = 0.0;
#pragma unroll
  for (int a = 0; a < (G > 0 ? G : 1); ++a) accd[a] = 0.0;
  // Software-pipelined tile loop: the next tile's coordinates / weights are requested from LDS before the current
  // tile's ~55 FP64 instructions run, so the ds_read latency hides behind them instead of stalling every tile.
  typename tile_ptr<XL>::type xt = (typename tile_ptr<XL>::type)(xs + lane);
  lds_tile_ptr wt = (lds_tile_ptr)(aw + lane);
  constexpr int NX = DP + (DOT ? 1 : 0);  // rows this pass reads: the |x|^2 row only where it is used
  double cx[NX], cw[1 + G];
#pragma unroll
  for (int k = 0; k < NX; ++k) cx[k] = xt[k * 64];
#pragma unroll
  for (int a = 0; a < 1 + G; ++a) cw[a] = wt[a * 64];
#pragma unroll(SMALL ? 1 : (WG ? 2 : 4))
  for (int t = 0; t < ntiles; ++t) {
    double nx[NX], nw[1 + G];
    // unconditional advance (constant stride: the unrolled tiles share one address register and use immediate offsets);
    // the last iteration prefetches one tile past the end -- the host pads both arrays by one tile, the values are unused
    xt += XR * 64;
    wt += (1 + G) * 64;
#pragma unroll
    for (int k = 0; k < NX; ++k) nx[k] = xt[k * 64];
#pragma unroll
    for (int a = 0; a < 1 + G; ++a) nw[a] = wt[a * 64];
    double diff[DP];
    double r2;
    if (DOT) {
      r2 = cx[DP] + qq;
#pragma unroll
      for (int k = 0; k < DP; ++k) r2 = fma(cx[k], q2[k], r2);
      r2 = fmax(r2, 1.0e-300);  // rounding can leave a point that coincides with the query a hair below zero
      if (G > 0) {
#pragma unroll
        for (int a = 0; a < G; ++a) diff[a] = cx[a] - xq[a];
      }
    } else {
      r2 = 1.0e-300;  // keeps r2 > 0 for the rsq-based sqrt at no cost (invisible next to any r2 >= 1e-284)
#pragma unroll
      for (int k = 0; k < DP; ++k) {
        diff[k] = cx[k] - xq[k];
        r2 = fma(diff[k], diff[k], r2);
      }
    }
    const double w0 = cw[0];  // alpha * (function-value weight)
    double base, first, second;
    radial3<COV, (WG || G > 0), (WG && G > 0)>(r2, etab, base, first, second);
    double sd = 0.0;  // sum_a w_a diff[a]  (derivative-observation weights; table rows a < G are the observed dims)
    if (G > 0) {
#pragma unroll
      for (int a = 0; a < G; ++a) sd = fma(cw[1 + a], diff[a], sd);
    }
    accf = fma(w0, base, accf);
    if (G > 0) accf = fma(first, sd, accf);
    if (GDOT) {
      const double coef = w0 * first;
      accs += coef;
#pragma unroll
      for (int k = 0; k < DP; ++k) accg[k] = fma(coef, cx[k], accg[k]);
    } else if (WG) {
      double coef = w0 * first;
      if (G > 0) {
        coef = fma(second, sd, coef);
#pragma unroll
        for (int a = 0; a < G; ++a) accd[a] = fma(first, cw[1 + a], accd[a]);
      }
#pragma unroll
      for (int k = 0; k < DP; ++k) accg[k] = fma(coef, diff[k], accg[k]);
    }
#pragma unroll
    for (int k = 0; k < NX; ++k) cx[k] = nx[k];
#pragma unroll
    for (int a = 0; a < 1 + G; ++a) cw[a] = nw[a];
  }
  if (GDOT) {
    double sg[DP], sf, ss;
    if (scr != nullptr)
      wave_sum_packed_lds<DP>(accg, scr, lane, sg);
    else
      wave_sum_packed<DP>(accg, sg);
    wave_sum2_uniform(accf, accs, sf, ss);
#pragma unroll
    for (int k = 0; k < DP; ++k) {
      const double v = fma(-xq[k], ss, sg[k]);  // sum coef x_k - q_k sum coef
      grad[k] = FRAMEG ? -v : -(v * inv_lp[k]);
    }
    return -(mean + sf);
  }
  const double mu = mean + wave_sum_uniform(accf);
  if (WG) {
    double sg[DP];
    if (scr != nullptr)
      wave_sum_packed_lds<DP>(accg, scr, lane, sg);  // DP sums folded into DP / 4 in-row reductions
    else
      wave_sum_packed<DP>(accg, sg);
#pragma unroll
    for (int k = 0; k < DP; ++k) {
      // d mu / d x_k = inv_l[k] * ( sum coef (Xs_k - xq_k)  -  [k < G] sum first w_k );   f = -mu
      double v = sg[k];
      if (G > 0 && k < G) v -= wave_sum_uniform(accd[k < G ? k : 0]);
      grad[k] = FRAMEG ? -v : -(v * inv_lp[k]);
    }
  }
  return -mu;
}

// The covariance type is wave-uniform: branch ONCE per pass (a branch inside the tile loop would split it into basic blocks
// and stop the scheduler from interleaving the independent per-tile dependency chains).
template <int DP, int G, bool WG, bool SMALL, bool XL, bool Q2IN = false, bool FRAMEG = false>
__device__ __forceinline__ double eval_pass(const double* __restrict__ xs, const double* __restrict__ aw,
                                            const double* __restrict__ etab, int ntiles, int cov_type, double mean,
                                            const double (&xq)[DP], const double* inv_lp, double (&grad)[DP], int lane,
                                            double* __restrict__ scr = nullptr) {
  if (cov_type == MOE_COV_SQUARE_EXPONENTIAL)
    return eval_loop<DP, G, WG, MOE_COV_SQUARE_EXPONENTIAL, SMALL, XL, Q2IN, FRAMEG>(xs, aw, etab, ntiles, mean, xq, inv_lp, grad,
                                                                                      lane, scr);
  return eval_loop<DP, G, WG, MOE_COV_MATERN_NU_2P5, SMALL, XL, Q2IN, FRAMEG>(xs, aw, etab, ntiles, mean, xq, inv_lp, grad, lane,
                                                                                 scr);
}

// T Armijo trial points x' + alpha_t g s^2, alpha_t = alpha0 / 2^t, in ONE sweep over the tiles (no derivative observations, LDS
// table).  With x2 = -2 x' and d2 = -2 g s^2 (line_search_frame) the trial's q2 is x2 + alpha_t d2, so
//   r2_t = |x_j|^2 + |q_t|^2 + x_j . q2_t = (|x_j|^2 + x_j . x2) + alpha_t (x_j . d2) + |q_t|^2:
// the two projections p0, p1 are formed once per point (2 DP fmas) and every trial costs an add, an fma and a max on top of
// its square root / exp / polynomial -- (2 DP + 1) / T + 3 instructions per point for r2 instead of DP + 2, and one set of
// coordinate / weight loads, one reduction round and one decision round per T trials.  |q_t|^2 = (|x2|^2 + 2 alpha_t x2.d2 +
// alpha_t^2 |d2|^2) / 4 (wave-uniform).  Returns false without evaluating when a trial lies beyond kFarRadius (see eval_loop;
// |q(alpha)|^2 is convex in alpha, so the two ends of the bracket bound all trials).
// XL = false (coordinates streamed from L2: no |x|^2 row): the same with direct differences about x0' = -x2 / 2, as the
// workgroup-per-sample kernel does (point_terms_multi): r2_t = |x_j - x0'|^2 - 2 alpha_t (x_j - x0') . dv + alpha_t^2 |dv|^2, dv = -d2 / 2
// -- there one coordinate sweep through L2 serves T trials instead of one, which is most of what that kernel paid per trial.
// G > 0 (derivative observations; table rows a < G are the observed dimensions): a point also carries the derivative-weight sum
// sum_a w_a (x_j - q_t)_a = sdA - alpha_t sdB,  sdA = sum_a w_a (x_j - x0')_a,  sdB = sum_a w_a dv_a  (2 G fmas once per point, one
// fma per trial), multiplied by the first-derivative coefficient of the trial's distance.
// (eval_multi_loop_s: the three sums |x2|^2, x2.d2, |d2|^2 -- fixed along a trial line -- handed in by a caller that forms them once
//  per bracket, kg_mc_lane.hpp; eval_multi_loop forms them itself, in the same order)
template <int DP, int COV, int T, bool SMALL, bool XL = true, int G = 0>
__device__ __forceinline__ bool eval_multi_loop_s(const double* __restrict__ xs, const double* __restrict__ aw,
                                                  const double* __restrict__ etab, int ntiles, double mean, const double (&x2)[DP],
                                                  const double (&d2)[DP], double sxx, double sxd, double sdd, double alpha0, int lane,
                                                  double (&f)[T]) {
  constexpr int WR = 1 + G;  // weight rows per tile
  double al[T], qq[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    al[t] = (t == 0) ? alpha0 : 0.5 * al[t > 0 ? t - 1 : 0];
    qq[t] = fma(0.25, fma(al[t], fma(al[t], sdd, 2.0 * sxd), sxx), 1.0e-300);
  }
  if (!(uniform(fmax(qq[0], 0.25 * sxx)) <= kFarRadius * kFarRadius)) return false;
  double acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = 0.0;
  constexpr int NX = DP + (XL ? 1 : 0);  // rows per coordinate tile
  typename tile_ptr<XL>::type xt = (typename tile_ptr<XL>::type)(xs + lane);
  lds_tile_ptr wt = (lds_tile_ptr)(aw + lane);
  double cx[NX], cwa[WR];
#pragma unroll
  for (int k = 0; k < NX; ++k) cx[k] = xt[k * 64];
#pragma unroll
  for (int a = 0; a < WR; ++a) cwa[a] = wt[a * 64];
  double x0[DP], dv[DP], tt[T];  // (direct-difference form, and the derivative-weight sums)
  if (!XL || G > 0) {
#pragma unroll
    for (int k = 0; k < DP; ++k) {
      x0[k] = -0.5 * x2[k];
      dv[k] = -0.5 * d2[k];
    }
#pragma unroll
    for (int t = 0; t < T; ++t) tt[t] = (al[t] * al[t]) * (0.25 * sdd);  // alpha_t^2 |dv|^2
  }
#pragma unroll(SMALL ? 1 : 2)  // (four tiles for few trials: +-0.3 %, r5)
  for (int tile = 0; tile < ntiles; ++tile) {
    double nx[NX], nwa[WR];
    xt += NX * 64;  // (one tile of padding behind both arrays: see eval_loop)
    wt += WR * 64;
#pragma unroll
    for (int k = 0; k < NX; ++k) nx[k] = xt[k * 64];
#pragma unroll
    for (int a = 0; a < WR; ++a) nwa[a] = wt[a * 64];
    const double cw = cwa[0];
    // derivative-weight sum of trial t: sum_a w_a (x_ja - x0_a - alpha_t dv_a) = sdA - alpha_t sdB
    double sdA = 0.0, sdB = 0.0;
    if (G > 0) {
#pragma unroll
      for (int a = 0; a < G; ++a) {
        sdA = fma(cwa[1 + a], cx[a] - x0[a], sdA);
        sdB = fma(cwa[1 + a], dv[a], sdB);
      }
    }
    if (XL) {
      double p0 = cx[XL ? DP : 0];
#pragma unroll
      for (int k = 0; k < DP; ++k) p0 = fma(cx[k], x2[k], p0);
      double p1 = cx[0] * d2[0];
#pragma unroll
      for (int k = 1; k < DP; ++k) p1 = fma(cx[k], d2[k], p1);
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const double r2 = fmax(fma(al[t], p1, p0 + qq[t]), 1.0e-300);
        double base, first, second;
        radial3<COV, (G > 0), false>(r2, etab, base, first, second);
        acc[t] = fma(cw, base, acc[t]);
        if (G > 0) acc[t] = fma(first, fma(-al[t], sdB, sdA), acc[t]);
      }
    } else {
      double A = 1.0e-300, B = 0.0;
#pragma unroll
      for (int k = 0; k < DP; ++k) {
        const double d0 = cx[k] - x0[k];
        A = fma(d0, d0, A);
        B = fma(d0, dv[k], B);
      }
      const double mB2 = -2.0 * B;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const double r2 = fmax(fma(al[t], mB2, A + tt[t]), 1.0e-300);
        double base, first, second;
        radial3<COV, (G > 0), false>(r2, etab, base, first, second);
        acc[t] = fma(cw, base, acc[t]);
        if (G > 0) acc[t] = fma(first, fma(-al[t], sdB, sdA), acc[t]);
      }
    }
#pragma unroll
    for (int k = 0; k < NX; ++k) cx[k] = nx[k];
#pragma unroll
    for (int a = 0; a < WR; ++a) cwa[a] = nwa[a];
  }
  // T wave sums, folded four / two at a time (packed reductions above)
  double sum[T];
  constexpr int T4 = T / 4 * 4;
#pragma unroll
  for (int t = 0; t < T4; t += 4) {
    double o[4];
    wave_sum4_uniform(acc[t], acc[t + 1], acc[t + 2], acc[t + 3], o);
    sum[t] = o[0];
    sum[t + 1] = o[1];
    sum[t + 2] = o[2];
    sum[t + 3] = o[3];
  }
  if constexpr (T - T4 >= 2) wave_sum2_uniform(acc[T4], acc[T4 + 1], sum[T4], sum[T4 + 1]);
  if constexpr (((T - T4) & 1) != 0) sum[T - 1] = wave_sum_uniform(acc[T - 1]);
#pragma unroll
  for (int t = 0; t < T; ++t) f[t] = -(mean + sum[t]);
  return true;
}

// T Armijo trials in one sweep with EVERY trial computed exactly as a single-trial value pass computes it (r6): q2_t = x2 + alpha_t d2
// and |q_t|^2 formed per trial as eval_loop<..., Q2IN> forms them, r2 = (|x_j|^2 + |q_t|^2) then the DP fmas in row order, the tile
// sums in tile order, one wave_sum_uniform per trial -- the bits of T separate passes (T DP fmas per point instead of the 2 DP + 2 T of
// the line decomposition above: no dearer up to DP = 4 and T = 5), with one set of coordinate / weight loads and one decision round.
// For the small shapes whose end points the reference's 100-step x 10-restart fixtures pin: there the line decomposition's rounding moved
// them by 1.01e-6 (kg.hip), so until r6 those shapes ran one trial per pass.  LDS table only.  Returns false without evaluating when a
// trial lies beyond kFarRadius (the single-trial pass returns the prior mean there: the caller falls back to it).
template <int DP, int COV, int T, int G>
__device__ __forceinline__ bool eval_multi_exact(const double* __restrict__ xs, const double* __restrict__ aw,
                                                 const double* __restrict__ etab, int ntiles, double mean, const double (&x2)[DP],
                                                 const double (&d2)[DP], double alpha0, int lane, double (&f)[T]) {
  constexpr int WR = 1 + G;
  constexpr int NX = DP + 1;
  double al[T], qq[T], q2[T][DP], xq[T][G > 0 ? G : 1];
  bool near_all = true;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    al[t] = (t == 0) ? alpha0 : 0.5 * al[t > 0 ? t - 1 : 0];
    double ss = 0.0;
#pragma unroll
    for (int k = 0; k < DP; ++k) {
      q2[t][k] = fma(al[t], d2[k], x2[k]);
      ss = fma(q2[t][k], q2[t][k], ss);
    }
#pragma unroll
    for (int a = 0; a < G; ++a) xq[t][a] = -0.5 * q2[t][a];
    qq[t] = fma(ss, 0.25, 1.0e-300);
    near_all = near_all && (uniform(qq[t]) <= kFarRadius * kFarRadius);
  }
  if (!near_all) return false;
  double acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = 0.0;
  lds_tile_ptr xt = (lds_tile_ptr)(xs + lane);
  lds_tile_ptr wt = (lds_tile_ptr)(aw + lane);
  double cx[NX], cw[WR];
#pragma unroll
  for (int k = 0; k < NX; ++k) cx[k] = xt[k * 64];
#pragma unroll
  for (int a = 0; a < WR; ++a) cw[a] = wt[a * 64];
#pragma unroll 1
  for (int tile = 0; tile < ntiles; ++tile) {
    double nx[NX], nw[WR];
    xt += NX * 64;  // (one tile of padding behind both arrays: see eval_loop)
    wt += WR * 64;
#pragma unroll
    for (int k = 0; k < NX; ++k) nx[k] = xt[k * 64];
#pragma unroll
    for (int a = 0; a < WR; ++a) nw[a] = wt[a * 64];
    const double w0 = cw[0];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      double r2 = cx[DP] + qq[t];
#pragma unroll
      for (int k = 0; k < DP; ++k) r2 = fma(cx[k], q2[t][k], r2);
      r2 = fmax(r2, 1.0e-300);
      double base, first, second;
      radial3<COV, (G > 0), false>(r2, etab, base, first, second);
      acc[t] = fma(w0, base, acc[t]);
      if (G > 0) {
        double sd = 0.0;
#pragma unroll
        for (int a = 0; a < G; ++a) sd = fma(cw[1 + a], cx[a] - xq[t][a], sd);
        acc[t] = fma(first, sd, acc[t]);
      }
    }
#pragma unroll
    for (int k = 0; k < NX; ++k) cx[k] = nx[k];
#pragma unroll
    for (int a = 0; a < WR; ++a) cw[a] = nw[a];
  }
#pragma unroll
  for (int t = 0; t < T; ++t) f[t] = -(mean + wave_sum_uniform(acc[t]));
  return true;
}

template <int DP, int COV, int T, bool SMALL, bool XL = true, int G = 0>
__device__ __forceinline__ bool eval_multi_loop(const double* __restrict__ xs, const double* __restrict__ aw,
                                                const double* __restrict__ etab, int ntiles, double mean, const double (&x2)[DP],
                                                const double (&d2)[DP], double alpha0, int lane, double (&f)[T]) {
  double sxx = 0.0, sxd = 0.0, sdd = 0.0;
#pragma unroll
  for (int k = 0; k < DP; ++k) {
    sxx = fma(x2[k], x2[k], sxx);
    sxd = fma(x2[k], d2[k], sxd);
    sdd = fma(d2[k], d2[k], sdd);
  }
  return eval_multi_loop_s<DP, COV, T, SMALL, XL, G>(xs, aw, etab, ntiles, mean, x2, d2, sxx, sxd, sdd, alpha0, lane, f);
}

// TensorProductDomain::LimitUpdate (gpp_domain.cpp:64-105) on one coordinate.
__device__ __forceinline__ double limit_update_1d(double lo, double hi, double max_relative_change, double x, double desired) {
  double dist = fmin(x - lo, hi - x);
  if (fabs(desired) > max_relative_change * dist) desired = copysign(max_relative_change * dist, desired);
  const double next = x + desired;
  if (next < lo || next > hi) {
    if (next < lo) {
      dist = lo - x;
      desired = (x + desired * 0.5 < lo) ? dist * 0.5 : desired * 0.5;
    } else {
      dist = hi - x;
      desired = (x + desired * 0.5 > hi) ? dist * 0.5 : desired * 0.5;
    }
  }
  return desired;
}

// 2-norm of the first `size` entries.  The reference's VectorNorm (gpp_linear_algebra.cpp:53-72) uses the scaled
// (overflow-safe) recurrence with one FP64 division per entry; here the value only feeds the two threshold tests of the
// line search (|step| < tolerance / max_steps, |x - x_start| > tolerance) on O(1)-magnitude vectors, where sqrt(sum v^2)
// agrees with it to 2 ulp, so the plain form (an order of magnitude fewer instructions) is used.
template <int DP>
__device__ __forceinline__ double vector_norm(const double (&v)[DP], int size) {
  double ss = 0.0;
#pragma unroll
  for (int i = 0; i < DP; ++i)
    if (i < size) ss = fma(v[i], v[i], ss);
  return sqrt(ss);
}

// out[r] = v[perm[r]] for wave-uniform v (select chains on uniform data; identity when the GP has no derivatives)
template <int DP, int G>
__device__ __forceinline__ void to_table_order(const double (&v)[DP], const int* perm, double (&out)[DP]) {
  if (G == 0) {
#pragma unroll
    for (int r = 0; r < DP; ++r) out[r] = v[r];
  } else {
#pragma unroll
    for (int r = 0; r < DP; ++r) {
      double o = 0.0;
#pragma unroll
      for (int k = 0; k < DP; ++k) o = (perm[r] == k) ? v[k] : o;
      out[r] = o;
    }
  }
}

template <int DP, int G>
__device__ __forceinline__ void from_table_order(const double (&v)[DP], const int* perm, double (&out)[DP]) {
  if (G == 0) {
#pragma unroll
    for (int k = 0; k < DP; ++k) out[k] = v[k];
  } else {
#pragma unroll
    for (int k = 0; k < DP; ++k) {
      double o = 0.0;
#pragma unroll
      for (int r = 0; r < DP; ++r) o = (perm[r] == k) ? v[r] : o;
      out[k] = o;
    }
  }
}

#ifndef MOE_KG_STREAM_TRIALS
#define MOE_KG_STREAM_TRIALS 8
#endif
// (one trial more than the previous step consumed in a step's first sweep: measured slower everywhere -- C5 MC 4.22 -> 4.44 ms, d = 16 at
//  n = 1000 0.188 -> 0.197 -- kept as a switch, off)
#ifndef MOE_KG_STREAM_PRED_EXTRA
#define MOE_KG_STREAM_PRED_EXTRA 0
#endif
#ifndef MOE_KG_LDS_CARRY
#define MOE_KG_LDS_CARRY 1
#endif
#ifndef MOE_KG_STREAM_FIRST
#define MOE_KG_STREAM_FIRST 6
#endif
#ifndef MOE_KG_STREAM_FOLLOWUP
#define MOE_KG_STREAM_FOLLOWUP 4
#endif
constexpr int kPartLen = 48;  // doubles per partial slot of a packed reduction: f | DP gradient sums | sum of coefficients | G derivative sums (<= 1 + 32 + 1 + 12)
constexpr int kLsRows = 6;    // line-search vectors per wave in LDS (line_search_lds): x | masked gradient | step | x at restart start | x0 and dv of the trial line (frame)
// per-wave LDS scratch of the wide-dimension evaluator (WideEval, d > 16) behind the z / beta scratch of a weight slab
constexpr int kWideScratch = kLsRows * kMaxDimPadded + kPartLen;
// Which wave-per-sample instantiations use it: the streamed ones from 16 coordinate rows on.  (r3, streamed tables at 8 / 12 rows:
// the evaluator above with its fused value + gradient passes stays ahead -- n = 1500 .. 3000, d = 8: 1.15 / 1.39 / 2.15 ms per
// evaluation against 1.24 / 1.62 / 2.52; at 16 rows WideEval wins, n = 1000: 0.27 against 0.345 ms.)
constexpr bool wide_eval(int dp, bool xlds) { return !xlds && dp >= 16; }

// Evaluator of the wave-per-sample kernel: one pass = eval_pass over the LDS tables.
template <int DP, int G, bool SMALL, bool XL>
struct WaveEval {
  const double* __restrict__ xs;
  const double* __restrict__ aw;
  const double* __restrict__ etab;
  int ntiles, cov_type;
  double mean;
  const double* inv_lp;
  int lane;
  double* __restrict__ scr;  // per-wave LDS scratch for the gradient sums (DP doubles), or NULL
#if MOE_BLOCK_PROF
  unsigned long long c_v = 0, c_g = 0, n_v = 0, n_g = 0;
#endif
  template <bool WG>
  __device__ __forceinline__ double eval(const double (&xq)[DP], double (&grad)[DP]) {
    return eval_pass<DP, G, WG, SMALL, XL>(xs, aw, etab, ntiles, cov_type, mean, xq, inv_lp, grad, lane);
  }
  // frame line search: f and its gradient with respect to the FRAME coordinates at the frame point xf; f at the point whose
  // frame coordinates are -q2 / 2
  __device__ __forceinline__ double eval_grad_frame(const double (&xf)[DP], double (&gradf)[DP]) {
#if MOE_BLOCK_PROF
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
    const double f = eval_pass<DP, G, true, SMALL, XL, false, true>(xs, aw, etab, ntiles, cov_type, mean, xf, inv_lp, gradf, lane, scr);
#if MOE_BLOCK_PROF
    c_g += __builtin_amdgcn_s_memtime() - t0;
    n_g++;
#endif
    return f;
  }
  // up to kMaxTrials Armijo trials alpha / 2^t in one pass (eval_multi_loop): q-KG on the LDS table only
  static constexpr int kMaxTrials = !SMALL ? 5 : 1;
  // T trials and the reference's sequence of decisions over them (gpp_optimization.hpp:752-769), with compile-time indices
  // (a runtime-sized result array would live in scratch memory): stops at the first accepted trial (done), halves alpha and
  // counts `search` for every rejected one, counts consumed trials only.  Returns false (nothing evaluated, nothing changed)
  // when a trial lies beyond the far radius.
  template <int T>
  __device__ __forceinline__ bool armijo_t(const double (&x2)[DP], const double (&d2)[DP], double f0, double norm, double& alpha_n,
                                           int& search, double& ftrial, bool& done, unsigned long long& n_val) {
    double f[T];
    const bool ok = (cov_type == MOE_COV_SQUARE_EXPONENTIAL)
                        ? eval_multi_loop<DP, MOE_COV_SQUARE_EXPONENTIAL, T, SMALL, XL, G>(xs, aw, etab, ntiles, mean, x2, d2, alpha_n, lane, f)
                        : eval_multi_loop<DP, MOE_COV_MATERN_NU_2P5, T, SMALL, XL, G>(xs, aw, etab, ntiles, mean, x2, d2, alpha_n, lane, f);
    if (!ok) return false;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if (!done) {
        ftrial = f[t];
        n_val++;
#if MOE_BLOCK_PROF
        n_v++;
#endif
        if (ftrial - f0 > 0.5 * alpha_n * norm) {
          done = true;
        } else {
          alpha_n *= 0.5;
          if (++search >= 30) done = true;
        }
      }
    }
    return true;
  }
  // one pass over min(want, kMaxTrials) >= 2 trials
  __device__ __forceinline__ bool armijo_batch(int want, const double (&x2)[DP], const double (&d2)[DP], double f0, double norm,
                                               double& alpha_n, int& search, double& ftrial, bool& done,
                                               unsigned long long& n_val) {
    if constexpr (kMaxTrials >= 5) {
#if MOE_BLOCK_PROF
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
      bool ok;
      switch (want) {
        case 2: ok = armijo_t<2>(x2, d2, f0, norm, alpha_n, search, ftrial, done, n_val); break;
        case 3: ok = armijo_t<3>(x2, d2, f0, norm, alpha_n, search, ftrial, done, n_val); break;
        case 4: ok = armijo_t<4>(x2, d2, f0, norm, alpha_n, search, ftrial, done, n_val); break;
        default: ok = armijo_t<5>(x2, d2, f0, norm, alpha_n, search, ftrial, done, n_val); break;
      }
#if MOE_BLOCK_PROF
      c_v += __builtin_amdgcn_s_memtime() - t0;
#endif
      return ok;
    } else {
      return false;
    }
  }
  __device__ __forceinline__ double eval_value_q2(const double (&q2)[DP]) {
    double unused[DP];
#if MOE_BLOCK_PROF
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
    const double f = eval_pass<DP, G, false, SMALL, XL, true, false>(xs, aw, etab, ntiles, cov_type, mean, q2, inv_lp, unused, lane);
#if MOE_BLOCK_PROF
    c_v += __builtin_amdgcn_s_memtime() - t0;
    n_v++;
#endif
    return f;
  }
};

// The inner optimisation of one MC sample from the start point x (in/out, original dimension order): returns the final
// objective value f = -mu_after(x).  `ev.eval<WG>(xq, grad)` evaluates f (and, if WG, grad f in table-row order) at the scaled
// query xq; it must return wave-uniform values.
template <int DP, int G, class EV>
__device__ __forceinline__ double line_search(const KgMcParams& P, EV& ev, double (&x)[DP], unsigned long long& n_val,
                                              unsigned long long& n_grad) {
  // Everything below is in TABLE-ROW order (x, grad, step, bounds): the evaluator works in that order, so no permutation
  // happens per pass; the caller's x is permuted on entry and exit only.  (free_mask marks the optimised coordinates.)
  {
    double xp[DP];
    to_table_order<DP, G>(x, P.perm, xp);
#pragma unroll
    for (int r = 0; r < DP; ++r) x[r] = xp[r];
  }
  const double step_tolerance = P.tolerance / (double)P.max_num_steps;
  double fcur = 0.0;
  // lane k < DP carries coordinate k's bounds for the lane-parallel LimitUpdate below
  const int lane_id = (int)(threadIdx.x & 63u);
  const double lo_l = P.bounds[2 * (lane_id < DP ? lane_id : 0)], hi_l = P.bounds[2 * (lane_id < DP ? lane_id : 0) + 1];

  // GradientDescentOptimizerLineSearch::Optimize (gpp_optimization.hpp:1242-1283) around
  // GradientDescentOptimizationLineSearch (:708-828), as plain nested wave-uniform loops.  The Armijo back-tracking loop --
  // where 5 of every 6 passes are spent -- carries only (alpha, search) through its back edge; x and grad are loop
  // invariant there, so the compiler has no array phis to shuffle (a single-evaluation-site state machine cost ~250
  // v_mov_b64 / lane-spill instructions per pass).  Every posterior-mean value comes from the same dataflow (eval_loop,
  // explicit fma only), so re-evaluating a point reproduces its value bit for bit at any call site, which lets us reuse
  // f(x) where the reference recomputes it.
  if (P.max_num_restarts > 0) {
    double grad[DP], step[DP], xstart[DP], tq[DP], tqp[DP], gp[DP];
#pragma unroll
    for (int k = 0; k < DP; ++k) gp[k] = 0.0;
    for (int restart = 0; restart < P.max_num_restarts; ++restart) {
#pragma unroll
      for (int k = 0; k < DP; ++k) xstart[k] = x[k];
      for (int istep = 0; istep < P.max_num_steps;) {
        // ---- f(x), grad f(x) ----
        #pragma unroll
        for (int r = 0; r < DP; ++r) tqp[r] = to_frame(P, x[r], r);
        const double f0 = ev.template eval<true>(tqp, gp);
        n_grad++;
        fcur = f0;
        double norm = 0.0;
#pragma unroll
        for (int k = 0; k < DP; ++k) {
          grad[k] = ((P.free_mask >> k) & 1u) ? gp[k] : 0.0;  // fidelity / pad coordinates stay pinned
          norm = fma(grad[k], grad[k], norm);
        }
        // pre_mult * (i+1)^-gamma (gpp_optimization.hpp:741); x^-0 == 1 exactly, so gamma == 0 needs no pow()
        double alpha_n = (P.gamma == 0.0) ? P.pre_mult : P.pre_mult * pow((double)(istep + 1), -P.gamma);
        // ---- Armijo back-tracking (.hpp:745-760): unclamped trial points ----
        int search = 0;
        double ftrial;
        while (true) {
#pragma unroll
          for (int k = 0; k < DP; ++k) tq[k] = fma(alpha_n, grad[k], x[k]);
#pragma unroll
          for (int r = 0; r < DP; ++r) tqp[r] = to_frame(P, tq[r], r);
          ftrial = ev.template eval<false>(tqp, gp);
          n_val++;
          if (ftrial - f0 > 0.5 * alpha_n * norm) break;
          alpha_n *= 0.5;
          if (++search >= 30) break;
        }
        // ---- LimitUpdate, then accept only if f improves (.hpp:762-795) ----
        // One coordinate per LANE (lane k < DP handles coordinate k) instead of DP wave-uniform copies of the ~40
        // instruction clamp: the coordinates are gathered into a lane vector with selects, clamped once, and the steps
        // come back as wave-uniform values through v_readlane.  Same arithmetic per coordinate.
        bool changed, nonzero;
        {
          double x_l = 0.0, want_l = 0.0;
#pragma unroll
          for (int k = 0; k < DP; ++k) {
            x_l = (lane_id == k) ? x[k] : x_l;
            want_l = (lane_id == k) ? alpha_n * grad[k] : want_l;
          }
          const bool free_l = lane_id < DP && ((P.free_mask >> (lane_id & 31)) & 1u);
          double step_l = 0.0;
          if (free_l) step_l = limit_update_1d(lo_l, hi_l, P.max_relative_change, x_l, want_l);
          changed = __ballot(free_l && step_l != want_l) != 0ull;
          nonzero = __ballot(free_l && step_l != 0.0) != 0ull;
#pragma unroll
          for (int k = 0; k < DP; ++k) {
            const int slo = __builtin_amdgcn_readlane(__double2loint(step_l), k);
            const int shi = __builtin_amdgcn_readlane(__double2hiint(step_l), k);
            step[k] = __hiloint2double(shi, slo);
          }
        }
        if (search == 30 || !nonzero) break;  // .hpp:781-785: x restored (a zero step re-evaluates f(x) == f0: rejected)
        double obj2 = ftrial;  // clamp left the step untouched: f(x + step) is the last trial value
        if (changed) {
#pragma unroll
          for (int k = 0; k < DP; ++k) tq[k] = x[k] + step[k];
#pragma unroll
          for (int r = 0; r < DP; ++r) tqp[r] = to_frame(P, tq[r], r);
          obj2 = ev.template eval<false>(tqp, gp);
          n_val++;
        }
        if (obj2 <= f0) break;
#pragma unroll
        for (int k = 0; k < DP; ++k) x[k] += step[k];
        fcur = obj2;
        istep += 1;
        if (vector_norm<DP>(step, DP) < step_tolerance) break;  // step is 0 on pinned coordinates
      }
      double delta[DP];
#pragma unroll
      for (int k = 0; k < DP; ++k) delta[k] = xstart[k] - x[k];
      if (!(vector_norm<DP>(delta, DP) > P.tolerance)) break;
    }
  } else {
    // reference returns without touching its outputs (.cpp:425-427): value 0, point filled with 1.0 (.cpp:163)
#pragma unroll
    for (int k = 0; k < DP; ++k) x[k] = (P.perm[k] < P.dim) ? 1.0 : 0.0;
    fcur = 0.0;
  }
  {
    double xo[DP];
    from_table_order<DP, G>(x, P.perm, xo);
#pragma unroll
    for (int k = 0; k < DP; ++k) x[k] = xo[k];
  }
  return fcur;
}

// ---------------------------------------------------------------------------------------------------------------------
// The line search of the wave-per-sample kernel, carried out in the FRAME of the tables (x' = (x - c) s per table row, s the
// frame scale): the same sequence of decisions as line_search above (gpp_optimization.hpp:708-828, 1242-1283;
// TensorProductDomain::LimitUpdate, gpp_domain.cpp:64-105 -- every comparison in them is invariant under a positive affine
// map of a coordinate), but a pass no longer converts its point: with g the gradient in the original coordinates and
// gf = g / s the one the evaluator returns,
//     x + alpha g   <->   x' + alpha gf s^2,       |g|^2 = sum (gf s)^2,       |step|^2 = sum (step' / s)^2,
// and the value passes take q2 = -2 (x' + alpha gf s^2) = fma(alpha, d2, x2) directly (eval_loop Q2IN).  The per-row
// constants live in an LDS block `cst` (written once per workgroup): [0, DP) frame scale s | [DP, 2DP) 1 / s (0 in pad rows) |
// [2DP, 3DP) centre c | [3DP, 4DP) lower bound' | [4DP, 5DP) upper bound' (bounds already in the frame) | [5DP, 6DP) the value a
// pinned row holds.  `scr` is 3 DP doubles of per-wave LDS scratch (the restart's starting point | the evaluator's gradient sums |
// the clamped step).  Nothing of
// KgMcParams' per-row arrays is touched between the first and the last pass of a sample: they used to sit in ~100 SGPRs and
// were spilled to / reloaded from VGPR lanes (v_readlane -- a VALU slot each) around every pass.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kCstRows = 6;  // arrays of DP doubles in the LDS constant block

template <int DP>
__device__ __forceinline__ void fill_frame_constants(const KgMcParams& P, double* __restrict__ cst) {
  const int r = threadIdx.x;
  if (r < DP) {
    const double sc = P.inv_lp[r];
    cst[r] = sc;
    cst[DP + r] = (sc != 0.0) ? 1.0 / sc : 0.0;
    cst[2 * DP + r] = P.center[r];
    cst[3 * DP + r] = (P.bounds[2 * r] - P.center[r]) * sc;
    cst[4 * DP + r] = (P.bounds[2 * r + 1] - P.center[r]) * sc;
    cst[5 * DP + r] = (P.perm[r] < P.dim) ? 1.0 : 0.0;  // what a pinned row holds: fidelity coordinates 1, pads 0 (.cpp:353-357)
  }
}

template <int DP, int G, class EV>
__device__ __forceinline__ double line_search_frame(const KgMcParams& P, const double* __restrict__ cst, double* __restrict__ scr,
                                                    EV& ev, double (&x)[DP], unsigned long long& n_val,
                                                    unsigned long long& n_grad) {
  typedef const volatile __attribute__((address_space(3))) double* cst_ptr;  // (volatile: read where used, never hoisted)
  cst_ptr C = (cst_ptr)cst;
  const int lane_id = (int)(threadIdx.x & 63u);
  const unsigned int free_mask = P.free_mask;
  const int max_num_steps = P.max_num_steps, max_num_restarts = P.max_num_restarts;
  const double tolerance = P.tolerance;
  if (max_num_restarts <= 0) {
    // reference returns without touching its outputs (.cpp:425-427): value 0, point filled with 1.0 (.cpp:163)
#pragma unroll
    for (int k = 0; k < DP; ++k) x[k] = (k < P.dim) ? 1.0 : 0.0;
    return 0.0;
  }
  const double step_tolerance = tolerance / (double)max_num_steps;
  // entry: table-row order, then into the frame (pinned rows -- fidelity coordinates, pads -- keep their frame value throughout)
  // The point lives twice: xo_l -- lane r holds coordinate r in the ORIGINAL units, advanced by exactly the reference's operations
  // (x += LimitUpdate(alpha grad), TensorProductDomain::LimitUpdate on the original bounds) -- and xf, its image in the frame,
  // which only feeds the evaluator.  LimitUpdate's "would the step leave the domain" test is a knife edge when a step goes to the
  // wall itself (max_relative_change = 1: x + (upper - x) is exactly upper in the original units, by Sterbenz, but can be one
  // ulp beyond the wall's image in the centred frame -- which halves the step): the decisions must be taken where the reference
  // takes them (found by tools/fuzz_parity.py, seed 101 case 93).
  double xf[DP];
  double xo_l = 0.0;
  {
    double xp[DP];
    to_table_order<DP, G>(x, P.perm, xp);
#pragma unroll
    for (int r = 0; r < DP; ++r) {
      xf[r] = (xp[r] - C[2 * DP + r]) * C[r];
      xo_l = (lane_id == r) ? xp[r] : xo_l;
    }
  }
  volatile __attribute__((address_space(3))) double* S = (volatile __attribute__((address_space(3))) double*)scr;
  const int lk = lane_id < DP ? lane_id : 0;
  const double lo_l = P.bounds[2 * lk], hi_l = P.bounds[2 * lk + 1];  // original units, table-row order (once per sample)
  const double s_l = C[lk];
  const bool free_l = lane_id < DP && ((free_mask >> (lane_id & 31)) & 1u);
  double fcur = 0.0;
  double gf[DP];
#pragma unroll
  for (int k = 0; k < DP; ++k) gf[k] = 0.0;
  // A clamped step needs f(x + step) before it is accepted (.hpp:771-786), and -- once accepted -- the next iteration starts by
  // evaluating f and grad f at that very point.  While another iteration can follow, that evaluation is therefore done as ONE
  // value + gradient pass and carried over (have_g): a step costs its Armijo trials + one pass instead of + two.  A carried
  // gradient that ends up unused (step rejected, |step| below the tolerance, no further restart) is counted as the value pass
  // it replaced, so the device counters never exceed what was computed.
  bool have_g = false;
  double f_carried = 0.0;
  int pred = 1;  // Armijo trials the previous step consumed (how many the next step evaluates in its first pass)
  for (int restart = 0; restart < max_num_restarts; ++restart) {
#pragma unroll
    for (int k = 0; k < DP; ++k) S[k] = xf[k];  // the restart's starting point (every lane writes the same value)
    for (int istep = 0; istep < max_num_steps;) {
      // ---- f(x), grad f(x) ----
      double f0;
      if (have_g) {
        f0 = f_carried;
        have_g = false;
      } else {
        f0 = ev.eval_grad_frame(xf, gf);
      }
      n_grad++;
      fcur = f0;
      // d2 = -2 g s (the trial direction in the frame, pre-multiplied for q2), x2 = -2 x', |g|^2
      double d2[DP], x2[DP];
      double norm = 0.0;
#pragma unroll
      for (int k = 0; k < DP; ++k) {
        const double sk = C[k];
        const double g = ((free_mask >> k) & 1u) ? gf[k] * sk : 0.0;  // fidelity / pad coordinates stay pinned
        norm = fma(g, g, norm);
        d2[k] = -2.0 * (g * sk);
        x2[k] = -2.0 * xf[k];
      }
      // pre_mult * (i+1)^-gamma (gpp_optimization.hpp:741); x^-0 == 1 exactly, so gamma == 0 needs no pow()
      double alpha_n = (P.gamma == 0.0) ? P.pre_mult : P.pre_mult * pow((double)(istep + 1), -P.gamma);
      // ---- Armijo back-tracking (.hpp:745-760): unclamped trial points ----
      // The trial step sizes alpha, alpha / 2, alpha / 4, ... are known in advance, so several of them can be evaluated in
      // one pass (EV::eval_value_multi, where the evaluator has it): as many as the previous step of THIS sample consumed
      // (the first step of a sample starts with one; a sample's results therefore depend on nothing but the sample), then
      // in pairs.  The sequence of decisions is the reference's; only consumed trials are counted.
      int search = 0;
      double ftrial = 0.0;
      {
        int batch = pred;
        bool done = false;
        while (!done) {
          bool evaluated = false;
          if (EV::kMaxTrials >= 2 && P.multi_trial != 0) {
            const int want = min(batch, 30 - search);
            if (want >= 2) evaluated = ev.armijo_batch(want, x2, d2, f0, norm, alpha_n, search, ftrial, done, n_val);
          }
          if (!evaluated) {
            double q2[DP];
#pragma unroll
            for (int k = 0; k < DP; ++k) q2[k] = fma(alpha_n, d2[k], x2[k]);
            ftrial = ev.eval_value_q2(q2);
            n_val++;
            if (ftrial - f0 > 0.5 * alpha_n * norm) {
              done = true;
            } else {
              alpha_n *= 0.5;
              if (++search >= 30) done = true;
            }
          }
          // (a first trial that fails is usually followed by several halvings: four more at once, then pairs)
          batch = (batch == 1 && search == 1) ? 4 : 2;
        }
        pred = min(search + 1, EV::kMaxTrials);
      }
      // ---- LimitUpdate in the frame, one coordinate per lane, then accept only if f improves (.hpp:762-795) ----
      bool changed, nonzero;
      double step[DP];
      double step_o = 0.0;  // this lane's coordinate of the step, original units
      {
        double g_l = 0.0, d2_l = 0.0;
#pragma unroll
        for (int k = 0; k < DP; ++k) {
          g_l = (lane_id == k) ? gf[k] : g_l;
          d2_l = (lane_id == k) ? d2[k] : d2_l;
        }
        const double want_o = alpha_n * (g_l * s_l);  // alpha grad_r in the original units (grad = frame gradient x scale)
        if (free_l) step_o = limit_update_1d(lo_l, hi_l, P.max_relative_change, xo_l, want_o);
        changed = __ballot(free_l && step_o != want_o) != 0ull;
        nonzero = __ballot(free_l && step_o != 0.0) != 0ull;
        // the step in the frame: the trial point's own offset where the clamp left it alone (its value is reused below)
        const double step_f = changed ? step_o * s_l : (-0.5 * alpha_n) * d2_l;
        if (lane_id < DP) S[2 * DP + lane_id] = free_l ? step_f : 0.0;  // back as wave-uniform values through the scratch
#pragma unroll
        for (int k = 0; k < DP; ++k) step[k] = S[2 * DP + k];
      }
      if (search == 30 || !nonzero) break;  // .hpp:781-785: x restored (a zero step re-evaluates f(x) == f0: rejected)
      double obj2 = ftrial;  // the clamp left the step untouched: f(x + step) is the last trial value
      bool carried = false;
      if (changed) {
        if (istep + 1 < max_num_steps || restart + 1 < max_num_restarts) {
          double xn[DP];
#pragma unroll
          for (int k = 0; k < DP; ++k) xn[k] = xf[k] + step[k];
          obj2 = ev.eval_grad_frame(xn, gf);  // (the old gradient is spent: d2 carries the direction)
          carried = true;
        } else {
          double q2[DP];
#pragma unroll
          for (int k = 0; k < DP; ++k) q2[k] = fma(-2.0, step[k], x2[k]);
          obj2 = ev.eval_value_q2(q2);
          n_val++;
        }
      }
      if (obj2 <= f0) {
        if (carried) n_val++;
        break;
      }
      double ss = 0.0;  // |step|^2 in the original coordinates
#pragma unroll
      for (int k = 0; k < DP; ++k) {
        xf[k] += step[k];
        const double so = step[k] * C[DP + k];
        ss = fma(so, so, ss);
      }
      xo_l += step_o;
      fcur = obj2;
      istep += 1;
      have_g = carried;
      f_carried = obj2;
      if (sqrt(ss) < step_tolerance) break;
    }
    double ds = 0.0;
#pragma unroll
    for (int k = 0; k < DP; ++k) {
      const double dk = (S[k] - xf[k]) * C[DP + k];
      ds = fma(dk, dk, ds);
    }
    if (!(sqrt(ds) > tolerance)) break;
  }
  if (have_g) n_val++;  // carried but never used
  // exit: the original-unit state (pinned rows hold what a fidelity coordinate / pad row holds) in the original dimension order
  {
    if (lane_id < DP) S[lane_id] = free_l ? xo_l : C[5 * DP + lk];
    double xo[DP];
#pragma unroll
    for (int r = 0; r < DP; ++r) xo[r] = S[r];
    from_table_order<DP, G>(xo, P.perm, x);
  }
  return fcur;
}

// SimplexIntersectTensorProductDomain::LimitUpdate's second half (gpp_domain.cpp:255-289) on the wave's LDS rows: xs / ss = the
// current point and the step AFTER the tensor-product limit (table-row order; pinned rows hold a zero step), walked in the reference's
// coordinate order j = 0 .. size - 1 through inv_perm.  If the proposed point leaves the unit simplex the step becomes
// relaxed * (step / norm) -- half the distance to the diagonal face along its own direction; returns whether it does.  Every lane
// computes the same scalars (uniform LDS reads).
__device__ __forceinline__ bool simplex_limit(const KgMcParams& P, const double* __restrict__ xs, const double* __restrict__ ss,
                                              double& relaxed, double& norm) {
  const int size = P.dim - P.f;
  // VectorNorm (gpp_linear_algebra.cpp:53-72): the scaled recurrence
  if (size == 1) {
    norm = fabs(ss[P.inv_perm[0]]);
  } else {
    double sc = 0.0, scaled = 1.0;
    for (int j = 0; j < size; ++j) {
      const double v = ss[P.inv_perm[j]];
      if (v != 0.0) {
        const double a = fabs(v);
        if (sc < a) {
          const double t = sc / a;
          scaled = 1.0 + scaled * (t * t);
          sc = a;
        } else {
          const double t = a / sc;
          scaled += t * t;
        }
      }
    }
    norm = sc * sqrt(scaled);
  }
  if (norm == 0.0) norm = 2.2250738585072014e-308;
  // CheckPointInUnitSimplex (gpp_geometry.hpp:313-325) on x + step
  bool inside = true;
  double sum = 0.0;
  for (int j = 0; j < size; ++j) {
    const int r = P.inv_perm[j];
    const double nx = xs[r] + ss[r];
    if (nx < 0.0) inside = false;
    sum += nx;
  }
  inside = inside && (sum - 4.0 * 2.220446049250313e-16) <= 1.0;
  relaxed = 0.0;
  if (inside) return false;
  // Plane::DistanceToPlaneAlongVector (gpp_geometry.hpp:252-272) for the plane sum x_i / sqrt(size) - 1 / sqrt(size) = 0
  double xn = 0.0, vn = 0.0;
  for (int j = 0; j < size; ++j) {
    const int r = P.inv_perm[j];
    xn += xs[r] * P.inv_sqrt_size;
    vn += (ss[r] / norm) * P.inv_sqrt_size;
  }
  const double numerator = P.inv_sqrt_size - xn;
  double dist = (vn == 0.0) ? (numerator == 0.0 ? 0.0 : INFINITY) : numerator / vn;
  if (dist < 0.0) dist = 0.0;
  relaxed = 0.5 * dist;  // kInvalidStepScaleFactor
  return true;
}

// The same line search with its wave-uniform vectors (x, grad, step, x at restart start) parked in an LDS scratch `st`
// (4 x kMaxDimPadded doubles, private to the wave) between evaluations, for kernels whose registers are better spent on
// point data (workgroup-per-sample variant: the evaluator's __syncthreads is an LDS fence, so nothing is cached across
// passes).  Identical arithmetic to line_search; uniform LDS reads are broadcasts that do not occupy the VALU.
template <int DP, int G, class EV>
__device__ __forceinline__ double line_search_lds(const KgMcParams& P, EV& ev, double* __restrict__ st, double (&x)[DP],
                                                  unsigned long long& n_val, unsigned long long& n_grad) {
  const double step_tolerance = P.tolerance / (double)P.max_num_steps;
  const int lane_id = (int)(threadIdx.x & 63u);
  const double lo_l = P.bounds[2 * (lane_id < DP ? lane_id : 0)], hi_l = P.bounds[2 * (lane_id < DP ? lane_id : 0) + 1];
  // frame centre and scale of this lane's row (behind the bounds in the blob): the conversions into the frame are done one row
  // per lane and handed over through the wave's LDS scratch -- as wave-uniform arithmetic on KgMcParams' arrays they kept ~50
  // SGPRs live across the whole line search, and the scalar file's spills (v_readlane / v_writelane) were this kernel's most
  // frequent instructions
  const bool free_l = lane_id < DP && ((P.free_mask >> (lane_id & 31)) & 1u);
  const double c_l = P.bounds[2 * kMaxDimPadded + (lane_id < DP ? lane_id : 0)];
  const double s_l = P.bounds[3 * kMaxDimPadded + (lane_id < DP ? lane_id : 0)];
  double* sX = st;
  double* sG = st + kMaxDimPadded;
  double* sS = st + 2 * kMaxDimPadded;
  double* sX0 = st + 3 * kMaxDimPadded;
  double* sF = st + 4 * kMaxDimPadded;  // x0 of the trial line, frame coordinates
  double* sD = st + 5 * kMaxDimPadded;  // its direction dv
  double fcur = 0.0;
  if (P.max_num_restarts <= 0) {
#pragma unroll
    for (int k = 0; k < DP; ++k) x[k] = (k < P.dim) ? 1.0 : 0.0;
    return 0.0;
  }
  double tq[DP], tqp[DP], gp[DP];
  to_table_order<DP, G>(x, P.perm, tq);  // table-row order from here on (see line_search)
#pragma unroll
  for (int k = 0; k < DP; ++k) sX[k] = tq[k];
#pragma unroll
  for (int k = 0; k < DP; ++k) gp[k] = 0.0;
  // Armijo trials the previous step of this sample consumed (>= 2): the size of the next step's first batch.  (A sample's first step:
  // two -- or six where a sweep is bound by the weight stream, the streamed-weights kernel: a bracket there is ~6 trials long.)
  int pred = (EV::kMaxTrials > 5) ? MOE_KG_STREAM_FIRST : 2;
#if MOE_BLOCK_PROF
  ev.seg_last = __builtin_amdgcn_s_memtime();
  ev.seg_tot = ev.c_tot;
#endif
  // A clamped step needs f(x + step) before it is accepted, and -- once accepted -- the next iteration starts with f and grad f at that
  // very point: while another iteration can follow, that evaluation is ONE value + gradient pass carried over (as in
  // line_search_frame: a step costs its Armijo sweeps + one pass instead of + two -- r3: with max_relative_change = 0.1 nearly every
  // step is clamped, so a sample saves five of its ~21 sweeps).  A carried gradient that ends up unused is counted as the value pass it replaced.
  bool have_g = false;
  double f_carried = 0.0, g_carried_l = 0.0;
  for (int restart = 0; restart < P.max_num_restarts; ++restart) {
    if (lane_id < DP) sX0[lane_id] = sX[lane_id];
    for (int istep = 0; istep < P.max_num_steps;) {
      if (lane_id < DP) sF[lane_id] = (sX[lane_id] - c_l) * s_l;  // the iterate in the frame: this pass's query AND the trial line's x0
#if MOE_BLOCK_PROF
      ev.seg_mark(3);  // (3: step end -> this gradient pass, loop control, restart bookkeeping)
#endif
      double g_l;
      double f0;
      if (have_g) {
        f0 = f_carried;
        g_l = g_carried_l;
        have_g = false;
      } else {
        f0 = ev.eval_p_lane(sF, s_l, g_l);
      }
      n_grad++;
      fcur = f0;
      if (lane_id < DP) sG[lane_id] = free_l ? g_l : 0.0;  // fidelity / pad coordinates stay pinned
      double norm = 0.0;
#pragma unroll
      for (int k = 0; k < DP; ++k) {
        const double gk = sG[k];
        norm = fma(gk, gk, norm);
      }
      double alpha_n = (P.gamma == 0.0) ? P.pre_mult : P.pre_mult * pow((double)(istep + 1), -P.gamma);
      int search = 0;
      double ftrial;
      // Armijo back-tracking, several trial step sizes per pass (alpha, alpha / 2, alpha / 4, ...: EV::armijo_batch): as many as
      // the previous step of this sample consumed (at least two -- in this kernel a pass is dominated by its fixed cost), then
      // in pairs.  The sequence of decisions is the reference's (gpp_optimization.hpp:752-769): a trial's value is only
      // looked at if all earlier ones failed, and only consumed trials are counted.  A trial that would need clamp_query
      // (millions of length scales away) sends the batch down the two-point path, which clamps.
      {
        // (x0 and dv go to the wave's LDS scratch and are read back inside each pass: as register arrays they stayed live
        //  across the passes of the bracket -- 48 VGPRs of wave-uniform data next to the register tiles -- and the kernel's
        //  scratch-memory spills sat exactly in this code between the passes)
        double dd = 0.0, q0 = 0.0, qa = 0.0;
        if (lane_id < DP) sD[lane_id] = sG[lane_id] * s_l;
#pragma unroll
        for (int r = 0; r < DP; ++r) {
          const double x0r = sF[r];
          const double dvr = sD[r];
          dd = fma(dvr, dvr, dd);
          q0 = fma(x0r, x0r, q0);
          const double xa = fma(alpha_n, dvr, x0r);
          qa = fma(xa, xa, qa);
        }
        // (|x0 + alpha dv|^2 is convex in alpha: the two ends bound every trial of the bracket)
        const bool near = uniform(fmax(q0, qa)) <= kQueryClamp * kQueryClamp;
        int batch = pred;
        bool done = false;
#if MOE_BLOCK_PROF
        ev.seg_mark(0);  // (0: gradient post-processing, norm, trial-line set-up)
#endif
        while (!done) {
          const int want = min(batch, 30 - search);
          if (near && want >= 2) {
            ev.armijo_batch(want, sF, sD, dd, f0, norm, alpha_n, search, ftrial, done, n_val);
          } else {
            const double a1 = alpha_n, a2 = 0.5 * alpha_n;
            double tqb[DP], f1, f2;
#pragma unroll
            for (int r = 0; r < DP; ++r) {
              tqp[r] = to_frame(P, fma(a1, sG[r], sX[r]), r);
              tqb[r] = to_frame(P, fma(a2, sG[r], sX[r]), r);
            }
            ev.eval2(tqp, tqb, f1, f2);
            ftrial = f1;
            n_val++;
            if (f1 - f0 > 0.5 * a1 * norm) break;
            alpha_n = a2;
            if (++search >= 30) break;
            if (want >= 2) {  // (want == 1: the 30th trial -- the second value is not consumed)
              ftrial = f2;
              n_val++;
              if (f2 - f0 > 0.5 * a2 * norm) break;
              alpha_n = 0.5 * a2;
              if (++search >= 30) break;
            }
          }
          // follow-up batches: pairs -- or fours where a sweep is bound by the weight stream, not by its trials (streamed-weights
          // kernel: kMaxTrials > 5), so that a bracket longer than predicted costs one more sweep, not two or three
          batch = (EV::kMaxTrials > 5) ? MOE_KG_STREAM_FOLLOWUP : 2;
        }
        pred = max(2, min(search + 1 + ((EV::kMaxTrials > 5) ? MOE_KG_STREAM_PRED_EXTRA : 0), EV::kMaxTrials));
#if MOE_BLOCK_PROF
        ev.seg_mark(1);  // (1: the Armijo loop outside its passes: dispatch, decisions)
#endif
      }
      // LimitUpdate with one coordinate per lane (the state already lives in per-wave LDS arrays, so lane k simply reads
      // entry k): one clamp instead of DP wave-uniform copies
      bool changed, nonzero;
      {
        const int lk = lane_id < DP ? lane_id : 0;
        const double want_l = alpha_n * sG[lk];
        double step_l = 0.0;
        if (free_l) step_l = limit_update_1d(lo_l, hi_l, P.max_relative_change, sX[lk], want_l);
        if (P.simplex != 0) {  // (the bounds are the box clipped to the unit hypercube, max_relative_change carries the reference's tweak)
          if (lane_id < DP) sS[lane_id] = step_l;
          __builtin_amdgcn_wave_barrier();
          double relaxed, vnorm;
          if (simplex_limit(P, sX, sS, relaxed, vnorm)) step_l = free_l ? relaxed * (step_l / vnorm) : 0.0;
          __builtin_amdgcn_wave_barrier();
        }
        changed = __ballot(free_l && step_l != want_l) != 0ull;
        nonzero = __ballot(free_l && step_l != 0.0) != 0ull;
        if (lane_id < DP) sS[lane_id] = step_l;  // read back below by every lane of this wave (in-order LDS, same wave)
      }
      if (search == 30 || !nonzero) break;
      double obj2 = ftrial;
      bool carried = false;
      double gn_l = 0.0;
      if (changed) {
        if (lane_id < DP) sF[lane_id] = ((sX[lane_id] + sS[lane_id]) - c_l) * s_l;
        if (MOE_KG_LDS_CARRY && (istep + 1 < P.max_num_steps || restart + 1 < P.max_num_restarts)) {
          obj2 = ev.eval_p_lane(sF, s_l, gn_l);
          carried = true;
        } else {
          obj2 = ev.template eval_p<false>(sF, gp);
          n_val++;
        }
      }
#if MOE_BLOCK_PROF
      ev.seg_mark(2);  // (2: LimitUpdate, clamped re-evaluation set-up)
#endif
      if (obj2 <= f0) {
        if (carried) n_val++;
        break;
      }
      // x += step one row per lane (as a wave-uniform loop the read-modify-writes of sX were twelve dependent LDS round trips:
      // most of the 2.8 k cycles per pass this kernel spent between its passes); |step|^2 from the step row alone, in k order
      if (lane_id < DP) sX[lane_id] = sX[lane_id] + sS[lane_id];
      double ss = 0.0;
#pragma unroll
      for (int k = 0; k < DP; ++k) {
        const double sk = sS[k];
        ss = fma(sk, sk, ss);
      }
      fcur = obj2;
      istep += 1;
      have_g = carried;
      f_carried = obj2;
      g_carried_l = gn_l;
      if (sqrt(ss) < step_tolerance) break;
    }
    double ds = 0.0;
#pragma unroll
    for (int k = 0; k < DP; ++k) {
      const double dk = sX0[k] - sX[k];
      ds = fma(dk, dk, ds);
    }
    if (!(sqrt(ds) > P.tolerance)) break;
  }
  if (have_g) n_val++;  // carried but never used
#pragma unroll
  for (int k = 0; k < DP; ++k) tq[k] = sX[k];
  from_table_order<DP, G>(tq, P.perm, x);
  return fcur;
}

// Evaluator of the wave-per-sample kernel for the WIDE padded dimensions (d = 17 .. 32), used with line_search_lds.
// At 24 / 32 coordinate rows the evaluator above needs x2, d2, x0, dv, the current and the prefetched tile -- 6 DP doubles, 384
// VGPRs at DP = 32 -- inside its tile loop and compiled to 2.3 KB of scratch per lane with the spills in the loop (r3: d = 32
// ran 15x slower than d = 16, d = 24 5x).  Here the line-search state lives in the wave's LDS scratch and a pass keeps only what
// its loop needs:
//   value passes (T trials along one line):   x0, dv (2 DP) + a ring of PF rows + T accumulators  -- 80 + T doubles at DP = 32
//   gradient pass:                            xq, grad sums (2 DP) + the tile's rows (DP)          -- 96 doubles at DP = 32; the
//     gradient is accumulated as sum_j coef_j x_j - q sum_j coef_j (the centred frame keeps both terms of the size of the
//     gradient: see eval_loop GDOT), so a row is dead -- and refilled from the next tile -- right after its fma.
// The table of the wide dimensions holds the rows in PAIRS, [tile][DP / 2][64][2]: a lane fetches rows 2i, 2i + 1 of its point
// with one 16-byte load (8-byte loads run at 0.54 - 0.70x the 16-byte rate out of L2, and these passes are bound by exactly
// that stream: ~200 KB per sweep and wave at n = 1000, d = 24).  The first `ntl` tiles are served from an LDS copy shared by
// the workgroup's waves (whatever the weight slabs leave of the 160 KB), the rest from L2; a sweep is the same loop over the
// two segments (the ring's slot indices are static: PF divides DP).
// Same arithmetic as BlockEval (direct differences about x0); queries beyond kQueryClamp are pulled back (clamp_query).
typedef double d2t __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) d2t* lds_pair_ptr;

// Where a sweep takes the 1 + G weights of a lane's point from: the wave's LDS slab [tile][1 + G][64] (+ lane), or the sample's row of
// the weight table in global memory, [point][1 + G] (+ lane (1 + G)) -- the rows of a point travel as 16-byte pairs when there is an
// even number of them.
template <int G, bool WS>
struct WeightPtr {
  lds_tile_ptr p;
  __device__ __forceinline__ void load(double (&cw)[1 + G]) const {
#pragma unroll
    for (int a = 0; a < 1 + G; ++a) cw[a] = p[a * 64];
  }
  __device__ __forceinline__ void next() { p += (1 + G) * 64; }
};
template <int G>
struct WeightPtr<G, true> {
  const double* __restrict__ p;
  __device__ __forceinline__ void load(double (&cw)[1 + G]) const {
    if constexpr (((1 + G) & 1) == 0) {
      const d2t* q = reinterpret_cast<const d2t*>(p);
#pragma unroll
      for (int a2 = 0; a2 < (1 + G) / 2; ++a2) {
        const d2t v = q[a2];
        cw[2 * a2] = v.x;
        cw[2 * a2 + ((1 + G) > 1 ? 1 : 0)] = v.y;
      }
    } else {
#pragma unroll
      for (int a = 0; a < 1 + G; ++a) cw[a] = p[a];
    }
  }
  __device__ __forceinline__ void next() { p += 64 * (1 + G); }
};

template <int DP, int G, bool WS = false>
struct WideEval {
  const d2t* __restrict__ xg;   // coordinate table of this evaluation in global memory (one tile of padding behind), + lane
  lds_pair_ptr xl;              // LDS copy of its first `ntl` tiles, + lane
  WeightPtr<G, WS> w0;              // this sample's weights, first tile (see WeightPtr)
  const double* __restrict__ etab;
  double* __restrict__ red;         // kPartLen doubles of LDS, private to the wave: packed sums of a gradient pass
  int ntiles, ntl, cov_type, lane;
  double mean;
  // Armijo trials per sweep: with the weights streamed from the table a sweep is bound by that stream (64 KB per sample and sweep at
  // C5), so a step's whole bracket goes into ONE sweep whenever the previous step's bracket predicts up to eight trials.
  // (the L2-streamed sweeps of 16 / 24 rows gain too -- n = 1000: d = 16 0.257 -> 0.230 ms, d = 24 0.306 -> 0.296; at 32 rows the eight
  //  accumulators cost more registers than the saved sweeps give back: 0.403 -> 0.429 -- r3, `profiles/r03_dim_sweep_after.txt`)
  static constexpr int kMaxTrials = (WS || DP <= 24) ? MOE_KG_STREAM_TRIALS : 5;
  static constexpr int HP = DP / 2;                        // row pairs per tile
  static constexpr int PF2 = (HP <= 8) ? HP : ((HP % 8 == 0) ? 8 : 6);  // ring depth in row pairs
  static_assert(HP % PF2 == 0 && PF2 <= HP, "ring slots must be static");
#if MOE_BLOCK_PROF
  unsigned long long c_tot = 0, seg_last = 0, seg_tot = 0;
  __device__ __forceinline__ void seg_mark(int) {}
#endif

  // one segment (LDS or L2 tiles) of a T-trial value sweep; `wt` walks on through the weight slab
  template <int COV, int T, class PP>
  __device__ __forceinline__ void segT(PP xt, WeightPtr<G, WS>& wt, int nt, const double (&x0)[DP], const double (&dv)[DP],
                                       const double (&al)[T], double dd, double (&acc)[T]) {
    if (nt <= 0) return;
    d2t ring[PF2];
    double cw[1 + G];
#pragma unroll
    for (int i = 0; i < PF2; ++i) ring[i] = xt[i * 64];
    wt.load(cw);
#pragma unroll 1
    for (int tile = 0; tile < nt; ++tile) {
      double A = 1.0e-300, B = 0.0, sdA = 0.0, sdB = 0.0;
#pragma unroll
      for (int i = 0; i < HP; ++i) {
        const d2t c = ring[i % PF2];
        ring[i % PF2] = xt[(i + PF2) * 64];  // pair i + PF2 of the stream (runs into the next tile; behind the segment: unused)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int k = 2 * i + h;
          const double d0 = (h == 0 ? c.x : c.y) - x0[k];
          A = fma(d0, d0, A);
          B = fma(d0, dv[k], B);
          if (G > 0 && k < G) {
            sdA = fma(cw[1 + (k < G ? k : 0)], d0, sdA);
            sdB = fma(cw[1 + (k < G ? k : 0)], dv[k], sdB);
          }
        }
      }
      xt += HP * 64;
      wt.next();
      double nw[1 + G];
      wt.load(nw);  // (the last tile's prefetch reads one tile past the sample's weights: padded / the next sample's, unused)
      const double mB2 = -2.0 * B;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const double r2 = fmax(fma(al[t], fma(al[t], dd, mB2), A), 1.0e-300);
        double base, first, second;
        radial3<COV, (G > 0), false>(r2, etab, base, first, second);
        acc[t] = fma(cw[0], base, acc[t]);
        if (G > 0) acc[t] = fma(first, fma(-al[t], sdB, sdA), acc[t]);
      }
#pragma unroll
      for (int a = 0; a < 1 + G; ++a) cw[a] = nw[a];
    }
  }

  template <int T>
  __device__ __forceinline__ void values(const double (&x0)[DP], const double (&dv)[DP], const double (&al)[T], double dd,
                                         double (&f)[T]) {
    double acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = 0.0;
    WeightPtr<G, WS> wt = w0;
    if (cov_type == MOE_COV_SQUARE_EXPONENTIAL) {
      segT<MOE_COV_SQUARE_EXPONENTIAL, T>(xl, wt, ntl, x0, dv, al, dd, acc);
      segT<MOE_COV_SQUARE_EXPONENTIAL, T>(xg + (long)ntl * HP * 64, wt, ntiles - ntl, x0, dv, al, dd, acc);
    } else {
      segT<MOE_COV_MATERN_NU_2P5, T>(xl, wt, ntl, x0, dv, al, dd, acc);
      segT<MOE_COV_MATERN_NU_2P5, T>(xg + (long)ntl * HP * 64, wt, ntiles - ntl, x0, dv, al, dd, acc);
    }
    double sum[T];
    constexpr int T4 = T / 4 * 4;
#pragma unroll
    for (int t = 0; t < T4; t += 4) {
      double o[4];
      wave_sum4_uniform(acc[t], acc[t + 1], acc[t + 2], acc[t + 3], o);
      sum[t] = o[0];
      sum[t + 1] = o[1];
      sum[t + 2] = o[2];
      sum[t + 3] = o[3];
    }
    if constexpr (T - T4 >= 2) wave_sum2_uniform(acc[T4], acc[T4 + 1], sum[T4], sum[T4 + 1]);
    if constexpr (((T - T4) & 1) != 0) sum[T - 1] = wave_sum_uniform(acc[T - 1]);
#pragma unroll
    for (int t = 0; t < T; ++t) f[t] = -(mean + sum[t]);
  }

  // up to kMaxTrials Armijo trials x0 + alpha 2^-t dv in one pass and the reference's decisions over them (see BlockEval::armijo_t)
  template <int T>
  __device__ __forceinline__ void armijo_t(const double* __restrict__ x0p, const double* __restrict__ dvp, double dd, double f0,
                                           double norm, double& alpha_n, int& search, double& ftrial, bool& done,
                                           unsigned long long& n_val) {
    double x0[DP], dv[DP];
#pragma unroll
    for (int k = 0; k < DP; ++k) {
      x0[k] = x0p[k];
      dv[k] = dvp[k];
    }
    double al[T], f[T];
#pragma unroll
    for (int t = 0; t < T; ++t) al[t] = (t == 0) ? alpha_n : 0.5 * al[t > 0 ? t - 1 : 0];
    values<T>(x0, dv, al, dd, f);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if (!done) {
        ftrial = f[t];
        n_val++;
        if (ftrial - f0 > 0.5 * alpha_n * norm) {
          done = true;
        } else {
          alpha_n *= 0.5;
          if (++search >= 30) done = true;
        }
      }
    }
  }
  __device__ __forceinline__ void armijo_batch(int want, const double* __restrict__ x0, const double* __restrict__ dv, double dd,
                                               double f0, double norm, double& alpha_n, int& search, double& ftrial, bool& done,
                                               unsigned long long& n_val) {
    switch (want) {
      case 2: armijo_t<2>(x0, dv, dd, f0, norm, alpha_n, search, ftrial, done, n_val); break;
      case 3: armijo_t<3>(x0, dv, dd, f0, norm, alpha_n, search, ftrial, done, n_val); break;
      case 4: armijo_t<4>(x0, dv, dd, f0, norm, alpha_n, search, ftrial, done, n_val); break;
      case 5: armijo_t<5>(x0, dv, dd, f0, norm, alpha_n, search, ftrial, done, n_val); break;
      default:
        if constexpr (kMaxTrials > 5) {
          switch (want) {
            case 6: armijo_t<6>(x0, dv, dd, f0, norm, alpha_n, search, ftrial, done, n_val); break;
            case 7: armijo_t<7>(x0, dv, dd, f0, norm, alpha_n, search, ftrial, done, n_val); break;
            default: armijo_t<8>(x0, dv, dd, f0, norm, alpha_n, search, ftrial, done, n_val); break;
          }
        } else {
          armijo_t<5>(x0, dv, dd, f0, norm, alpha_n, search, ftrial, done, n_val);
        }
        break;
    }
  }

  // f at one point: a one-trial sweep along the zero direction
  __device__ __forceinline__ double value_at(const double (&xq_in)[DP]) {
    double xq[DP], dv[DP];
#pragma unroll
    for (int k = 0; k < DP; ++k) {
      xq[k] = xq_in[k];
      dv[k] = 0.0;
    }
    clamp_query<DP>(xq);
    const double al[1] = {0.0};
    double f[1];
    values<1>(xq, dv, al, 0.0, f);
    return f[0];
  }
  __device__ __forceinline__ void eval2(const double (&xa)[DP], const double (&xb)[DP], double& fa, double& fb) {
    fa = value_at(xa);
    fb = value_at(xb);
  }
  template <bool WG>
  __device__ __forceinline__ double eval_p(const double* __restrict__ xq_ptr, double (&)[DP]) {
    static_assert(!WG, "gradient passes of the wide evaluator go through eval_p_lane");
    double xq[DP];
#pragma unroll
    for (int k = 0; k < DP; ++k) xq[k] = xq_ptr[k];
    return value_at(xq);
  }

  // one segment of a gradient sweep
  template <int COV, class PP>
  __device__ __forceinline__ void segG(PP xt, WeightPtr<G, WS>& wt, int nt, const double (&xq)[DP], double& accf, double& accs,
                                       double (&accg)[DP], double (&accd)[G > 0 ? G : 1]) {
    if (nt <= 0) return;
    d2t cx[HP];
    double cw[1 + G];
#pragma unroll
    for (int i = 0; i < HP; ++i) cx[i] = xt[i * 64];
    wt.load(cw);
#pragma unroll 1
    for (int tile = 0; tile < nt; ++tile) {
      double r2 = 1.0e-300, sd = 0.0;
#pragma unroll
      for (int k = 0; k < DP; ++k) {
        const double d0 = ((k & 1) ? cx[k / 2].y : cx[k / 2].x) - xq[k];
        r2 = fma(d0, d0, r2);
        if (G > 0 && k < G) sd = fma(cw[1 + (k < G ? k : 0)], d0, sd);
      }
      double base, first, second;
      radial3<COV, true, (G > 0)>(r2, etab, base, first, second);
      const double w0 = cw[0];
      accf = fma(w0, base, accf);
      double coef = w0 * first;
      if (G > 0) {
        accf = fma(first, sd, accf);
        coef = fma(second, sd, coef);
#pragma unroll
        for (int a = 0; a < G; ++a) accd[a] = fma(first, cw[1 + a], accd[a]);
      }
      accs += coef;
      xt += HP * 64;
      wt.next();
#pragma unroll
      for (int i = 0; i < HP; ++i) {
        accg[2 * i] = fma(coef, cx[i].x, accg[2 * i]);
        accg[2 * i + 1] = fma(coef, cx[i].y, accg[2 * i + 1]);
        cx[i] = xt[i * 64];  // the pair's slot is free: the next tile's rows (behind the segment: unused)
      }
      wt.load(cw);
    }
  }

  // value + gradient, the gradient handed back one component per lane in the original units (scale_l = this lane's frame scale)
  __device__ __forceinline__ double eval_p_lane(const double* __restrict__ xq_ptr, double scale_l, double& grad_l) {
    constexpr int NS = 2 + DP + (G > 0 ? G : 0);
    static_assert(NS <= kPartLen, "packed sums of a gradient pass exceed the wave's scratch");
    double xq[DP];
#pragma unroll
    for (int k = 0; k < DP; ++k) xq[k] = xq_ptr[k];
    clamp_query<DP>(xq);
    double q_l = 0.0;  // this lane's (clamped) query coordinate
#pragma unroll
    for (int k = 0; k < DP; ++k)
      if (lane == k) q_l = xq[k];
    double accf = 0.0, accs = 0.0, accg[DP], accd[G > 0 ? G : 1];
#pragma unroll
    for (int k = 0; k < DP; ++k) accg[k] = 0.0;
#pragma unroll
    for (int a = 0; a < (G > 0 ? G : 1); ++a) accd[a] = 0.0;
    WeightPtr<G, WS> wt = w0;
    if (cov_type == MOE_COV_SQUARE_EXPONENTIAL) {
      segG<MOE_COV_SQUARE_EXPONENTIAL>(xl, wt, ntl, xq, accf, accs, accg, accd);
      segG<MOE_COV_SQUARE_EXPONENTIAL>(xg + (long)ntl * HP * 64, wt, ntiles - ntl, xq, accf, accs, accg, accd);
    } else {
      segG<MOE_COV_MATERN_NU_2P5>(xl, wt, ntl, xq, accf, accs, accg, accd);
      segG<MOE_COV_MATERN_NU_2P5>(xg + (long)ntl * HP * 64, wt, ntiles - ntl, xq, accf, accs, accg, accd);
    }
    double sums[NS];
    sums[0] = accf;
#pragma unroll
    for (int k = 0; k < DP; ++k) sums[1 + k] = accg[k];
    sums[1 + DP] = accs;
    if (G > 0) {
#pragma unroll
      for (int a = 0; a < G; ++a) sums[2 + DP + a] = accd[a];
    }
    wave_sum_packed_store<NS>(sums, red, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    volatile __attribute__((address_space(3))) double* R = (volatile __attribute__((address_space(3))) double*)red;
    const double f = R[0];
    const double ss = R[1 + DP];
    double v = R[1 + (lane < DP ? lane : 0)];
    v = fma(-q_l, ss, v);  // sum coef x_k - q_k sum coef
    if (G > 0) {
      const double vd = R[2 + DP + (lane < G ? lane : 0)];
      if (lane < G) v -= vd;
    }
    grad_l = -(v * scale_l);
    __builtin_amdgcn_wave_barrier();  // (the next pass's sums overwrite the slot only after every lane has read it: in-order LDS)
    return -(mean + uniform(f));
  }
};

// z_i (antithetic pairs, .cpp:171-180) and beta = L^-T z for global sample index s: lane c owns component c; copies go to
// the LDS scratch zb ([0, kMaxM) = z, [kMaxM, 2 kMaxM) = beta) for the uniform reads of the weight loop and the scan.
__device__ __forceinline__ void draw_z_beta(const KgMcParams& P, const double* __restrict__ Lsm, int s, int lane,
                                            double* __restrict__ zb, double& zc, double& bc) {
  const int m = P.m;
  const double sign = (s & 1) ? -1.0 : 1.0;
  zc = 0.0;
  if (lane < m) zc = sign * P.normals[(long)(s >> 1) * m + lane];
  // Back substitution L^T beta = z, column-oriented: once beta_r is final it is broadcast and every lane l < r folds
  // L[r][l] beta_r into its running sum -- no wave reduction per step, and row r of L does not depend on the recurrence, so
  // eight rows are requested at a time (one round trip to L2 per eight steps instead of one per step).
  bc = 0.0;
  double acc = 0.0;
  for (int r0 = m - 1; r0 >= 0; r0 -= 8) {
    double Lrow[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      // unconditional loads from clamped addresses (a predicated load becomes a branch, and the eight loads would be
      // waited for one by one); entries of lanes above the diagonal are never used
      Lrow[i] = Lsm[max(r0 - i, 0) + min(lane, m - 1) * m];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = r0 - i;
      if (r >= 0) {
        if (lane == r) bc = (zc - acc) / Lrow[i];
        const int blo = __builtin_amdgcn_readlane(__double2loint(bc), r);
        const int bhi = __builtin_amdgcn_readlane(__double2hiint(bc), r);
        const double br = __hiloint2double(bhi, blo);
        if (lane < r) acc = fma(Lrow[i], br, acc);
      }
    }
  }
  zb[lane] = zc;  // kMaxM == 64 == wavefront size
  zb[kMaxM + lane] = bc;
  __builtin_amdgcn_wave_barrier();  // keep later cross-lane LDS reads after these writes
}

// Discretised-set scan (.cpp:436-449): f_j = -(mu_n(x_j) + c_j . z); returns the index of the FIRST best point.
__device__ __forceinline__ int discrete_scan(const KgMcParams& P, const double* __restrict__ rec, const double* __restrict__ zb,
                                             int lane) {
  const int m = P.m;
  const double* mu_disc = rec + P.rec.mu_disc;
  const double* C_disc = rec + P.rec.C_disc;
  double best_f = -INFINITY;
  int best_j = 0;
  for (int j0 = 0; j0 < P.A; j0 += 64) {
    const int j = j0 + lane;
    double fj = -INFINITY;
    if (j < P.A) {
      double v = mu_disc[j];
      const double* Cj = C_disc + (long)j * m;
      for (int c0 = 0; c0 < m; c0 += 8) {  // eight loads in flight, folded in the same order as one at a time
        double cv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) cv[i] = Cj[min(c0 + i, m - 1)];
#pragma unroll
        for (int i = 0; i < 8; ++i) v = fma(cv[i], (c0 + i < m) ? zb[min(c0 + i, kMaxM - 1)] : 0.0, v);
      }
      fj = -v;
    }
    double wmax = fj;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wmax = fmax(wmax, __shfl_xor(wmax, off, 64));
    const unsigned long long ballot = __ballot(fj == wmax);
    const int first_lane = __ffsll((long long)ballot) - 1;
    if (wmax > best_f) {  // strict: an earlier chunk wins ties (priority-queue semantics of .cpp:440-447)
      best_f = wmax;
      best_j = j0 + first_lane;
    }
  }
  return __builtin_amdgcn_readfirstlane(best_j);
}

// One MC sample: weights, discretised-set scan, line-search gradient descent.  Called with the whole wave converged.
template <int DP, int G, bool SMALL, bool XL>
__device__ __forceinline__ void kg_sample(const KgMcParams& P, int e, int sl, const double* __restrict__ xs,
                                          double* __restrict__ aw, double* __restrict__ zb,
                                          const double* __restrict__ etab, const double* __restrict__ cst, int lane,
                                          unsigned long long& tot_val, unsigned long long& tot_grad) {
  const int m = P.m, u = P.u, n = P.n, g1 = 1 + P.g;
  const int s = P.first_sample + sl;  // global sample index
  const int size = P.dim - P.f;       // problem size of the inner optimisation
  const double* rec = P.blob + (long)e * P.rec.stride;
  const double* Lsm = rec + P.rec.L;
  const double* We = P.W + (long)e * P.w_stride;

  double zc = 0.0, bc = 0.0;
  MOE_PROF_T(w0);
  const long so0 = (long)e * P.num_local + sl;
  const bool prepped = P.best_j != nullptr;  // beta and the discretised-set winner come from the pre-pass (kg_sample_prep_kernel)
  int best_j = 0;
  if (prepped) {
    bc = (lane < m) ? P.beta[so0 * m + lane] : 0.0;
    best_j = P.best_j[so0];
    zb[kMaxM + lane] = bc;
    __builtin_amdgcn_wave_barrier();
  } else {
    draw_z_beta(P, Lsm, s, lane, zb, zc, bc);
  }
  MOE_PROF_T(w1);
  // ---- per-sample weights (see file header) into this wave's LDS slab ----
  // v(j,a) = KinvY[(j,a)] - sum_c W[(j,a), c] beta_c.
  if (G == 0 && g1 == 1 && m <= 4) {
    // q-KG with up to four fantasy points: the loads of EIGHT tiles (40 independent L2 reads per lane) are issued together, so a
    // sample pays two round trips to L2 instead of one per pair of tiles (the phase is pure latency: 33k of a sample's 316k
    // cycles before).  Same operation order as the general loop below.
    const double b0 = zb[kMaxM], b1 = zb[kMaxM + 1], b2 = zb[kMaxM + 2], b3 = zb[kMaxM + 3];  // 0 beyond m
    const long c1 = (long)min(1, m - 1) * P.N, c2 = (long)min(2, m - 1) * P.N, c3 = (long)min(3, m - 1) * P.N;
    for (int t0 = 0; t0 < P.ntiles; t0 += 8) {
      double ky[8], l0[8], l1[8], l2[8], l3[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const long row = min(min(t0 + i, P.ntiles - 1) * 64 + lane, n - 1);  // clamped: always a valid address
        ky[i] = P.KinvY[row];
        l0[i] = We[row];
        l1[i] = We[row + c1];
        l2[i] = We[row + c2];
        l3[i] = We[row + c3];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int t = t0 + i;
        if (t < P.ntiles) {
          const int j = t * 64 + lane;
          double v = ky[i];
          v = fma(-l0[i], b0, v);
          v = fma(-l1[i], b1, v);
          v = fma(-l2[i], b2, v);
          v = fma(-l3[i], b3, v);
          if (j >= n) v = (j < n + u) ? zb[kMaxM + min(j - n, kMaxM - 1)] : 0.0;
          aw[(long)t * 64 + lane] = v * P.alpha;
        }
      }
    }
  } else {
  // The W loads are issued four columns at a time (column index clamped
  // to m - 1; beta is 0 beyond m) and two tiles per iteration, so ~10 independent L2 loads are in flight per wait
  // instead of one dependent load per fma.
#pragma unroll 2
  for (int t = 0; t < P.ntiles; ++t) {
    const int j = t * 64 + lane;
    double* w = aw + (long)t * (1 + G) * 64 + lane;
#pragma unroll
    for (int a = 0; a < 1 + G; ++a) {
      double v = 0.0;
      if (a < g1) {
        if (j < n) {
          const long row = (long)j * g1 + a;
          v = P.KinvY[row];
          for (int c0 = 0; c0 < m; c0 += 4) {
            const double l0 = We[row + (long)c0 * P.N];
            const double l1 = We[row + (long)min(c0 + 1, m - 1) * P.N];
            const double l2 = We[row + (long)min(c0 + 2, m - 1) * P.N];
            const double l3 = We[row + (long)min(c0 + 3, m - 1) * P.N];
            v = fma(-l0, zb[kMaxM + c0], v);
            v = fma(-l1, zb[kMaxM + min(c0 + 1, kMaxM - 1)], v);
            v = fma(-l2, zb[kMaxM + min(c0 + 2, kMaxM - 1)], v);
            v = fma(-l3, zb[kMaxM + min(c0 + 3, kMaxM - 1)], v);
          }
        } else if (j < n + u) {
          v = zb[kMaxM + (j - n) * g1 + a];
        }
        // fold alpha and, for derivative weights, the -(frame scale) of the derivative row (radial3)
        v *= (a == 0) ? P.alpha : -P.alpha * cst[a > 0 ? a - 1 : 0];
      }
      w[a * 64] = v;
    }
  }
  }
  // (each lane only ever reads back the weight entries it wrote itself -- no cross-lane hazard; zb is read by every lane
  //  but was written before the wave-wide butterflies above/below execute, and LDS ops of one wave complete in order)

  MOE_PROF_T(w2);
  if (!prepped) best_j = discrete_scan(P, rec, zb, lane);
  MOE_PROF_T(w3);

  const double* disc = rec + P.rec.disc;
  double x[DP];
#pragma unroll
  for (int k = 0; k < DP; ++k) x[k] = (k < size) ? disc[(long)best_j * size + k] : ((k < P.dim) ? 1.0 : 0.0);

  unsigned long long n_val = 0, n_grad = 0;  // passes over the n + u points (the A-point scan is O(A m), not counted)
  double fcur;
  if constexpr (wide_eval(DP, XL)) {
    double* st = zb + 2 * kMaxM;  // line-search vectors | packed-sum slot: kWideScratch doubles behind the z / beta scratch
    const d2t* xg = reinterpret_cast<const d2t*>(xs) + lane;
    lds_pair_ptr xl = (lds_pair_ptr)(cst + kCstRows * DP) + lane;  // the workgroup's LDS copy of the first tiles (kg_mc_kernel)
    WideEval<DP, G> ev{xg, xl, {(lds_tile_ptr)(aw + lane)}, etab, st + kLsRows * kMaxDimPadded, P.ntiles, P.wide_lds_tiles, P.cov_type, lane, P.mean};
    fcur = line_search_lds<DP, G>(P, ev, st, x, n_val, n_grad);
  } else {
    WaveEval<DP, G, SMALL, XL> ev{xs, aw, etab, P.ntiles, P.cov_type, P.mean, P.inv_lp, lane, zb + DP};
    fcur = line_search_frame<DP, G>(P, cst, zb, ev, x, n_val, n_grad);  // (z / beta scratch is idle by now)
#if MOE_BLOCK_PROF
    {
      const unsigned long long w4 = __builtin_amdgcn_s_memtime();
      if (lane == 0) {  // [0] z/beta [1] weights [2] scan [3] line search | [4] in value passes [5] in gradient passes | counts
        atomicAdd(&P.prof[0], w1 - w0);
        atomicAdd(&P.prof[1], w2 - w1);
        atomicAdd(&P.prof[2], w3 - w2);
        atomicAdd(&P.prof[3], w4 - w3);
        atomicAdd(&P.prof[4], ev.c_v);
        atomicAdd(&P.prof[5], ev.c_g);
        atomicAdd(&P.prof[6], ev.n_v);
        atomicAdd(&P.prof[7], ev.n_g);
        atomicAdd(&P.prof[13], 1ull);
      }
    }
#endif
  }

  const long so = so0;
  if (lane == 0) P.best_value[so] = fcur;
  tot_val += n_val;  // flushed once per wave and evaluation by the caller: 10^4 samples x 2 atomics on one cache line per
  tot_grad += n_grad;  // evaluation serialise in the L2 atomic unit (and delay the sample tickets queued behind them)
  if (lane < DP) {
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < DP; ++k)
      if (lane == k) v = x[k];
    P.best_point[so * DP + lane] = v;
  }
  if (!prepped && lane < m) P.beta[so * m + lane] = bc;
}

// SMALL: the many-wavefront instantiation for point sets of a few tiles, where one pass is a short dependent chain (LDS
// read -> ~43 dependent FP64 operations -> wave reduction -> decision) that two wavefronts per SIMD cannot cover: compiled
// for 16 wavefronts per workgroup (<= 128 VGPRs, tile loop not unrolled) so that four wavefronts share each SIMD.
template <int DP, int G, bool XLDS, bool SMALL>
struct kg_mc_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const KgMcParams& P) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int ntiles = P.ntiles;
    const int tab = ntiles * (DP + 1) * 64;  // LDS copy: the DP coordinate rows + the |x|^2 row per tile (see eval_loop)
    const int wslab = ntiles * (1 + G) * 64 + 2 * kMaxM + (wide_eval(DP, XLDS) ? kWideScratch : 0);
    // LDS: [64] exp table | [kCstRows x DP] per-row constants of the frame line search | [tab] coordinates (if XLDS) | per-wave slabs
    double* cst = smem + kExpTabLen;
    double* coords = cst + kCstRows * DP;
    const int wtab = wide_eval(DP, XLDS) ? P.wide_lds_tiles * DP * 64 : 0;  // streamed table: its leading tiles (paired rows: WideEval)
    double* aw = coords + (XLDS ? tab : wtab) + wave * wslab;
    double* zb = aw + ntiles * (1 + G) * 64;
    if (threadIdx.x < kExpTabLen) smem[threadIdx.x] = kExp2Tab64[threadIdx.x];
    fill_frame_constants<DP>(P, cst);
    if (!XLDS) __syncthreads();
    // evaluation of this workgroup: workgroups b, b + E, b + 2E, ... serve evaluation b mod E (b mod 8 is also the XCD, so
    // with E = 8 each evaluation's W / table stay in one XCD's L2); with fewer workgroups than evaluations they loop.
    for (int e = blockIdx.x % P.E; e < P.E; e += (gridDim.x < (unsigned)P.E ? gridDim.x : P.E)) {
      const double* xs = P.XsTab + (long)e * P.tab_stride;
      if (XLDS) {
        __syncthreads();  // previous evaluation's readers are done
        for (int pt = threadIdx.x; pt < ntiles * 64; pt += blockDim.x) {  // one point per thread and step
          const int tl = pt >> 6, l = pt & 63;
          const double* src = xs + (long)tl * DP * 64 + l;
          double* dst = coords + tl * (DP + 1) * 64 + l;
          double xx = 0.0;
  #pragma unroll
          for (int k = 0; k < DP; ++k) {
            const double v = src[k * 64];
            dst[k * 64] = v;
            xx = fma(v, v, xx);
          }
          dst[DP * 64] = xx;
        }
        xs = coords;
        __syncthreads();
      }
      if constexpr (wide_eval(DP, XLDS)) {
        __syncthreads();  // previous evaluation's readers are done
        const d2t* src = reinterpret_cast<const d2t*>(xs);
        d2t* dst = reinterpret_cast<d2t*>(coords);
        for (int i = threadIdx.x; i < wtab / 2; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
      }
      // sample tickets are drawn ONE AHEAD: the atomic's round trip to L2 overlaps the current sample instead of stalling
      // the wave between samples (each wave ends up drawing one ticket it does not use)
      unsigned int ticket = 0;
      unsigned int* next = P.next_sample + (long)e * kTicketStride;  // one cache line per evaluation
      if (lane == 0) ticket = atomicAdd(next, 1u);
      unsigned long long tot_val = 0, tot_grad = 0;
      while (true) {
        const unsigned int sl = (unsigned int)__builtin_amdgcn_readfirstlane((int)ticket);
        if (sl >= (unsigned int)P.num_local) break;
        if (lane == 0) ticket = atomicAdd(next, 1u);
        kg_sample<DP, G, SMALL, XLDS>(P, e, (int)sl, xs, aw, zb, smem, cst, lane, tot_val, tot_grad);
      }
      if (lane == 0 && (tot_val | tot_grad) != 0) {
        atomicAdd(&P.counters[2 * e], tot_val);
        atomicAdd(&P.counters[2 * e + 1], tot_grad);
      }
      if (gridDim.x >= (unsigned)P.E) break;
    }
  }
};
template <int DP, int G, bool XLDS, bool SMALL>
__global__ __launch_bounds__(SMALL ? 1024 : 512) void kg_mc_kernel(KgMcParams P) {
  kg_mc_kernel_body<DP, G, XLDS, SMALL>::run(MOE_VBLOCK, MOE_VGRID, nullptr, P);
}

// =====================================================================================================================
// Streamed-weights variant of the wave-per-sample kernel (r3), for point sets whose per-sample weights are too large for eight
// LDS slabs (d-KG at n = 2000 with 3 observed derivatives: 64 KB per sample).  The sample pre-pass (kg_sample_prep_kernel) and the
// weight table V (kg_sample_weights_kernel: every sample's weights, the fantasy points' and the zero padding included) exist
// already for the workgroup-per-sample kernel; here a WAVE owns a sample and reads its row of V inside the sweeps (16-byte loads, two per
// point at g = 3; a sample's 64 KB are re-read once per sweep -- eight waves x 256 CUs keep 131 MB of rows live, which the
// Infinity Cache holds), so the 160 KB of LDS serve the COORDINATES, shared by the workgroup's eight waves (22 of C5's 32
// tiles; the rest streams from L2).  No barrier, no lock-step line search: the passes are WideEval's.
// LDS: [64] exp table | leading tiles of the paired-row table | per wave: kWideScratch doubles.
// =====================================================================================================================
template <int DP, int G>
struct kg_mc_stream_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const KgMcParams& P) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wtab = P.wide_lds_tiles * DP * 64;
    double* coords = smem + kExpTabLen;
    double* st = coords + wtab + wave * kWideScratch;
    if (threadIdx.x < kExpTabLen) smem[threadIdx.x] = kExp2Tab64[threadIdx.x];
    const int size = P.dim - P.f;
    for (int e = blockIdx.x % P.E; e < P.E; e += (gridDim.x < (unsigned)P.E ? gridDim.x : P.E)) {
      const double* xs = P.XsTab + (long)e * P.tab_stride;
      const double* rec = P.blob + (long)e * P.rec.stride;
      __syncthreads();  // previous evaluation's readers are done (and the exp table is visible)
      {
        const d2t* src = reinterpret_cast<const d2t*>(xs);
        d2t* dst = reinterpret_cast<d2t*>(coords);
        for (int i = threadIdx.x; i < wtab / 2; i += blockDim.x) dst[i] = src[i];
      }
      __syncthreads();
      unsigned int ticket = 0;  // drawn one ahead (see kg_mc_kernel)
      unsigned int* next = P.next_sample + (long)e * kTicketStride;
      if (lane == 0) ticket = atomicAdd(next, 1u);
      unsigned long long tot_val = 0, tot_grad = 0;
      while (true) {
        const unsigned int sl = (unsigned int)__builtin_amdgcn_readfirstlane((int)ticket);
        if (sl >= (unsigned int)P.num_local) break;
        if (lane == 0) ticket = atomicAdd(next, 1u);
        const long so = (long)e * P.num_local + sl;
        const int best_j = P.best_j[so];
        const double* disc = rec + P.rec.disc;
        double x[DP];
  #pragma unroll
        for (int k = 0; k < DP; ++k) x[k] = (k < size) ? disc[(long)best_j * size + k] : ((k < P.dim) ? 1.0 : 0.0);
        const d2t* xg = reinterpret_cast<const d2t*>(xs) + lane;
        lds_pair_ptr xl = (lds_pair_ptr)coords + lane;
        WideEval<DP, G, true> ev{xg, xl, {P.V + so * P.v_stride + (long)lane * (1 + G)}, smem, st + kLsRows * kMaxDimPadded,
                                 P.ntiles, P.wide_lds_tiles, P.cov_type, lane, P.mean};
        unsigned long long n_val = 0, n_grad = 0;
        const double fcur = line_search_lds<DP, G>(P, ev, st, x, n_val, n_grad);
        if (lane == 0) P.best_value[so] = fcur;
        tot_val += n_val;
        tot_grad += n_grad;
        if (lane < DP) {
          double v = 0.0;
  #pragma unroll
          for (int k = 0; k < DP; ++k)
            if (lane == k) v = x[k];
          P.best_point[so * DP + lane] = v;
        }
      }
      if (lane == 0 && (tot_val | tot_grad) != 0) {
        atomicAdd(&P.counters[2 * e], tot_val);
        atomicAdd(&P.counters[2 * e + 1], tot_grad);
      }
      if (gridDim.x >= (unsigned)P.E) break;
    }
  }
};
template <int DP, int G>
__global__ __launch_bounds__(512) void kg_mc_stream_kernel(KgMcParams P) {
  kg_mc_stream_kernel_body<DP, G>::run(MOE_VBLOCK, MOE_VGRID, nullptr, P);
}

template <int DP, int G>
inline void launch_stream_inst(const KgMcParams& P, int blocks, int waves, size_t shm, hipStream_t s) {
  auto kern = kg_mc_stream_kernel<DP, G>;
  MOE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  launch_kernel_ens<kg_mc_stream_kernel_body<DP, G>, 512>(kern, dim3(blocks), dim3(waves * 64), shm, s, P);
  MOE_HIP_CHECK(hipGetLastError());
}

// built for 1 .. 4 derivative slots with every slot observed (1 + g == 1 + G: the table rows of a point are its weights)
template <int DP>
inline void launch_stream_dp(const KgMcParams& P, int G, int blocks, int waves, size_t shm, hipStream_t s) {
  switch (G) {
    case 0: launch_stream_inst<DP, 0>(P, blocks, waves, shm, s); break;
    case 1: launch_stream_inst<DP, 1>(P, blocks, waves, shm, s); break;
    case 2: launch_stream_inst<DP, 2>(P, blocks, waves, shm, s); break;
    case 3: launch_stream_inst<DP, 3>(P, blocks, waves, shm, s); break;
    case 4: launch_stream_inst<DP, 4>(P, blocks, waves, shm, s); break;
    // r4: 8 and 12 observed derivatives (C5's stretch point, g = 12: a sample's weights are 13 doubles per point, 213 KB at n = 2000 --
    // they fit no on-chip store, and the workgroup-per-sample kernel's all-register instantiation spilt them to scratch: 713 GB of
    // traffic per evaluation; here they stream once per sweep)
    case 8:
      if constexpr (DP >= 8) {
        launch_stream_inst<DP, 8>(P, blocks, waves, shm, s);
        break;
      }
      [[fallthrough]];
    case 12:
      if constexpr (DP >= 12) {
        if (G == 12) {
          launch_stream_inst<DP, 12>(P, blocks, waves, shm, s);
          break;
        }
      }
      [[fallthrough]];
    default: throw Error(MOE_ERR_RUNTIME, "unsupported derivative-slot count in the streamed-weights MC kernel");
  }
}
template <int DP>
inline void launch_stream_dp_wide(const KgMcParams& P, int G, int blocks, int waves, size_t shm, hipStream_t s) {
  switch (G) {
    case 0: launch_stream_inst<DP, 0>(P, blocks, waves, shm, s); break;
    case 4: launch_stream_inst<DP, 4>(P, blocks, waves, shm, s); break;
    case 8: launch_stream_inst<DP, 8>(P, blocks, waves, shm, s); break;
    case 12: launch_stream_inst<DP, 12>(P, blocks, waves, shm, s); break;
    default: throw Error(MOE_ERR_RUNTIME, "unsupported derivative-slot count in the streamed-weights MC kernel");
  }
}

// =====================================================================================================================
// Workgroup-per-sample variant for training sets whose coordinate table + per-wave weight slabs no longer fit the 160 KB of
// LDS (e.g. d-KG at n = 2000, d = 12, g = 3: 196 KB of coordinates, 80 KB of weights per sample).  ONE sample at a time
// per workgroup of 8 wavefronts; the point tiles are split statically over the waves and over the two on-chip stores:
//   * the first T_L tiles live in LDS (coordinates, loaded once per workgroup, and the current sample's weights), each wave
//     sweeping its share with the same software-pipelined loop as the wave-per-sample kernel;
//   * the last 8 * TR tiles live in REGISTERS: TR tiles per wave, coordinates for the kernel's lifetime and weights per
//     sample (the register file, 512 KB per CU, is the largest on-chip store; at TR = 2, d = 12, g = 3 that is 68 VGPRs).
// The host picks the smallest TR in {0, 2, 4} for which the LDS part fits.  All waves run the same line search in lockstep:
// each pass ends with one wavefront sum per wave, one LDS slot per wave, ONE __syncthreads, and a fixed-order sum of the
// per-wave partials, so every wave takes bit-identical decisions.
// =====================================================================================================================
constexpr int kMaxBlockWaves = 8;

// One point's contribution to the accumulators (shared by the LDS-tile loop and the register tiles).
template <int DP, int G, bool WG, int COV>
__device__ __forceinline__ void point_terms(const double (&cx)[DP], const double (&cw)[1 + G], const double (&xq)[DP],
                                            const double* __restrict__ etab, double& accf, double (&accg)[DP],
                                            double (&accd)[G > 0 ? G : 1]) {
  double diff[DP];
  double r2 = 1.0e-300;
#pragma unroll
  for (int k = 0; k < DP; ++k) {
    diff[k] = cx[k] - xq[k];
    r2 = fma(diff[k], diff[k], r2);
  }
  double base, first, second;
  radial3<COV, (WG || G > 0), (WG && G > 0)>(r2, etab, base, first, second);
  double sd = 0.0;
  if (G > 0) {
#pragma unroll
    for (int a = 0; a < G; ++a) sd = fma(cw[1 + a], diff[a], sd);
  }
  accf = fma(cw[0], base, accf);
  if (G > 0) accf = fma(first, sd, accf);
  if (WG) {
    double coef = cw[0] * first;
    if (G > 0) {
      coef = fma(second, sd, coef);
#pragma unroll
      for (int a = 0; a < G; ++a) accd[a] = fma(first, cw[1 + a], accd[a]);
    }
#pragma unroll
    for (int k = 0; k < DP; ++k) accg[k] = fma(coef, diff[k], accg[k]);
  }
}

// One point's contribution to T Armijo trial values f(x0 + alpha_t dv), alpha_t = al[t] (frame coordinates; see
// eval_multi_loop for the idea).  With diff0 = x_j - x0:
//   r2_t = |diff0 - alpha_t dv|^2 = A - 2 alpha_t B + alpha_t^2 |dv|^2,   A = |diff0|^2, B = diff0 . dv   (3 DP instructions once),
//   sum_a w_a (diff0 - alpha_t dv)_a = sdA - alpha_t sdB                                                  (2 G once),
// so a trial costs two fmas and a max for its distance instead of 2 DP, and one for its derivative-weight sum instead of G.
// Centred on x0 (the current iterate, inside the domain), so A and B are of the size of the distances themselves.
template <int DP, int G, int COV, int T>
__device__ __forceinline__ void point_terms_multi(const double (&cx)[DP], const double (&cw)[1 + G], const double (&x0)[DP],
                                                  const double (&dv)[DP], const double (&al)[T], double dd,
                                                  const double* __restrict__ etab, double (&acc)[T]) {
  double A = 1.0e-300, B = 0.0, sdA = 0.0, sdB = 0.0;
#pragma unroll
  for (int k = 0; k < DP; ++k) {
    const double d0 = cx[k] - x0[k];
    A = fma(d0, d0, A);
    B = fma(d0, dv[k], B);
    if (G > 0 && k < G) {
      sdA = fma(cw[1 + (k < G ? k : 0)], d0, sdA);
      sdB = fma(cw[1 + (k < G ? k : 0)], dv[k], sdB);
    }
  }
  const double mB2 = -2.0 * B;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const double r2 = fmax(fma(al[t], fma(al[t], dd, mB2), A), 1.0e-300);
    double base, first, second;
    radial3<COV, (G > 0), false>(r2, etab, base, first, second);
    acc[t] = fma(cw[0], base, acc[t]);
    if (G > 0) acc[t] = fma(first, fma(-al[t], sdB, sdA), acc[t]);
  }
}

template <int DP, int G, int TR>
struct BlockEval {
  double cx[TR > 0 ? TR : 1][DP];     // register tiles: scaled coordinates (resident for the kernel's life)
  double cw[TR > 0 ? TR : 1][1 + G];  // register tiles: weights of the current sample
  const double* __restrict__ xl;      // this wave's LDS tiles: coordinates [ntl][DP][64] (+lane)
  const double* __restrict__ wl;      // this wave's LDS tiles: weights of the current sample [ntl][1+G][64] (+lane)
  int ntl;                            // number of LDS tiles of this wave
  const double* __restrict__ etab;
  double* __restrict__ part;  // LDS [2][kMaxBlockWaves][kPartLen]
  const double* inv_lp;
  double mean;
  int nw, wave, lane, cov_type, par;
#if MOE_BLOCK_PROF
  unsigned long long c_acc = 0, c_red = 0, c_bar = 0, c_post = 0, c_n = 0, c_gtot = 0, c_gn = 0;
  unsigned long long c_tot = 0;        // cycles inside passes (all kinds)
  unsigned long long seg[4] = {0, 0, 0, 0};  // line_search_lds: cycles OUTSIDE the passes, by segment of a step
  unsigned long long seg_last = 0, seg_tot = 0;
  __device__ __forceinline__ void seg_mark(int i) {  // closes segment i: wall clock since the last mark minus pass time since then
    const unsigned long long now = __builtin_amdgcn_s_memtime();
    seg[i] += (now - seg_last) - (c_tot - seg_tot);
    seg_last = now;
    seg_tot = c_tot;
  }
#endif

  template <bool WG, int COV>
  __device__ __forceinline__ void accumulate(const double (&xq)[DP], double& accf, double (&accg)[DP],
                                             double (&accd)[G > 0 ? G : 1]) {
    // ---- LDS tiles: software-pipelined (next tile requested before the current one is consumed) ----
    if (ntl > 0) {
      lds_tile_ptr xt = (lds_tile_ptr)xl;  // single ds_read_b64 each (see lds_tile_ptr)
      lds_tile_ptr wt = (lds_tile_ptr)wl;
      double c0[DP], w0[1 + G];
#pragma unroll
      for (int k = 0; k < DP; ++k) c0[k] = xt[k * 64];
#pragma unroll
      for (int a = 0; a < 1 + G; ++a) w0[a] = wt[a * 64];
#pragma unroll 2
      for (int t = 0; t < ntl; ++t) {
        double c1[DP], w1[1 + G];
        if (t + 1 < ntl) {
          xt += DP * 64;
          wt += (1 + G) * 64;
        }
#pragma unroll
        for (int k = 0; k < DP; ++k) c1[k] = xt[k * 64];
#pragma unroll
        for (int a = 0; a < 1 + G; ++a) w1[a] = wt[a * 64];
        point_terms<DP, G, WG, COV>(c0, w0, xq, etab, accf, accg, accd);
#pragma unroll
        for (int k = 0; k < DP; ++k) c0[k] = c1[k];
#pragma unroll
        for (int a = 0; a < 1 + G; ++a) w0[a] = w1[a];
      }
    }
    // ---- register tiles ----
#pragma unroll
    for (int t = 0; t < TR; ++t) point_terms<DP, G, WG, COV>(cx[t], cw[t], xq, etab, accf, accg, accd);
  }

  // Value of the objective at TWO query points in one sweep over the tiles (the coordinates and weights of a point are
  // loaded once and used for both): the pass costs twice the arithmetic but ONE wave reduction round, one barrier and one
  // decision round -- and in this kernel a pass is dominated by exactly those (see line_search_lds).
  template <int COV>
  __device__ __forceinline__ void accumulate2(const double (&xa)[DP], const double (&xb)[DP], double& fa, double& fb) {
    double dg[DP], dd[G > 0 ? G : 1];  // unused gradient accumulators of the value-only instantiation
    if (ntl > 0) {
      lds_tile_ptr xt = (lds_tile_ptr)xl;  // single ds_read_b64 each (see lds_tile_ptr)
      lds_tile_ptr wt = (lds_tile_ptr)wl;
      double c0[DP], w0[1 + G];
#pragma unroll
      for (int k = 0; k < DP; ++k) c0[k] = xt[k * 64];
#pragma unroll
      for (int a = 0; a < 1 + G; ++a) w0[a] = wt[a * 64];
#pragma unroll 1
      for (int t = 0; t < ntl; ++t) {
        double c1[DP], w1[1 + G];
        if (t + 1 < ntl) {
          xt += DP * 64;
          wt += (1 + G) * 64;
        }
#pragma unroll
        for (int k = 0; k < DP; ++k) c1[k] = xt[k * 64];
#pragma unroll
        for (int a = 0; a < 1 + G; ++a) w1[a] = wt[a * 64];
        point_terms<DP, G, false, COV>(c0, w0, xa, etab, fa, dg, dd);
        point_terms<DP, G, false, COV>(c0, w0, xb, etab, fb, dg, dd);
#pragma unroll
        for (int k = 0; k < DP; ++k) c0[k] = c1[k];
#pragma unroll
        for (int a = 0; a < 1 + G; ++a) w0[a] = w1[a];
      }
    }
#pragma unroll
    for (int t = 0; t < TR; ++t) {
      point_terms<DP, G, false, COV>(cx[t], cw[t], xa, etab, fa, dg, dd);
      point_terms<DP, G, false, COV>(cx[t], cw[t], xb, etab, fb, dg, dd);
    }
  }

  // T trial values along one line in one sweep (point_terms_multi): one set of coordinate / weight loads, one reduction round,
  // one barrier and one decision round per T trials.
  template <int COV, int T>
  __device__ __forceinline__ void accumulateT(const double (&x0)[DP], const double (&dv)[DP], const double (&al)[T], double dd,
                                              double (&acc)[T]) {
    if (ntl > 0) {
      lds_tile_ptr xt = (lds_tile_ptr)xl;  // single ds_read_b64 each (see lds_tile_ptr)
      lds_tile_ptr wt = (lds_tile_ptr)wl;
      double c0[DP], w0[1 + G];
#pragma unroll
      for (int k = 0; k < DP; ++k) c0[k] = xt[k * 64];
#pragma unroll
      for (int a = 0; a < 1 + G; ++a) w0[a] = wt[a * 64];
#pragma unroll 1
      for (int t = 0; t < ntl; ++t) {
        double c1[DP], w1[1 + G];
        if (t + 1 < ntl) {
          xt += DP * 64;
          wt += (1 + G) * 64;
        }
#pragma unroll
        for (int k = 0; k < DP; ++k) c1[k] = xt[k * 64];
#pragma unroll
        for (int a = 0; a < 1 + G; ++a) w1[a] = wt[a * 64];
        point_terms_multi<DP, G, COV, T>(c0, w0, x0, dv, al, dd, etab, acc);
#pragma unroll
        for (int k = 0; k < DP; ++k) c0[k] = c1[k];
#pragma unroll
        for (int a = 0; a < 1 + G; ++a) w0[a] = w1[a];
      }
    }
#pragma unroll
    for (int t = 0; t < TR; ++t) point_terms_multi<DP, G, COV, T>(cx[t], cw[t], x0, dv, al, dd, etab, acc);
  }

  // Up to kMaxTrials Armijo trials x0 + alpha 2^-t dv (frame coordinates) in one pass, followed by the reference's sequence of
  // decisions over them with compile-time indices (gpp_optimization.hpp:752-769; see WaveEval::armijo_t).  The caller has
  // checked that no trial needs clamp_query.
  static constexpr int kMaxTrials = 5;
  template <int T>
  __device__ __forceinline__ void armijo_t(const double* __restrict__ x0p, const double* __restrict__ dvp, double dd, double f0,
                                           double norm, double& alpha_n, int& search, double& ftrial, bool& done,
                                           unsigned long long& n_val) {
    double x0[DP], dv[DP];  // (from the wave's LDS scratch: see line_search_lds)
#pragma unroll
    for (int k = 0; k < DP; ++k) {
      x0[k] = x0p[k];
      dv[k] = dvp[k];
    }
    double al[T], acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      al[t] = (t == 0) ? alpha_n : 0.5 * al[t > 0 ? t - 1 : 0];
      acc[t] = 0.0;
    }
    MOE_PROF_T(t0);
    if (cov_type == MOE_COV_SQUARE_EXPONENTIAL)
      accumulateT<MOE_COV_SQUARE_EXPONENTIAL, T>(x0, dv, al, dd, acc);
    else
      accumulateT<MOE_COV_MATERN_NU_2P5, T>(x0, dv, al, dd, acc);
    MOE_PROF_T(t1);
    double* slot = part + (par * kMaxBlockWaves + wave) * kPartLen;
    wave_sum_packed_store<T>(acc, slot, lane);
    MOE_PROF_T(t2);
    __syncthreads();
    MOE_PROF_T(t3);
    const double* all = part + par * kMaxBlockWaves * kPartLen;
    par ^= 1;
    double f[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      double tot = 0.0;
#pragma unroll
      for (int w = 0; w < kMaxBlockWaves; ++w) tot += (w < nw) ? all[w * kPartLen + t] : 0.0;  // (reads issued together)
      f[t] = -(mean + uniform(tot));
    }
    MOE_PROF_T(t4);
    MOE_PROF_ADD(c_acc, t0, t1);
    MOE_PROF_ADD(c_red, t1, t2);
    MOE_PROF_ADD(c_bar, t2, t3);
    MOE_PROF_ADD(c_post, t3, t4);
    MOE_PROF_ADD(c_tot, t0, t4);
#if MOE_BLOCK_PROF
    c_n++;
#endif
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if (!done) {
        ftrial = f[t];
        n_val++;
        if (ftrial - f0 > 0.5 * alpha_n * norm) {
          done = true;
        } else {
          alpha_n *= 0.5;
          if (++search >= 30) done = true;
        }
      }
    }
  }
  __device__ __forceinline__ void armijo_batch(int want, const double* __restrict__ x0, const double* __restrict__ dv, double dd,
                                               double f0, double norm, double& alpha_n, int& search, double& ftrial, bool& done,
                                               unsigned long long& n_val) {
    switch (want) {
      case 2: armijo_t<2>(x0, dv, dd, f0, norm, alpha_n, search, ftrial, done, n_val); break;
      case 3: armijo_t<3>(x0, dv, dd, f0, norm, alpha_n, search, ftrial, done, n_val); break;
      case 4: armijo_t<4>(x0, dv, dd, f0, norm, alpha_n, search, ftrial, done, n_val); break;
      default: armijo_t<5>(x0, dv, dd, f0, norm, alpha_n, search, ftrial, done, n_val); break;
    }
  }

  __device__ __forceinline__ void eval2(const double (&xa_in)[DP], const double (&xb_in)[DP], double& fa_out, double& fb_out) {
    double fa = 0.0, fb = 0.0;
    double xa[DP], xb[DP];
#pragma unroll
    for (int k = 0; k < DP; ++k) {
      xa[k] = xa_in[k];
      xb[k] = xb_in[k];
    }
    clamp_query<DP>(xa);
    clamp_query<DP>(xb);
    MOE_PROF_T(t0);
    if (cov_type == MOE_COV_SQUARE_EXPONENTIAL)
      accumulate2<MOE_COV_SQUARE_EXPONENTIAL>(xa, xb, fa, fb);
    else
      accumulate2<MOE_COV_MATERN_NU_2P5>(xa, xb, fa, fb);
    MOE_PROF_T(t1);
    double* slot = part + (par * kMaxBlockWaves + wave) * kPartLen;
    const double sa = wave_sum_uniform(fa), sb = wave_sum_uniform(fb);
    if (lane == 0) {
      slot[0] = sa;
      slot[1] = sb;
    }
    MOE_PROF_T(t2);
    __syncthreads();
    MOE_PROF_T(t3);
    const double* all = part + par * kMaxBlockWaves * kPartLen;
    par ^= 1;
    double ta = 0.0, tb = 0.0;
#pragma unroll
    for (int w = 0; w < kMaxBlockWaves; ++w) {  // fixed trip count: the eight reads are issued together
      ta += (w < nw) ? all[w * kPartLen] : 0.0;
      tb += (w < nw) ? all[w * kPartLen + 1] : 0.0;
    }
    fa_out = -(mean + uniform(ta));
    fb_out = -(mean + uniform(tb));
    MOE_PROF_T(t4);
    MOE_PROF_ADD(c_acc, t0, t1);
    MOE_PROF_ADD(c_red, t1, t2);
    MOE_PROF_ADD(c_bar, t2, t3);
    MOE_PROF_ADD(c_post, t3, t4);
    MOE_PROF_ADD(c_tot, t0, t4);
#if MOE_BLOCK_PROF
    c_n++;
#endif
  }

  // the query read from the wave's LDS scratch (line_search_lds)
  template <bool WG>
  __device__ __forceinline__ double eval_p(const double* __restrict__ xq_ptr, double (&grad)[DP]) {
    double xq[DP];
#pragma unroll
    for (int k = 0; k < DP; ++k) xq[k] = xq_ptr[k];
    return eval<WG>(xq, grad);
  }
  // value + gradient with the gradient handed back ONE COMPONENT PER LANE (lane k < DP: d f / d x_k in the original units,
  // scale_l = this lane's frame scale): the cross-wave sums are lane-distributed anyway, and fetching them as wave-uniform
  // values cost 2 (1 + DP + G) v_readlane into a scalar file that was already spilling
  __device__ __forceinline__ double eval_p_lane(const double* __restrict__ xq_ptr, double scale_l, double& grad_l) {
    double xq[DP], unused[DP];
#pragma unroll
    for (int k = 0; k < DP; ++k) xq[k] = xq_ptr[k];
    return eval<true, true>(xq, unused, scale_l, &grad_l);
  }

  template <bool WG, bool LANEOUT = false>
  __device__ __forceinline__ double eval(const double (&xq_in)[DP], double (&grad)[DP], double scale_l = 0.0,
                                         double* grad_l = nullptr) {
    double accf = 0.0, accg[DP], accd[G > 0 ? G : 1];
    double xq[DP];
#pragma unroll
    for (int k = 0; k < DP; ++k) xq[k] = xq_in[k];
    clamp_query<DP>(xq);
#pragma unroll
    for (int k = 0; k < DP; ++k) accg[k] = 0.0;
#pragma unroll
    for (int a = 0; a < (G > 0 ? G : 1); ++a) accd[a] = 0.0;
    MOE_PROF_T(t0);
    if (cov_type == MOE_COV_SQUARE_EXPONENTIAL)
      accumulate<WG, MOE_COV_SQUARE_EXPONENTIAL>(xq, accf, accg, accd);
    else
      accumulate<WG, MOE_COV_MATERN_NU_2P5>(xq, accf, accg, accd);
    MOE_PROF_T(t1);
    double* slot = part + (par * kMaxBlockWaves + wave) * kPartLen;
    if (WG) {
      double all_sums[1 + DP + (G > 0 ? G : 0)];  // f | gradient sums | derivative-weight sums: one packed reduction
      all_sums[0] = accf;
#pragma unroll
      for (int k = 0; k < DP; ++k) all_sums[1 + k] = accg[k];
      if (G > 0) {
#pragma unroll
        for (int a = 0; a < G; ++a) all_sums[1 + DP + a] = accd[a];
      }
      wave_sum_packed_store<1 + DP + (G > 0 ? G : 0)>(all_sums, slot, lane);
    } else {
      const double sf = wave_sum_uniform(accf);
      if (lane == 0) slot[0] = sf;
    }
    MOE_PROF_T(t2);
    __syncthreads();
    MOE_PROF_T(t3);
    MOE_PROF_ADD(c_acc, t0, t1);
    MOE_PROF_ADD(c_red, t1, t2);
    MOE_PROF_ADD(c_bar, t2, t3);
#if MOE_BLOCK_PROF
    c_n++;
#endif
    const double* all = part + par * kMaxBlockWaves * kPartLen;
    par ^= 1;
    double f;
    if (WG) {
      // One component per LANE: lane l sums slot entry l over the waves (eight independent LDS reads, wave order), then the
      // components come back as wave-uniform values through v_readlane -- instead of (1 + DP + G) x nw dependent
      // broadcast reads, each a full LDS round trip (12k cycles per gradient pass at DP = 12, G = 3).
      double comp = 0.0;
      const int lc = lane < kPartLen ? lane : 0;
#pragma unroll
      for (int w = 0; w < kMaxBlockWaves; ++w) comp += (w < nw) ? all[w * kPartLen + lc] : 0.0;
      auto take = [&](int idx) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(comp), idx);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(comp), idx);
        return __hiloint2double(hi, lo);
      };
      f = take(0);
      if (LANEOUT) {
        auto from_lane = [&](int src) {  // comp of lane `src` (per-lane index): two ds_bpermute_b32
          const int lo = __builtin_amdgcn_ds_bpermute(src << 2, __double2loint(comp));
          const int hi = __builtin_amdgcn_ds_bpermute(src << 2, __double2hiint(comp));
          return __hiloint2double(hi, lo);
        };
        double v = from_lane(min(lane + 1, 63));
        if (G > 0) {
          const double vd = from_lane(min(lane + 1 + DP, 63));
          if (lane < G) v -= vd;
        }
        *grad_l = -(v * scale_l);
      } else {
#pragma unroll
        for (int k = 0; k < DP; ++k) {
          double v = take(1 + k);
          if (G > 0 && k < G) v -= take(1 + DP + (k < G ? k : 0));
          grad[k] = -(v * inv_lp[k]);
        }
      }
    } else {
      f = 0.0;
#pragma unroll
      for (int w = 0; w < kMaxBlockWaves; ++w) f += (w < nw) ? all[w * kPartLen] : 0.0;
    }
    const double fret = -(mean + uniform(f));
    MOE_PROF_T(t4);
    MOE_PROF_ADD(c_post, t3, t4);
    MOE_PROF_ADD(c_tot, t0, t4);
#if MOE_BLOCK_PROF
    if (WG) {
      c_gtot += t4 - t0;
      c_gn++;
    }
#endif
    return fret;
  }
};

// v(j, a) of the weight block for point j (see file header), alpha and the derivative scaling folded in.  The W loads
// of 16 columns x (1 + G) rows are issued together (64 independent L2 loads per lane and round at G = 3): issued four at a
// time this phase took a quarter of the kernel at m = 16 (latency x 64 rounds per wave and sample).
template <int G>
__device__ __forceinline__ void point_weights(const KgMcParams& P, const double* __restrict__ We, const double* __restrict__ zb,
                                              int j, double (&w)[1 + G]) {
  const int n = P.n, u = P.u, m = P.m, g1 = 1 + P.g;
  double v[1 + G];
#pragma unroll
  for (int a = 0; a < 1 + G; ++a) v[a] = (a < g1 && j < n) ? P.KinvY[(long)j * g1 + a] : 0.0;
  if (j < n) {
    // the 1 + G rows of a point are contiguous in a column of W: with an even row count they are read as 16-byte pairs, so
    // that a wavefront's load covers each cache line once (row by row, a lane stride of (1 + G) doubles touches every line
    // 1 + G times and the phase becomes L2 -> CU bandwidth: 4 MB per sample instead of 1 MB at g = 3)
    typedef double d2 __attribute__((ext_vector_type(2)));
    const bool pairs = (G & 1) == 1 && g1 == 1 + G && (P.N & 1) == 0 && (reinterpret_cast<size_t>(We) & 15) == 0;
    for (int c0 = 0; c0 < m; c0 += 16) {
      double l[1 + G][16];
      if (pairs) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const d2* src = reinterpret_cast<const d2*>(We + (long)j * g1 + (long)min(c0 + i, m - 1) * P.N);
#pragma unroll
          for (int a2 = 0; a2 < (1 + G) / 2; ++a2) {
            const d2 pr = src[a2];
            l[2 * a2][i] = pr.x;
            l[2 * a2 + ((1 + G) > 1 ? 1 : 0)][i] = pr.y;
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {  // column outer: the rows of one point share their cache lines
#pragma unroll
          for (int a = 0; a < 1 + G; ++a)
            l[a][i] = We[(long)j * g1 + (a < g1 ? a : 0) + (long)min(c0 + i, m - 1) * P.N];
        }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const double beta = zb[kMaxMB + min(c0 + i, kMaxMB - 1)];  // 0 beyond m
#pragma unroll
        for (int a = 0; a < 1 + G; ++a) v[a] = fma(-l[a][i], beta, v[a]);
      }
    }
  }
#pragma unroll
  for (int a = 0; a < 1 + G; ++a) {
    double t = 0.0;
    if (a < g1) {
      t = v[a];
      if (j >= n) t = (j < n + u) ? zb[kMaxMB + (j - n) * g1 + a] : 0.0;
      // fold alpha and, for derivative weights, the -1/l of (x - X)_{d_a} / l^2 = -diff_scaled[a] / l
      t *= (a == 0) ? P.alpha : -P.alpha * P.inv_lp[a > 0 ? a - 1 : 0];
    }
    w[a] = t;
  }
}

// The same weights read back from the precomputed table V (kg_sample_weights_kernel): Vs = V + sample * N.  Rows of a
// point are contiguous, so with an even row count they travel as 16-byte pairs (see point_weights).
template <int G>
__device__ __forceinline__ void point_weights_pre(const KgMcParams& P, const double* __restrict__ Vs,
                                                  const double* __restrict__ zb, int j, double (&w)[1 + G]) {
  const int n = P.n, u = P.u, g1 = 1 + P.g;
  typedef double d2 __attribute__((ext_vector_type(2)));
  const bool pairs = (G & 1) == 1 && g1 == 1 + G && (P.N & 1) == 0 && (reinterpret_cast<size_t>(Vs) & 15) == 0;
  if (j < n) {
    if (pairs) {
      const d2* src = reinterpret_cast<const d2*>(Vs + (long)j * g1);
#pragma unroll
      for (int a2 = 0; a2 < (1 + G) / 2; ++a2) {
        const d2 pr = src[a2];
        w[2 * a2] = pr.x;
        w[2 * a2 + ((1 + G) > 1 ? 1 : 0)] = pr.y;
      }
    } else {
#pragma unroll
      for (int a = 0; a < 1 + G; ++a) w[a] = (a < g1) ? Vs[(long)j * g1 + a] : 0.0;
    }
  } else {
#pragma unroll
    for (int a = 0; a < 1 + G; ++a) {
      double t = 0.0;
      if (a < g1 && j < n + u) {
        t = zb[kMaxMB + (j - n) * g1 + a];
        t *= (a == 0) ? P.alpha : -P.alpha * P.inv_lp[a > 0 ? a - 1 : 0];
      }
      w[a] = t;
    }
  }
}

// Fixed LDS words of the workgroup-per-sample kernel (doubles), before the tile data.
constexpr int kBlockFixed = kExpTabLen + 2 * kMaxMB + 2 * kMaxBlockWaves * kPartLen + 2 + kMaxBlockWaves * kLsRows * kMaxDimPadded;

template <int DP, int G, int TR>
struct kg_mc_block_kernel_body {
  static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void*, const KgMcParams& P, int num_lds_tiles) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    // LDS: [64] exp table | z / beta scratch of draw_z_beta [0, 2 kMaxM) = [0, kMaxMB), beta of the sample [kMaxMB, 2 kMaxMB) |
    //      partial slots [2][8][kPartLen] | control words (2 doubles) |
    //      line-search state [8][kLsRows kMaxDimPadded] | coordinates of the LDS tiles [T_L][DP][64] | their weights [T_L][1+G][64]
    double* etab = smem;
    double* zb = smem + kExpTabLen;
    double* part = zb + 2 * kMaxMB;
    int* ctl = reinterpret_cast<int*>(part + 2 * kMaxBlockWaves * kPartLen);  // [0] sample index, [1] best discretised point
    double* stw = part + 2 * kMaxBlockWaves * kPartLen + 2 + (threadIdx.x >> 6) * (kLsRows * kMaxDimPadded);
    double* ldsx = smem + kBlockFixed;
    double* ldsw = ldsx + (long)num_lds_tiles * DP * 64;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nw = blockDim.x >> 6;
    const int m = P.m;
    const int size = P.dim - P.f;
    // LDS tiles [0, T_L) are dealt to the waves in contiguous runs; register tiles are T_L + wave * TR + t
    const int TL = num_lds_tiles;
    const int per = (TL + nw - 1) / nw;
    const int tl0 = min(wave * per, TL), tl1 = min(tl0 + per, TL);
    if (threadIdx.x < kExpTabLen) etab[threadIdx.x] = kExp2Tab64[threadIdx.x];
    BlockEval<DP, G, TR> ev;
    ev.xl = ldsx + (long)tl0 * DP * 64 + lane;
    ev.wl = ldsw + (long)tl0 * (1 + G) * 64 + lane;
    ev.ntl = tl1 - tl0;
    ev.etab = etab;
    ev.part = part;
    ev.inv_lp = P.inv_lp;
    ev.mean = P.mean;
    ev.nw = nw;
    ev.wave = wave;
    ev.lane = lane;
    ev.cov_type = P.cov_type;
    ev.par = 0;
    for (int e = blockIdx.x % P.E; e < P.E; e += (gridDim.x < (unsigned)P.E ? gridDim.x : P.E)) {
      const double* tab = P.XsTab + (long)e * P.tab_stride;
      const double* rec = P.blob + (long)e * P.rec.stride;
      const double* Lsm = rec + P.rec.L;
      const double* We = P.W + (long)e * P.w_stride;
      __syncthreads();  // previous evaluation's readers of the LDS coordinates are done
      if constexpr (DP > 16) {  // (the table of the wide dimensions holds its rows in pairs: WideEval)
        for (int t = threadIdx.x; t < TL * DP * 64; t += blockDim.x) {
          const int l = t & 63, r = (t >> 6) % DP, tile = (t >> 6) / DP;
          ldsx[t] = tab[(((long)tile * (DP / 2) + (r >> 1)) * 64 + l) * 2 + (r & 1)];
        }
      } else {
        for (int t = threadIdx.x; t < TL * DP * 64; t += blockDim.x) ldsx[t] = tab[t];
      }
  #pragma unroll
      for (int t = 0; t < TR; ++t) {
        const int tile = TL + wave * TR + t;
  #pragma unroll
        for (int k = 0; k < DP; ++k) ev.cx[t][k] = (tile < P.ntiles) ? tab[((long)tile * DP + k) * 64 + lane] : 0.0;
      }
  #if MOE_BLOCK_PROF
      unsigned long long p_tick = 0, p_setup = 0, p_w = 0, p_ls = 0, p_zb = 0, p_scan = 0;
  #endif
      while (true) {
        MOE_PROF_T(k0);
        if (threadIdx.x == 0) ctl[0] = (int)atomicAdd(&P.next_sample[(long)e * kTicketStride], 1u);
        __syncthreads();
        const int sl = ctl[0];
        if (sl >= P.num_local) break;
        const int s = P.first_sample + sl;
        double zc = 0.0, bc = 0.0;
        MOE_PROF_T(k1);
        if (wave == 0) {
          MOE_PROF_T(k1a0);
          if (P.best_j != nullptr) {
            // beta and the discretised-set winner depend on z alone: a pre-pass computed them for every sample with the whole
            // chip (here they cost one wavefront 10 % of the sample while seven wait at the barrier)
            const long so0 = (long)e * P.num_local + sl;
            for (int c = lane; c < kMaxMB; c += 64) zb[kMaxMB + c] = (c < m) ? P.beta[so0 * m + c] : 0.0;
            if (lane == 0) ctl[1] = P.best_j[so0];
          } else {  // (m <= 64 only: one lane per component)
            draw_z_beta(P, Lsm, s, lane, zb, zc, bc);
            zb[kMaxMB + lane] = bc;
            zb[kMaxMB + 64 + lane] = 0.0;
            const int bj = discrete_scan(P, rec, zb, lane);
            if (lane == 0) ctl[1] = bj;
          }
          MOE_PROF_T(k1a);
          MOE_PROF_T(k1b);
          (void)zc;
          MOE_PROF_ADD(p_zb, k1a0, k1a);
          MOE_PROF_ADD(p_scan, k1a, k1b);
        }
        __syncthreads();
        MOE_PROF_T(k2);
        // ---- weights of this wave's points for this sample: LDS tiles into the LDS slab, register tiles into registers ----
        {
          double* wdst = ldsw + (long)tl0 * (1 + G) * 64 + lane;
          if (P.V != nullptr) {
            const double* Vs = P.V + ((long)e * P.num_local + sl) * P.v_stride;
  #pragma unroll 2
            for (int t = tl0; t < tl1; ++t) {
              double w[1 + G];
              point_weights_pre<G>(P, Vs, zb, t * 64 + lane, w);
  #pragma unroll
              for (int a = 0; a < 1 + G; ++a) wdst[a * 64] = w[a];  // read back only by this lane
              wdst += (1 + G) * 64;
            }
  #pragma unroll
            for (int t = 0; t < TR; ++t) point_weights_pre<G>(P, Vs, zb, (TL + wave * TR + t) * 64 + lane, ev.cw[t]);
          } else {
  #pragma unroll 1
            for (int t = tl0; t < tl1; ++t) {
              double w[1 + G];
              point_weights<G>(P, We, zb, t * 64 + lane, w);
  #pragma unroll
              for (int a = 0; a < 1 + G; ++a) wdst[a * 64] = w[a];  // read back only by this lane
              wdst += (1 + G) * 64;
            }
  #pragma unroll
            for (int t = 0; t < TR; ++t) point_weights<G>(P, We, zb, (TL + wave * TR + t) * 64 + lane, ev.cw[t]);
          }
        }
        MOE_PROF_T(k3);
        const int best_j = ctl[1];
        const double* disc = rec + P.rec.disc;
        double x[DP];
  #pragma unroll
        for (int k = 0; k < DP; ++k) x[k] = (k < size) ? disc[(long)best_j * size + k] : ((k < P.dim) ? 1.0 : 0.0);
        unsigned long long n_val = 0, n_grad = 0;
        const double fcur = line_search_lds<DP, G>(P, ev, stw, x, n_val, n_grad);
        MOE_PROF_T(k4);
        MOE_PROF_ADD(p_tick, k0, k1);
        MOE_PROF_ADD(p_setup, k1, k2);
        MOE_PROF_ADD(p_w, k2, k3);
        MOE_PROF_ADD(p_ls, k3, k4);
        if (wave == 0) {
          const long so = (long)e * P.num_local + sl;
          if (lane == 0) {
            P.best_value[so] = fcur;
            atomicAdd(&P.counters[2 * e], n_val);
            atomicAdd(&P.counters[2 * e + 1], n_grad);
          }
          if (lane < DP) {
            double v = 0.0;
  #pragma unroll
            for (int k = 0; k < DP; ++k)
              if (lane == k) v = x[k];
            P.best_point[so * DP + lane] = v;
          }
          if (P.best_j == nullptr && lane < m) P.beta[so * m + lane] = bc;  // (the pre-pass already stored it)
        }
      }
  #if MOE_BLOCK_PROF
      if (threadIdx.x == 0) {  // wave 0's view: [0..3] ticket, z/beta/scan, weights, line search; [4..8] inside the passes
        atomicAdd(&P.prof[0], p_tick);
        atomicAdd(&P.prof[1], p_setup);
        atomicAdd(&P.prof[2], p_w);
        atomicAdd(&P.prof[3], p_ls);
        atomicAdd(&P.prof[4], ev.c_acc);
        atomicAdd(&P.prof[5], ev.c_red);
        atomicAdd(&P.prof[6], ev.c_bar);
        atomicAdd(&P.prof[7], ev.c_post);
        atomicAdd(&P.prof[8], ev.c_n);
        atomicAdd(&P.prof[9], ev.c_gtot);
        atomicAdd(&P.prof[10], ev.c_gn);
        atomicAdd(&P.prof[11], p_zb);
        atomicAdd(&P.prof[12], p_scan);
        atomicAdd(&P.prof[13], ev.seg[0]);
        atomicAdd(&P.prof[14], ev.seg[1]);
        atomicAdd(&P.prof[15], ev.seg[2] + ev.seg[3]);
      }
  #endif
      if (gridDim.x >= (unsigned)P.E) break;
    }
  }
};
template <int DP, int G, int TR>
__global__ __launch_bounds__(512) void kg_mc_block_kernel(KgMcParams P, int num_lds_tiles) {
  kg_mc_block_kernel_body<DP, G, TR>::run(MOE_VBLOCK, MOE_VGRID, nullptr, P, num_lds_tiles);
}

template <int DP, int G, int TR>
inline void launch_block_inst(const KgMcParams& P, int num_lds_tiles, int blocks, int waves, hipStream_t s) {
  const size_t shm = sizeof(double) * (kBlockFixed + (size_t)num_lds_tiles * (DP + 1 + G) * 64);
  auto kern = kg_mc_block_kernel<DP, G, TR>;
  MOE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  launch_kernel_ens<kg_mc_block_kernel_body<DP, G, TR>, 512>(kern, dim3(blocks), dim3(waves * 64), shm, s, P, num_lds_tiles);
  MOE_HIP_CHECK(hipGetLastError());
}

template <int DP, int G>
inline void launch_block_g(const KgMcParams& P, int tr, int num_lds_tiles, int blocks, int waves, hipStream_t s) {
  switch (tr) {
    case 0: launch_block_inst<DP, G, 0>(P, num_lds_tiles, blocks, waves, s); break;
    case 2: launch_block_inst<DP, G, 2>(P, num_lds_tiles, blocks, waves, s); break;
    case 4: launch_block_inst<DP, G, 4>(P, num_lds_tiles, blocks, waves, s); break;
    default: throw Error(MOE_ERR_RUNTIME, "unsupported register-tile count in the workgroup-per-sample MC kernel");
  }
}

template <int DP>
inline void launch_block_dp(const KgMcParams& P, int G, int tr, int num_lds_tiles, int blocks, int waves, hipStream_t s) {
  switch (G) {
    case 0: launch_block_g<DP, 0>(P, tr, num_lds_tiles, blocks, waves, s); break;
    case 1: launch_block_g<DP, 1>(P, tr, num_lds_tiles, blocks, waves, s); break;
    case 2: launch_block_g<DP, 2>(P, tr, num_lds_tiles, blocks, waves, s); break;
    case 3: launch_block_g<DP, 3>(P, tr, num_lds_tiles, blocks, waves, s); break;
    case 4: launch_block_g<DP, 4>(P, tr, num_lds_tiles, blocks, waves, s); break;
    // more observed derivatives (up to 12: every dimension of C5) run in the next larger slot count, the unused slots
    // carrying zero weights; only this kernel is instantiated for them (the host never sends them to the wave-per-sample one)
    case 8: launch_block_g<DP, 8>(P, tr, num_lds_tiles, blocks, waves, s); break;
    case 12: launch_block_g<DP, 12>(P, tr, num_lds_tiles, blocks, waves, s); break;
    default: throw Error(MOE_ERR_RUNTIME, "unsupported derivative-slot count in the MC kernel");
  }
}

#include "kg_mc_lane.hpp"

template <int DP, int G, bool XLDS, bool SMALL>
inline void launch_inst2(const KgMcParams& P, int blocks, int waves, size_t shm, hipStream_t s) {
  auto kern = kg_mc_kernel<DP, G, XLDS, SMALL>;
  MOE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  launch_kernel_ens<kg_mc_kernel_body<DP, G, XLDS, SMALL>, (SMALL ? 1024 : 512)>(kern, dim3(blocks), dim3(waves * 64), shm, s, P);
  MOE_HIP_CHECK(hipGetLastError());
}

// waves > 8 selects the many-wavefront instantiation (only built with the LDS coordinate table)
template <int DP, int G, bool XLDS>
inline void launch_inst(const KgMcParams& P, int blocks, int waves, size_t shm, hipStream_t s) {
  if (XLDS && waves > 8)
    launch_inst2<DP, G, XLDS, XLDS>(P, blocks, waves, shm, s);
  else
    launch_inst2<DP, G, XLDS, false>(P, blocks, waves, shm, s);
}

template <int DP>
inline void launch_dp(const KgMcParams& P, int G, bool xlds, int blocks, int waves, size_t shm, hipStream_t s) {
  switch (G) {
    case 0:
      if (xlds) launch_inst<DP, 0, true>(P, blocks, waves, shm, s); else launch_inst<DP, 0, false>(P, blocks, waves, shm, s);
      break;
    case 1:
      if (xlds) launch_inst<DP, 1, true>(P, blocks, waves, shm, s); else launch_inst<DP, 1, false>(P, blocks, waves, shm, s);
      break;
    case 2:
      if (xlds) launch_inst<DP, 2, true>(P, blocks, waves, shm, s); else launch_inst<DP, 2, false>(P, blocks, waves, shm, s);
      break;
    case 3:
      if (xlds) launch_inst<DP, 3, true>(P, blocks, waves, shm, s); else launch_inst<DP, 3, false>(P, blocks, waves, shm, s);
      break;
    case 4:
      if (xlds) launch_inst<DP, 4, true>(P, blocks, waves, shm, s); else launch_inst<DP, 4, false>(P, blocks, waves, shm, s);
      break;
    default: throw Error(MOE_ERR_RUNTIME, "unsupported derivative-slot count in the MC kernel");
  }
}

// The reduced instantiation sets of the wide padded dimensions (24, 32).
template <int DP>
inline void launch_block_dp_wide(const KgMcParams& P, int G, int tr, int num_lds_tiles, int blocks, int waves, hipStream_t s) {
  if (tr != 0) throw Error(MOE_ERR_RUNTIME, "d > 16: the workgroup-per-sample MC kernel is built with all tiles in LDS only");
  switch (G) {
    case 0: launch_block_inst<DP, 0, 0>(P, num_lds_tiles, blocks, waves, s); break;
    case 4: launch_block_inst<DP, 4, 0>(P, num_lds_tiles, blocks, waves, s); break;
    case 8: launch_block_inst<DP, 8, 0>(P, num_lds_tiles, blocks, waves, s); break;
    case 12: launch_block_inst<DP, 12, 0>(P, num_lds_tiles, blocks, waves, s); break;
    default: throw Error(MOE_ERR_RUNTIME, "unsupported derivative-slot count in the MC kernel");
  }
}

template <int DP>
inline void launch_dp_wide(const KgMcParams& P, int G, bool xlds, int blocks, int waves, size_t shm, hipStream_t s) {
  if (xlds || waves > 8) throw Error(MOE_ERR_RUNTIME, "d > 16: the wave-per-sample MC kernel streams coordinates from L2 only");
  switch (G) {
    case 0: launch_inst2<DP, 0, false, false>(P, blocks, waves, shm, s); break;
    case 4: launch_inst2<DP, 4, false, false>(P, blocks, waves, shm, s); break;
    default: throw Error(MOE_ERR_RUNTIME, "unsupported derivative-slot count in the MC kernel");
  }
}

}  // namespace mc
#endif  // __HIPCC__

}  // namespace moe
