// cornell_moe_amd/csrc/kg_mc_lane.hpp -- round 5: the wave-per-sample MC kernel with its line search LANE-PARKED (included by
// kg_mc.hpp, inside namespace moe::mc).
//
// kg_mc_kernel (kg_mc.hpp) keeps the wave-uniform vectors of a sample's line search -- the iterate, its gradient, the trial
// direction, the step: five arrays of DP doubles -- as wave-uniform VALUES, i.e. 10 DP vector registers that stay live across the
// tile loops (which sit on the 256-register cliff), every operation on them DP instructions on 64 identical lanes, and the scalar
// file around them spilt to VGPR lanes (472 spill slots: v_writelane / v_readlane, a vector-ALU slot each way).  Here lane r < DP
// OWNS row r of every such vector:
//   * the iterate (original units and frame), the gradient, the step, the restart's starting point are ONE register each; an
//     update of a vector is one instruction; TensorProductDomain::LimitUpdate runs once, on the lanes;
//   * what a tile loop needs wave-uniform (the query of a gradient pass, x2 / d2 of a trial line) is handed over through DP
//     doubles of the wave's LDS scratch and read back as broadcasts, immediately before the pass -- nothing of it is live
//     between passes;
//   * sums over the rows that the reference takes in coordinate order (|grad|^2, |step|^2, the restart displacement, and the
//     three sums of a trial line) are taken in that order over the broadcast values: every number below has the bits the
//     frame line search of kg_mc.hpp (line_search_frame) produces -- the two kernels can be compared bit for bit
//     (MOE_KG_LANE=0 selects the old one; tests/test_gpu_sweep.py::test_lane_kernel_bit_identical);
//   * everything a sample reads that does not depend on the sample -- L, mu_n and L^-1 cov_n of the discretised set, the set
//     itself, the bounds, the frame constants -- is copied to LDS once per workgroup and evaluation: a sample's set-up is one
//     round trip to L2 (its normal draws and the K^-1 y / W tiles, requested together), not five.
// Semantics: gpp_knowledge_gradient_optimization.cpp:164-196, 420-472; gpp_optimization.hpp:708-828, 1242-1283;
// gpp_domain.cpp:64-105.
#pragma once

// threads per workgroup the kernel is compiled for (512: two wavefronts per SIMD, up to 256 VGPRs)
constexpr int kLaneMaxThreads = 512;
constexpr int kLaneGradUnroll = 4;  // tiles unrolled in the gradient pass (r5 A/B: 1 / 2 / 4 -> -4 % / 0 / +0.8 %)
// (closed r5 A/Bs, `profiles/r05_a_lane_ab.txt`, `r05_b_unroll_variants_ab.txt`: a pass's wave-uniform operands as SCALAR registers --
//  v_readfirstlane -- against wave-uniform vector registers +-0.5 %; kernel arguments re-read per sample against held -1 %)

constexpr int kLaneCstRows = 8;  // s | 1 / s | centre | pinned value | lower bound | upper bound (original units) | perm | free (1 / 0)
constexpr int kMaxLaneDP = 16;

template <int DP>
__device__ __forceinline__ void fill_lane_constants(const KgMcParams& P, double* __restrict__ cst) {
  const int r = threadIdx.x;
  if (r < DP) {
    const double sc = P.inv_lp[r];
    cst[r] = sc;
    cst[DP + r] = (sc != 0.0) ? 1.0 / sc : 0.0;
    cst[2 * DP + r] = P.center[r];
    cst[3 * DP + r] = (P.perm[r] < P.dim) ? 1.0 : 0.0;  // what a pinned row holds: fidelity coordinates 1, pads 0 (.cpp:353-357)
    cst[4 * DP + r] = P.bounds[2 * r];
    cst[5 * DP + r] = P.bounds[2 * r + 1];
    cst[6 * DP + r] = (double)P.perm[r];
    cst[7 * DP + r] = ((P.free_mask >> r) & 1u) ? 1.0 : 0.0;
  }
}

typedef volatile __attribute__((address_space(3))) double* lds_rw_ptr;

// a lane-parked vector (lane r < DP holds row r) as DP wave-uniform values, through DP doubles of the wave's scratch
// (every lane writes: lanes >= DP into the row's spare slot DP -- a predicated store costs an exec-mask save / restore around it, and
//  the mask itself was one more scalar pair to keep or spill)
template <int DP>
__device__ __forceinline__ void lane_broadcast(double v_l, lds_rw_ptr row, int lane, double (&out)[DP]) {
  row[lane < DP ? lane : DP] = v_l;
#pragma unroll
  for (int k = 0; k < DP; ++k) out[k] = row[k];  // (LDS operations of one wave complete in order)
}

template <int DP>
__device__ __forceinline__ void make_scalar(double (&v)[DP]) {
#pragma unroll
  for (int k = 0; k < DP; ++k) v[k] = uniform(v[k]);
}

// max over the 64 lanes, wave-uniform (fmax is exact: any order gives the bits of the butterfly in discrete_scan)
__device__ __forceinline__ double wave_max_uniform(double v) {
  v = fmax(v, dpp_move<0xB1, 0xf, true>(v));
  v = fmax(v, dpp_move<0x4E, 0xf, true>(v));
  v = fmax(v, dpp_move<0x141, 0xf, true>(v));
  v = fmax(v, dpp_move<0x140, 0xf, true>(v));
  // rows -> lane 63: row_bcast15 / row_bcast31 leave lanes outside their row mask with 0 from dpp_move<.., false>; take the row
  // maxima with read-outs instead (four v_readlane pairs)
  const double r0 = lane_value(v, 0), r1 = lane_value(v, 16), r2 = lane_value(v, 32), r3 = lane_value(v, 48);
  return fmax(fmax(r0, r1), fmax(r2, r3));
}

// Value + gradient pass with the gradient returned LANE-PARKED (lane k < DP: d f / d x'_k in the frame): eval_loop<DP, G, true, COV,
// false, true, false, true> of kg_mc.hpp with the read-back of the packed sums changed -- lane k reads sum k.  `S` = 2 kMaxLaneDP doubles of scratch.
template <int DP, int G, int COV>
__device__ __forceinline__ double grad_pass_parked(const double* __restrict__ xs, const double* __restrict__ aw,
                                                   const double* __restrict__ etab, int ntiles, double mean, const double (&xq)[DP],
                                                   lds_rw_ptr S, int lane, double& g_l) {
  constexpr int XR = DP + 1;  // rows per coordinate tile (the |x|^2 row is not read here)
  constexpr int WR = 1 + G;  // weight rows per tile
  double accf = 0.0;
  double accg[DP];
  double accd[G > 0 ? G : 1];
#pragma unroll
  for (int k = 0; k < DP; ++k) accg[k] = 0.0;
#pragma unroll
  for (int a = 0; a < (G > 0 ? G : 1); ++a) accd[a] = 0.0;
  lds_tile_ptr xt = (lds_tile_ptr)(xs + lane);
  lds_tile_ptr wt = (lds_tile_ptr)(aw + lane);
  double cx[DP], cw[WR];
#pragma unroll
  for (int k = 0; k < DP; ++k) cx[k] = xt[k * 64];
#pragma unroll
  for (int a = 0; a < WR; ++a) cw[a] = wt[a * 64];
#pragma unroll kLaneGradUnroll
  for (int t = 0; t < ntiles; ++t) {
    double nx[DP], nw[WR];
    xt += XR * 64;
    wt += WR * 64;
#pragma unroll
    for (int k = 0; k < DP; ++k) nx[k] = xt[k * 64];
#pragma unroll
    for (int a = 0; a < WR; ++a) nw[a] = wt[a * 64];
    double diff[DP];
    double r2 = 1.0e-300;
#pragma unroll
    for (int k = 0; k < DP; ++k) {
      diff[k] = cx[k] - xq[k];
      r2 = fma(diff[k], diff[k], r2);
    }
    const double w0 = cw[0];
    double base, first, second;
    radial3<COV, true, (G > 0)>(r2, etab, base, first, second);
    double sd = 0.0;
    if (G > 0) {
#pragma unroll
      for (int a = 0; a < G; ++a) sd = fma(cw[1 + a], diff[a], sd);
    }
    accf = fma(w0, base, accf);
    if (G > 0) accf = fma(first, sd, accf);
    double coef = w0 * first;
    if (G > 0) {
      coef = fma(second, sd, coef);
#pragma unroll
      for (int a = 0; a < G; ++a) accd[a] = fma(first, cw[1 + a], accd[a]);
    }
#pragma unroll
    for (int k = 0; k < DP; ++k) accg[k] = fma(coef, diff[k], accg[k]);
#pragma unroll
    for (int k = 0; k < DP; ++k) cx[k] = nx[k];
#pragma unroll
    for (int a = 0; a < WR; ++a) cw[a] = nw[a];
  }
  const double mu = mean + wave_sum_uniform(accf);
  // the DP sums folded into DP / 4 in-row reductions (wave_sum_packed_lds's tree), written by the first lane of each row ...
  {
    const int row = lane >> 4;
    const int slot = ((row & 1) << 1) | (row >> 1);  // rows hold v0 | v2 | v1 | v3
    const int sidx = ((lane & 15) == 0) ? slot : kMaxLaneDP + slot;  // (the other lanes write behind the sums: no predicated store)
#pragma unroll
    for (int i = 0; i < DP; i += 4) {
      const double q = row_sum(fold16(fold32(accg[i], accg[i + 1]), fold32(accg[i + 2], accg[i + 3])));
      S[i + sidx] = q;
    }
  }
  // ... and read back one per lane
  double v = S[lane < DP ? lane : 0];
  if (G > 0) {
#pragma unroll
    for (int a = 0; a < G; ++a) {
      const double sa = wave_sum_uniform(accd[a]);
      v = (lane == a) ? v - sa : v;
    }
  }
  g_l = -v;
  return -mu;
}

// One MC sample on the lane-parked line search.  xs = the workgroup's LDS coordinate table, aw = this wave's weight slab, zb = its
// scratch (2 kMaxM doubles: z | beta during the set-up, then the line search's rows), cst = the lane constants, rc = the LDS copy of
// the evaluation's record head [L | mu_disc | C_disc | disc] (offsets as in KgRec).
// EXACT (r6): the Armijo trials of a bracket in one sweep, each computed as a single-trial pass computes it (eval_multi_exact) -- the
// instantiation of the small shapes (kg.hip: small_lane), whose passes were single-trial until then; same bits.
template <int DP, int G, bool EXACT>
__device__ __forceinline__ void kg_sample_lane(const KgMcParams& P, int e, int sl, const double* __restrict__ xs,
                                               double* __restrict__ aw, double* __restrict__ zb, const double* __restrict__ etab,
                                               const double* __restrict__ cst, const double* __restrict__ rc, int lane,
                                               unsigned int& tot_val, unsigned int& tot_grad) {
  typedef const volatile __attribute__((address_space(3))) double* lds_ro_ptr;
  const int m = P.m, u = P.u, n = P.n, g1 = 1 + P.g;
  const int s = P.first_sample + sl;  // global sample index
  const int size = P.dim - P.f;       // problem size of the inner optimisation
  const int ntiles = P.ntiles;
  const double* We = P.W + (long)e * P.w_stride;
  lds_ro_ptr C = (lds_ro_ptr)cst;
  lds_ro_ptr RC = (lds_ro_ptr)rc;
  lds_rw_ptr Z = (lds_rw_ptr)zb;
  const long so = (long)e * P.num_local + sl;

  // ---- z (antithetic pairs, .cpp:171-180) and beta = L^-T z: lane c owns component c (draw_z_beta's recurrence, L from LDS) ----
  const double sign = (s & 1) ? -1.0 : 1.0;
  double zc = 0.0;
  if (lane < m) zc = sign * P.normals[(long)(s >> 1) * m + lane];
  // (the K^-1 y / W loads of the slab's first tiles are independent of z: the compiler is free to issue them here)
  double bc = 0.0;
  {
    double acc = 0.0;
    const int lm = min(lane, m - 1) * m;
    for (int r = m - 1; r >= 0; --r) {
      const double Lr = RC[P.rec.L + r + lm];  // L[r][lane] (col-major): the diagonal on lane r, row r's entries on the lanes below
      if (lane == r) bc = (zc - acc) / Lr;
      const double br = lane_value(bc, r);
      if (lane < r) acc = fma(Lr, br, acc);
    }
  }
  Z[lane] = zc;  // kMaxM == 64 == wavefront size
  Z[kMaxM + lane] = bc;

  if (G == 0 && g1 == 1 && m <= 4) {
    const double b0 = Z[kMaxM], b1 = Z[kMaxM + 1], b2 = Z[kMaxM + 2], b3 = Z[kMaxM + 3];  // 0 beyond m
    const long c1 = (long)min(1, m - 1) * P.N, c2 = (long)min(2, m - 1) * P.N, c3 = (long)min(3, m - 1) * P.N;
    for (int t0 = 0; t0 < ntiles; t0 += 8) {
      double ky[8], l0[8], l1[8], l2[8], l3[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const long row = min(min(t0 + i, ntiles - 1) * 64 + lane, n - 1);  // clamped: always a valid address
        ky[i] = P.KinvY[row];
        l0[i] = We[row];
        l1[i] = We[row + c1];
        l2[i] = We[row + c2];
        l3[i] = We[row + c3];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int t = t0 + i;
        if (t < ntiles) {
          const int j = t * 64 + lane;
          double v = ky[i];
          v = fma(-l0[i], b0, v);
          v = fma(-l1[i], b1, v);
          v = fma(-l2[i], b2, v);
          v = fma(-l3[i], b3, v);
          if (j >= n) v = (j < n + u) ? Z[kMaxM + min(j - n, kMaxM - 1)] : 0.0;
          aw[(long)t * 64 + lane] = v * P.alpha;
        }
      }
    }
  } else {
#pragma unroll 2
    for (int t = 0; t < ntiles; ++t) {
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
              v = fma(-l0, Z[kMaxM + c0], v);
              v = fma(-l1, Z[kMaxM + min(c0 + 1, kMaxM - 1)], v);
              v = fma(-l2, Z[kMaxM + min(c0 + 2, kMaxM - 1)], v);
              v = fma(-l3, Z[kMaxM + min(c0 + 3, kMaxM - 1)], v);
            }
          } else if (j < n + u) {
            v = Z[kMaxM + (j - n) * g1 + a];
          }
          // fold alpha and, for derivative weights, the -(frame scale) of the derivative row (radial3)
          v *= (a == 0) ? P.alpha : -P.alpha * C[a > 0 ? a - 1 : 0];
        }
        w[a * 64] = v;
      }
    }
  }

  // ---- discretised-set scan (.cpp:436-449): f_j = -(mu_n(x_j) + c_j . z); the FIRST best point starts the line search ----
  int best_j = 0;
  {
    double best_f = -INFINITY;
    for (int j0 = 0; j0 < P.A; j0 += 64) {
      const int j = j0 + lane;
      const int jc = min(j, P.A - 1);
      double v = RC[P.rec.mu_disc + jc];
      for (int c = 0; c < m; ++c) v = fma(RC[P.rec.C_disc + jc * m + c], Z[c], v);
      const double fj = (j < P.A) ? -v : -INFINITY;
      const double wmax = wave_max_uniform(fj);
      const unsigned long long ballot = __ballot(fj == wmax);
      const int first_lane = __ffsll((long long)ballot) - 1;
      if (wmax > best_f) {  // strict: an earlier chunk wins ties (priority-queue semantics of .cpp:440-447)
        best_f = wmax;
        best_j = j0 + first_lane;
      }
    }
    best_j = __builtin_amdgcn_readfirstlane(best_j);
  }

  // ---- the line search, lane r < DP holding table row r (= original dimension perm_l) ----
  const int lk = lane < DP ? lane : 0;
  const bool in_l = lane < DP;
  const double s_l = C[lk], is_l = C[DP + lk], c_l = C[2 * DP + lk], pin_l = C[3 * DP + lk];
  const double lo_l = C[4 * DP + lk], hi_l = C[5 * DP + lk];
  const int perm_l = (int)C[6 * DP + lk];
  const bool free_l = in_l && C[7 * DP + lk] != 0.0;
  // start: the discretised point's coordinates on the optimised rows, 1 on fidelity rows, 0 on pads (.cpp:353-357)
  double xo_l = (perm_l < size) ? RC[P.rec.disc + best_j * size + min(perm_l, size - 1)] : pin_l;
  double fcur = 0.0;
  unsigned int n_val = 0, n_grad = 0;  // passes over the n + u points
  const int max_num_steps = P.max_num_steps, max_num_restarts = P.max_num_restarts;
  if (max_num_restarts <= 0) {
    // reference returns without touching its outputs (.cpp:425-427): value 0, point filled with 1.0 (.cpp:163)
    xo_l = pin_l;
  } else {
    const double tolerance = P.tolerance;
    const double step_tolerance = tolerance / (double)max_num_steps;
    const double mean = P.mean;
    const int cov_type = P.cov_type;
    // rows of the wave's scratch (z / beta are spent): three broadcast rows of kMaxLaneDP + 1, the gradient sums and their dump area
    lds_rw_ptr R0 = Z, R2 = Z + (kMaxLaneDP + 1), R3 = Z + 2 * (kMaxLaneDP + 1), R1 = Z + 3 * (kMaxLaneDP + 1);
    static_assert(3 * (kMaxLaneDP + 1) + 2 * kMaxLaneDP + 4 <= 2 * kMaxM, "line-search rows exceed the wave's scratch");
    double xf_l = (xo_l - c_l) * s_l;  // the iterate in the frame: it only feeds the evaluator (see line_search_frame)
    double gf_l = 0.0;
    bool have_g = false;  // a clamped step's f(x + step) and the next iteration's gradient are ONE pass, carried over
    double f_carried = 0.0, g_carried_l = 0.0;
    int pred = 1;  // Armijo trials the previous step consumed
    for (int restart = 0; restart < max_num_restarts; ++restart) {
      const double xstart_l = xf_l;
      for (int istep = 0; istep < max_num_steps;) {
        // ---- f(x), grad f(x) ----
        double f0;
        if (have_g) {
          f0 = f_carried;
          gf_l = g_carried_l;
          have_g = false;
        } else {
          double xq[DP];
          lane_broadcast<DP>(xf_l, R0, lane, xq);
          make_scalar<DP>(xq);
          f0 = (cov_type == MOE_COV_SQUARE_EXPONENTIAL)
                   ? grad_pass_parked<DP, G, MOE_COV_SQUARE_EXPONENTIAL>(xs, aw, etab, ntiles, mean, xq, R1, lane, gf_l)
                   : grad_pass_parked<DP, G, MOE_COV_MATERN_NU_2P5>(xs, aw, etab, ntiles, mean, xq, R1, lane, gf_l);
        }
        f0 = uniform(f0);  // (tells the compiler: the Armijo decisions below are scalar branches)
        n_grad++;
        fcur = f0;
        // the gradient in the original units (pinned rows 0), d2 = -2 g s, x2 = -2 x' (line_search_frame), |g|^2 in row order
        const double g_l = free_l ? gf_l * s_l : 0.0;
        const double d2_l = -2.0 * (g_l * s_l);
        const double x2_l = -2.0 * xf_l;
        double norm = 0.0;
        {
          double gk[DP];
          lane_broadcast<DP>(g_l, R0, lane, gk);
#pragma unroll
          for (int k = 0; k < DP; ++k) norm = fma(gk[k], gk[k], norm);
          norm = uniform(norm);
        }
        double x2[DP], d2[DP];
        lane_broadcast<DP>(x2_l, R2, lane, x2);
        lane_broadcast<DP>(d2_l, R3, lane, d2);
        double sxx = 0.0, sxd = 0.0, sdd = 0.0;  // (fixed along the trial line: formed once per step, eval_multi_loop's order)
#pragma unroll
        for (int k = 0; k < DP; ++k) {
          sxx = fma(x2[k], x2[k], sxx);
          sxd = fma(x2[k], d2[k], sxd);
          sdd = fma(d2[k], d2[k], sdd);
        }
        sxx = uniform(sxx);
        sxd = uniform(sxd);
        sdd = uniform(sdd);
        make_scalar<DP>(x2);
        make_scalar<DP>(d2);
        // pre_mult * (i+1)^-gamma (gpp_optimization.hpp:741); x^-0 == 1 exactly, so gamma == 0 needs no pow()
        double alpha_n = uniform((P.gamma == 0.0) ? P.pre_mult : P.pre_mult * pow((double)(istep + 1), -P.gamma));
        // ---- Armijo back-tracking (.hpp:745-760), several trial step sizes per sweep: line_search_frame's schedule ----
        int search = 0;
        double ftrial = 0.0;
        {
          int batch = pred;
          bool done = false;
          while (!done) {
            bool evaluated = false;
            if (EXACT || P.multi_trial != 0) {
              const int want = min(batch, 30 - search);
              if (want >= 2) {
                const bool se = cov_type == MOE_COV_SQUARE_EXPONENTIAL;
                // T trials in one sweep, then the reference's sequence of decisions over them (gpp_optimization.hpp:752-769): stops at
                // the first accepted one, halves alpha and counts `search` for every rejected one; only consumed trials are counted
#define MOE_LANE_TRIALS(T)                                                                                                              \
  {                                                                                                                                     \
    double ft[T];                                                                                                                       \
    if constexpr (EXACT)                                                                                                                \
      evaluated = se ? eval_multi_exact<DP, MOE_COV_SQUARE_EXPONENTIAL, T, G>(xs, aw, etab, ntiles, mean, x2, d2, alpha_n, lane, ft)    \
                     : eval_multi_exact<DP, MOE_COV_MATERN_NU_2P5, T, G>(xs, aw, etab, ntiles, mean, x2, d2, alpha_n, lane, ft);        \
    else                                                                                                                                \
      evaluated = se ? eval_multi_loop_s<DP, MOE_COV_SQUARE_EXPONENTIAL, T, false, true, G>(xs, aw, etab, ntiles, mean, x2, d2, sxx,    \
                                                                                            sxd, sdd, alpha_n, lane, ft)               \
                     : eval_multi_loop_s<DP, MOE_COV_MATERN_NU_2P5, T, false, true, G>(xs, aw, etab, ntiles, mean, x2, d2, sxx, sxd,    \
                                                                                       sdd, alpha_n, lane, ft);                        \
    if (evaluated) {                                                                                                                    \
      _Pragma("unroll") for (int t = 0; t < T; ++t) {                                                                                   \
        if (!done) {                                                                                                                    \
          ftrial = uniform(ft[t]);                                                                                                            \
          n_val++;                                                                                                                      \
          if (ftrial - f0 > 0.5 * alpha_n * norm) {                                                                                     \
            done = true;                                                                                                                \
          } else {                                                                                                                      \
            alpha_n *= 0.5;                                                                                                             \
            if (++search >= 30) done = true;                                                                                            \
          }                                                                                                                             \
        }                                                                                                                               \
      }                                                                                                                                 \
    }                                                                                                                                   \
  }
                switch (want) {
                  case 2: MOE_LANE_TRIALS(2) break;
                  case 3: MOE_LANE_TRIALS(3) break;
                  case 4: MOE_LANE_TRIALS(4) break;
                  default: MOE_LANE_TRIALS(5) break;
                }
#undef MOE_LANE_TRIALS
              }
            }
            if (!evaluated) {
              double q2[DP], unused[DP];
#pragma unroll
              for (int k = 0; k < DP; ++k) q2[k] = fma(alpha_n, d2[k], x2[k]);
              ftrial = eval_pass<DP, G, false, false, true, true, false>(xs, aw, etab, ntiles, cov_type, mean, q2, nullptr, unused, lane);
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
          pred = min(search + 1, 5);
        }
        // ---- LimitUpdate on the original units, one coordinate per lane, then accept only if f improves (.hpp:762-795) ----
        const double want_o = alpha_n * (gf_l * s_l);  // alpha grad_r (the frame gradient x scale)
        double step_o = 0.0;
        if (free_l) step_o = limit_update_1d(lo_l, hi_l, P.max_relative_change, xo_l, want_o);
        if (P.simplex != 0) {
          // r6: SimplexIntersectTensorProductDomain::LimitUpdate's second half (gpp_domain.cpp:255-289; simplex_limit of kg_mc.hpp) on
          // two rows of the wave's scratch -- the trial line's rows R2 / R3 are spent by now: the point and the clamped step in table-row
          // order (the bounds are the box clipped to the unit hypercube, max_relative_change carries the reference's tweak: kg.hip)
          R2[lane < DP ? lane : DP] = xo_l;
          R3[lane < DP ? lane : DP] = step_o;
          double relaxed, vnorm;
          if (simplex_limit(P, (const double*)R2, (const double*)R3, relaxed, vnorm)) step_o = free_l ? relaxed * (step_o / vnorm) : 0.0;
        }
        const bool changed = __ballot(free_l && step_o != want_o) != 0ull;
        const bool nonzero = __ballot(free_l && step_o != 0.0) != 0ull;
        // the step in the frame: the trial point's own offset where the clamp left it alone (its value is reused below)
        const double step_f = changed ? step_o * s_l : (-0.5 * alpha_n) * d2_l;
        const double st_l = free_l ? step_f : 0.0;
        if (search == 30 || !nonzero) break;  // .hpp:781-785: x restored (a zero step re-evaluates f(x) == f0: rejected)
        double obj2 = ftrial;  // the clamp left the step untouched: f(x + step) is the last trial value
        bool carried = false;
        double gn_l = 0.0;
        if (changed) {
          if (istep + 1 < max_num_steps || restart + 1 < max_num_restarts) {
            double xq[DP];
            lane_broadcast<DP>(xf_l + st_l, R0, lane, xq);
            make_scalar<DP>(xq);
            obj2 = (cov_type == MOE_COV_SQUARE_EXPONENTIAL)
                       ? grad_pass_parked<DP, G, MOE_COV_SQUARE_EXPONENTIAL>(xs, aw, etab, ntiles, mean, xq, R1, lane, gn_l)
                       : grad_pass_parked<DP, G, MOE_COV_MATERN_NU_2P5>(xs, aw, etab, ntiles, mean, xq, R1, lane, gn_l);
            carried = true;
          } else {
            double q2[DP], unused[DP];
            lane_broadcast<DP>(fma(-2.0, st_l, x2_l), R0, lane, q2);
            obj2 = eval_pass<DP, G, false, false, true, true, false>(xs, aw, etab, ntiles, cov_type, mean, q2, nullptr, unused, lane);
            n_val++;
          }
        }
        if (obj2 <= f0) {
          if (carried) n_val++;
          break;
        }
        xf_l += st_l;
        xo_l += step_o;
        double ss = 0.0;  // |step|^2 in the original coordinates, row order
        {
          double so[DP];
          lane_broadcast<DP>(st_l * is_l, R0, lane, so);
#pragma unroll
          for (int k = 0; k < DP; ++k) ss = fma(so[k], so[k], ss);
          ss = uniform(ss);  // (a volatile LDS read counts as divergent: without this the loop exits below are vector branches and
                             //  every scalar of the line search -- alpha, the counters -- lives in vector registers)
        }
        fcur = obj2;
        istep += 1;
        have_g = carried;
        f_carried = obj2;
        g_carried_l = gn_l;
        if (sqrt(ss) < step_tolerance) break;
      }
      double ds = 0.0;
      {
        double dk[DP];
        lane_broadcast<DP>((xstart_l - xf_l) * is_l, R0, lane, dk);
#pragma unroll
        for (int k = 0; k < DP; ++k) ds = fma(dk[k], dk[k], ds);
        ds = uniform(ds);
      }
      if (!(sqrt(ds) > tolerance)) break;
    }
    if (have_g) n_val++;  // carried but never used
    if (!free_l) xo_l = pin_l;
  }

  if (lane == 0) P.best_value[so] = fcur;
  tot_val += n_val;
  tot_grad += n_grad;
  if (in_l) P.best_point[so * DP + perm_l] = xo_l;  // original dimension order (perm is a permutation of the DP rows)
  if (lane < m) P.beta[so * m + lane] = bc;
}

// LDS: [64] exp table | [kLaneCstRows x DP] lane constants | [rec_head] the evaluation's record head | [tab] coordinates | per-wave slabs
// (weights [ntiles (1 + G) 64] + 2 kMaxM doubles of scratch) | one weight tile of padding.  Built for the LDS coordinate table, 8 waves.
// (r6: the body as a device function -- launch.hpp -- so that the members of a GP ensemble share one launch; `argbase`: where this
//  member's arguments lie, the kernarg segment or its record in the ensemble twin's table)
template <int DP, int G, bool EXACT = false>
struct kg_mc_lane_kernel_body {
static __device__ __forceinline__ void run(const VIdx blockIdx, const VIdx gridDim, const void* argbase, const KgMcParams& P, int rec_head) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int ntiles = P.ntiles;
  const int tab = ntiles * (DP + 1) * 64;
  const int wslab = ntiles * (1 + G) * 64 + 2 * kMaxM;
  double* cst = smem + kExpTabLen;
  double* rc = cst + kLaneCstRows * DP;
  double* coords = rc + rec_head;
  double* wshared = coords + tab;
  double* aw = wshared + wave * wslab;
  double* zb = aw + ntiles * (1 + G) * 64;
  if (threadIdx.x < kExpTabLen) smem[threadIdx.x] = kExp2Tab64[threadIdx.x];
  fill_lane_constants<DP>(P, cst);
  for (int e = blockIdx.x % P.E; e < P.E; e += (gridDim.x < (unsigned)P.E ? gridDim.x : P.E)) {
    const double* xg = P.XsTab + (long)e * P.tab_stride;
    __syncthreads();  // previous evaluation's readers are done
    for (int pt = threadIdx.x; pt < ntiles * 64; pt += blockDim.x) {  // one point per thread and step
      const int tl = pt >> 6, l = pt & 63;
      const double* src = xg + (long)tl * DP * 64 + l;
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
    {
      const double* rec = P.blob + (long)e * P.rec.stride;
      for (int i = threadIdx.x; i < rec_head; i += blockDim.x) rc[i] = rec[i];
    }
    __syncthreads();
    unsigned int ticket = 0;
    unsigned int* next = P.next_sample + (long)e * kTicketStride;  // one cache line per evaluation
    if (lane == 0) ticket = atomicAdd(next, 1u);
    unsigned int tot_val = 0, tot_grad = 0;
    while (true) {
      const unsigned int sl = (unsigned int)__builtin_amdgcn_readfirstlane((int)ticket);
      if (sl >= (unsigned int)P.num_local) break;
      if (lane == 0) ticket = atomicAdd(next, 1u);  // drawn ONE AHEAD: the atomic's round trip overlaps the sample
      // The kernel arguments are re-read from the kernarg segment by every sample (scalar loads through an opaque copy of its
      // address) instead of being loaded once at kernel entry and held -- or spilt to vector-register lanes -- for the kernel's lifetime.
      // (P is the kernel's first argument: offset 0 of the segment)
      const __attribute__((address_space(4))) KgMcParams* Pk = (const __attribute__((address_space(4))) KgMcParams*)argbase;
      asm volatile("" : "+s"(Pk));
      kg_sample_lane<DP, G, EXACT>(*(const KgMcParams*)Pk, e, (int)sl, coords, aw, zb, smem, cst, rc, lane, tot_val, tot_grad);
    }
    if (lane == 0 && (tot_val | tot_grad) != 0) {
      atomicAdd(&P.counters[2 * e], (unsigned long long)tot_val);
      atomicAdd(&P.counters[2 * e + 1], (unsigned long long)tot_grad);
    }
    if (gridDim.x >= (unsigned)P.E) break;
  }
}
};
template <int DP, int G, bool EXACT = false>
__global__ __launch_bounds__(kLaneMaxThreads) void kg_mc_lane_kernel(KgMcParams P, int rec_head) {
  kg_mc_lane_kernel_body<DP, G, EXACT>::run(MOE_VBLOCK, MOE_VGRID, (const void*)__builtin_amdgcn_kernarg_segment_ptr(), P, rec_head);
}

// LDS bytes of a workgroup of `waves` wavefronts (host side: kg.hip's geometry)
inline size_t lane_lds_bytes(int dp, int G, int ntiles, int rec_head, int waves) {
  const size_t fixed = (size_t)kExpTabLen + (size_t)kLaneCstRows * dp + (size_t)((rec_head + 1) & ~1);
  const size_t tab = (size_t)ntiles * (dp + 1) * 64;
  return sizeof(double) * (fixed + tab + (size_t)waves * ((size_t)ntiles * (1 + G) * 64 + 2 * kMaxM) + (size_t)(1 + G) * 64);
}

template <int DP, int G, bool EXACT = false>
inline void launch_lane_inst(const KgMcParams& P, int rec_head, int blocks, int waves, size_t shm, hipStream_t s) {
  auto kern = kg_mc_lane_kernel<DP, G, EXACT>;
  MOE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  launch_kernel_ens<kg_mc_lane_kernel_body<DP, G, EXACT>, kLaneMaxThreads>(kern, dim3(blocks), dim3(waves * 64), shm, s, P, rec_head);
  MOE_HIP_CHECK(hipGetLastError());
}

template <int DP>
inline void launch_lane_dp(const KgMcParams& P, int G, int rec_head, int blocks, int waves, size_t shm, hipStream_t s) {
  static_assert(DP <= kMaxLaneDP, "lane-parked line search: one scratch row holds kMaxLaneDP doubles");
  if constexpr (DP == 4) {  // (the small shapes live here: padded dimension 4, one or two tiles -- kg.hip)
    if (P.multi_trial == 2) {
      switch (G) {
        case 0: launch_lane_inst<DP, 0, true>(P, rec_head, blocks, waves, shm, s); break;
        case 1: launch_lane_inst<DP, 1, true>(P, rec_head, blocks, waves, shm, s); break;
        case 2: launch_lane_inst<DP, 2, true>(P, rec_head, blocks, waves, shm, s); break;
        case 3: launch_lane_inst<DP, 3, true>(P, rec_head, blocks, waves, shm, s); break;
        case 4: launch_lane_inst<DP, 4, true>(P, rec_head, blocks, waves, shm, s); break;
        default: throw Error(MOE_ERR_RUNTIME, "unsupported derivative-slot count in the lane-parked MC kernel");
      }
      return;
    }
  }
  if (P.multi_trial == 2) throw Error(MOE_ERR_RUNTIME, "exact multi-trial passes are built for the padded dimension 4 only");
  switch (G) {
    case 0: launch_lane_inst<DP, 0>(P, rec_head, blocks, waves, shm, s); break;
    case 1: launch_lane_inst<DP, 1>(P, rec_head, blocks, waves, shm, s); break;
    case 2: launch_lane_inst<DP, 2>(P, rec_head, blocks, waves, shm, s); break;
    case 3: launch_lane_inst<DP, 3>(P, rec_head, blocks, waves, shm, s); break;
    case 4: launch_lane_inst<DP, 4>(P, rec_head, blocks, waves, shm, s); break;
    default: throw Error(MOE_ERR_RUNTIME, "unsupported derivative-slot count in the lane-parked MC kernel");
  }
}
