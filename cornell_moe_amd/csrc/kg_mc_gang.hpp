// cornell_moe_amd/csrc/kg_mc_gang.hpp -- round 6: the q-KG / d-KG Monte-Carlo kernel for point sets whose per-sample weights fit no
// LDS slab (included by kg_mc.hpp, inside namespace moe::mc).  BASELINE configs[4]: d-KG at n = 2000, d = 12, g = 3 -- 193 KB of
// coordinates, 64 KB of weights per sample (209 KB at g = 12).
//
// The streamed-weights kernel (kg_mc_stream_kernel) gives a sample to ONE wavefront and re-reads the sample's row of the weight
// table in every sweep of its line search: ~14 sweeps x 64 KB x 20 000 samples = 17.8 GB of fabric traffic per evaluation (58 GB at
// g = 12), the sweeps waiting on those loads.  Here a sample belongs to a GANG of W wavefronts of one workgroup (W = 4 at g <= 4:
// two gangs per CU; W = 8 at g = 8 / 12):
//   * wavefront w of the gang owns tiles w, w + W, w + 2W, ... of the n + u points for the sample's lifetime and keeps THEIR weights
//     in registers (TPW tiles x (1 + G) doubles: 64 VGPRs at g = 3, 104 at g = 12), loaded ONCE per sample from the table
//     kg_sample_weights_kernel wrote: table traffic = one write + one read;
//   * the coordinates are shared by every sample: the leading tiles in the workgroup's LDS table (with the |x|^2 row of the
//     dot-product distances, as in the lane-parked kernel), the rest streamed from L2 one tile ahead;
//   * wavefront 0 of a gang -- the LEADER -- runs the sample's lane-parked line search (kg_mc_lane.hpp: lane_line_search), once: for
//     every pass it publishes a command in LDS (kind, trial count, the sample, the wave-uniform operands stored from their
//     lane-parked copies), sweeps its own tiles, waits for the followers' partial sums and adds the W partials in wave order; the
//     FOLLOWERS only loop over { wait for a command, sweep own tiles, packed wave reduction, partial sums + flag }.  The decisions
//     of a sample are taken once, not W times; while a leader decides, its followers' SIMDs belong to the other gang.
//     Synchronisation is a sequence number per direction in LDS (coherent within the CU; every wave of a workgroup is resident), polled
//     with s_sleep -- no workgroup barrier, the gangs of a CU never wait for each other.
// Semantics: gpp_knowledge_gradient_optimization.cpp:164-196, 420-472; gpp_optimization.hpp:708-828; the sums of a pass are the
// streamed-weights kernel's, associated per wavefront and then over the gang (agreement to rounding, not bit for bit).
#pragma once

constexpr int kGangSlot = 32;      // doubles per partial-sum slot: f | DP gradient sums | G derivative sums (<= 1 + 16 + 12), or T <= 5 values
constexpr int kGangMaxWaves = 8;   // wavefronts per workgroup
constexpr int kGangCmd = 40;       // doubles of a gang's command block: [0] alpha0 [1] sxx [2] sxd [3] sdd | [4 ..) vector A | [21 ..) vector B
constexpr int kGangVecB = 4 + kMaxLaneDP + 1;
constexpr int kGangCtl = 16;       // doubles (32 ints) of a gang's control block: [0] command sequence [1] kind [2] trial count [3] sample
                                   // | [8 + p] partial sequence of wave p
constexpr int kGangSpinLimit = 1 << 22;  // polls of a flag before a wavefront gives up (a desynchronised gang must not hang the GPU)
enum : int { kGangGrad = 0, kGangMulti = 1, kGangValue = 2, kGangDone = 3 };

typedef volatile __attribute__((address_space(3))) int* lds_flag_ptr;

// doubles of LDS in front of the coordinate table: exp table | lane constants | partial-sum slots [8 waves][kGangSlot] | scratch
// [8 waves][2 kMaxM] | per gang (8 provided): command block + control block
__host__ __device__ constexpr int gang_fixed_doubles(int dp) {
  return kExpTabLen + kLaneCstRows * dp + kGangMaxWaves * (kGangSlot + 2 * kMaxM) + kGangMaxWaves * (kGangCmd + kGangCtl);
}

// dst[i] = sum over the 64 lanes of v[i] (wave_sum_packed_store of kg_mc.hpp onto an LDS-typed destination)
template <int N>
__device__ __forceinline__ void gang_store_sums(const double (&v)[N], lds_rw_ptr dst, int lane) {
  constexpr int NP = (N + 3) / 4 * 4;
  static_assert(NP <= kGangSlot, "partial sums exceed the slot");
  const int row = lane >> 4;
  const int slot = ((row & 1) << 1) | (row >> 1);  // rows hold v0 | v2 | v1 | v3
#pragma unroll
  for (int i = 0; i < NP; i += 4) {
    const double a = v[i], b = (i + 1 < N) ? v[i + 1 < N ? i + 1 : 0] : 0.0, c = (i + 2 < N) ? v[i + 2 < N ? i + 2 : 0] : 0.0,
                 d = (i + 3 < N) ? v[i + 3 < N ? i + 3 : 0] : 0.0;
    const double q = row_sum(fold16(fold32(a, b), fold32(c, d)));
    if ((lane & 15) == 0) dst[i + slot] = q;  // (entries N .. NP - 1 of the slot receive zeros)
  }
}

// One wavefront's share of a sample: its tiles' weights in registers and the three sweeps over them.
// The tile loops are NOT unrolled: tile t's weights are picked out of the register array by a wave-uniform switch (1 + G moves per
// tile), the coordinates of tile t + 1 are requested as soon as tile t's have been consumed (after the projections / differences:
// one buffer, in flight behind the square roots and exponentials of tile t).
template <int DP, int G, int W, int TPW, int COV>
struct GangTiles {
  static constexpr int XR = DP + 1;  // rows of an LDS tile: the coordinates + |x|^2
  const double* __restrict__ xl;     // LDS table [tile][XR][64] + lane
  const double* __restrict__ xg;     // the evaluation's table in global memory [tile][DP][64] + lane
  const double* __restrict__ etab;
  int w;     // wave of the gang
  int nt;    // tiles of this wave: w, w + W, ... (<= TPW)
  int ntl;   // the first `ntl` of them are in the LDS table
  int lane;
  double wr[TPW][1 + G];  // this wave's weights of the current sample (alpha-scaled; zero for padded points)

  // coordinates (+ |x|^2) of this wave's tile t: from LDS, or from L2 with the |x|^2 row formed in the LDS copy's order
  __device__ __forceinline__ void load_tile(int t, double (&cx)[XR]) const {
    const int tile = w + t * W;
    if (t < ntl) {
      lds_tile_ptr p = (lds_tile_ptr)(xl + (long)tile * XR * 64);
#pragma unroll
      for (int k = 0; k < XR; ++k) cx[k] = p[k * 64];
    } else {
      const double* p = xg + (long)tile * DP * 64;
      double xx = 0.0;
#pragma unroll
      for (int k = 0; k < DP; ++k) cx[k] = p[k * 64];
#pragma unroll
      for (int k = 0; k < DP; ++k) xx = fma(cx[k], cx[k], xx);
      cx[DP] = xx;
    }
  }

  // the weights of tile t (wave-uniform t): a scalar branch per case, 1 + G register moves
  __device__ __forceinline__ void pick_weights(int t, double (&cw)[1 + G]) const {
#define MOE_GANG_PICK(I)                                     \
  case I:                                                    \
    _Pragma("unroll") for (int a = 0; a < 1 + G; ++a) cw[a] = wr[(I) < TPW ? (I) : 0][a]; \
    asm volatile("" ::: "memory");                           \
    break;
    switch (t) {
      MOE_GANG_PICK(0)
      MOE_GANG_PICK(1)
      MOE_GANG_PICK(2)
      MOE_GANG_PICK(3)
      MOE_GANG_PICK(4)
      MOE_GANG_PICK(5)
      MOE_GANG_PICK(6)
      default:
#pragma unroll
        for (int a = 0; a < 1 + G; ++a) cw[a] = wr[TPW - 1][a];
        break;
    }
#undef MOE_GANG_PICK
  }

  // this sample's weights of my tiles from its row of the table (point-major, 1 + G doubles per point)
  __device__ __forceinline__ void load_weights(const double* __restrict__ row) {
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int tile = w + min(t, max(nt - 1, 0)) * W;  // clamped: always a valid address; values of t >= nt are never used
      const double* p = row + ((long)tile * 64 + lane) * (1 + G);
      if constexpr (((1 + G) & 1) == 0) {
        const d2t* q = reinterpret_cast<const d2t*>(p);
#pragma unroll
        for (int a2 = 0; a2 < (1 + G) / 2; ++a2) {
          const d2t v = q[a2];
          wr[t][2 * a2] = v.x;
          wr[t][2 * a2 + ((1 + G) > 1 ? 1 : 0)] = v.y;
        }
      } else {
#pragma unroll
        for (int a = 0; a < 1 + G; ++a) wr[t][a] = p[a];
      }
    }
  }

  // value + gradient sweep at xq (grad_pass_parked's arithmetic per point); partial sums [f | DP gradient sums | G derivative sums]
  __device__ __forceinline__ void sweep_grad(const double (&xq)[DP], lds_rw_ptr slot) const {
    double accf = 0.0;
    double accg[DP];
    double accd[G > 0 ? G : 1];
#pragma unroll
    for (int k = 0; k < DP; ++k) accg[k] = 0.0;
#pragma unroll
    for (int a = 0; a < (G > 0 ? G : 1); ++a) accd[a] = 0.0;
    double cx[XR];
    if (nt > 0) load_tile(0, cx);
#pragma unroll 1
    for (int t = 0; t < nt; ++t) {
      double cw[1 + G];
      pick_weights(t, cw);
      double diff[DP];
      double r2 = 1.0e-300;
#pragma unroll
      for (int k = 0; k < DP; ++k) {
        diff[k] = cx[k] - xq[k];
        r2 = fma(diff[k], diff[k], r2);
      }
      if (t + 1 < nt) load_tile(t + 1, cx);  // (this tile's coordinates are spent)
      const double w0 = cw[0];
      double base, first, second;
      radial3<COV, true, (G > 0)>(r2, etab, base, first, second);
      double sd = 0.0;
      if (G > 0) {
#pragma unroll
        for (int a = 0; a < G; ++a) sd = fma(cw[1 + (a < G ? a : 0)], diff[a], sd);
      }
      accf = fma(w0, base, accf);
      if (G > 0) accf = fma(first, sd, accf);
      double coef = w0 * first;
      if (G > 0) {
        coef = fma(second, sd, coef);
#pragma unroll
        for (int a = 0; a < G; ++a) accd[a] = fma(first, cw[1 + (a < G ? a : 0)], accd[a]);
      }
#pragma unroll
      for (int k = 0; k < DP; ++k) accg[k] = fma(coef, diff[k], accg[k]);
    }
    constexpr int NS = 1 + DP + (G > 0 ? G : 0);
    double sums[NS];
    sums[0] = accf;
#pragma unroll
    for (int k = 0; k < DP; ++k) sums[1 + k] = accg[k];
    if (G > 0) {
#pragma unroll
      for (int a = 0; a < G; ++a) sums[1 + DP + a] = accd[a];
    }
    gang_store_sums<NS>(sums, slot, lane);
  }

  // T Armijo trials in one sweep (eval_multi_loop_s's arithmetic per point: dot-product distances from the |x|^2 row)
  template <int T>
  __device__ __forceinline__ void sweep_multi(const double (&x2)[DP], const double (&d2)[DP], double sxx, double sxd, double sdd,
                                              double alpha0, lds_rw_ptr slot) const {
    double al[T], qq[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      al[t] = (t == 0) ? alpha0 : 0.5 * al[t > 0 ? t - 1 : 0];
      qq[t] = fma(0.25, fma(al[t], fma(al[t], sdd, 2.0 * sxd), sxx), 1.0e-300);
    }
    double acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = 0.0;
    double x0[G > 0 ? G : 1], dv[G > 0 ? G : 1];
    if (G > 0) {
#pragma unroll
      for (int a = 0; a < G; ++a) {
        x0[a] = -0.5 * x2[a];
        dv[a] = -0.5 * d2[a];
      }
    }
    double cx[XR];
    if (nt > 0) load_tile(0, cx);
#pragma unroll 1
    for (int tile = 0; tile < nt; ++tile) {
      double cwa[1 + G];
      pick_weights(tile, cwa);
      const double cw = cwa[0];
      double sdA = 0.0, sdB = 0.0;
      if (G > 0) {
#pragma unroll
        for (int a = 0; a < G; ++a) {
          sdA = fma(cwa[1 + (a < G ? a : 0)], cx[a] - x0[a], sdA);
          sdB = fma(cwa[1 + (a < G ? a : 0)], dv[a], sdB);
        }
      }
      double p0 = cx[DP];
#pragma unroll
      for (int k = 0; k < DP; ++k) p0 = fma(cx[k], x2[k], p0);
      double p1 = cx[0] * d2[0];
#pragma unroll
      for (int k = 1; k < DP; ++k) p1 = fma(cx[k], d2[k], p1);
      if (tile + 1 < nt) load_tile(tile + 1, cx);  // (this tile's coordinates are spent)
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const double r2 = fmax(fma(al[t], p1, p0 + qq[t]), 1.0e-300);
        double base, first, second;
        radial3<COV, (G > 0), false>(r2, etab, base, first, second);
        acc[t] = fma(cw, base, acc[t]);
        if (G > 0) acc[t] = fma(first, fma(-al[t], sdB, sdA), acc[t]);
      }
    }
    gang_store_sums<T>(acc, slot, lane);
  }

  // f at the frame point -q2 / 2 (eval_loop's Q2IN value pass)
  __device__ __forceinline__ void sweep_value(const double (&q2)[DP], lds_rw_ptr slot) const {
    double ss = 0.0;
#pragma unroll
    for (int k = 0; k < DP; ++k) ss = fma(q2[k], q2[k], ss);
    const double qq = fma(ss, 0.25, 1.0e-300);
    double xq[G > 0 ? G : 1];
    if (G > 0) {
#pragma unroll
      for (int a = 0; a < G; ++a) xq[a] = -0.5 * q2[a];
    }
    double accf = 0.0;
    double cx[XR];
    if (nt > 0) load_tile(0, cx);
#pragma unroll 1
    for (int t = 0; t < nt; ++t) {
      double cw[1 + G];
      pick_weights(t, cw);
      double r2 = cx[DP] + qq;
#pragma unroll
      for (int k = 0; k < DP; ++k) r2 = fma(cx[k], q2[k], r2);
      r2 = fmax(r2, 1.0e-300);
      double sd = 0.0;
      if (G > 0) {
#pragma unroll
        for (int a = 0; a < G; ++a) sd = fma(cw[1 + (a < G ? a : 0)], cx[a] - xq[a], sd);
      }
      if (t + 1 < nt) load_tile(t + 1, cx);
      double base, first, second;
      radial3<COV, (G > 0), false>(r2, etab, base, first, second);
      accf = fma(cw[0], base, accf);
      if (G > 0) accf = fma(first, sd, accf);
    }
    const double sums[1] = {accf};
    gang_store_sums<1>(sums, slot, lane);
  }
};

// What the wavefronts of a gang share in LDS.
struct GangShared {
  lds_rw_ptr cmd;      // the gang's command block (kGangCmd doubles)
  lds_flag_ptr ctl;    // the gang's control block (see kGangCtl)
  lds_rw_ptr slots;    // the partial-sum slots of the gang's waves: wave p at slots + p kGangSlot
};

// polls ctl[idx] until it has reached `want`; false = gave up
__device__ __forceinline__ bool gang_wait(lds_flag_ptr ctl, int idx, int want) {
  int spins = 0;
  while (true) {
    const int fv = __builtin_amdgcn_readfirstlane(ctl[idx]);
    if (fv - want >= 0) return true;
    __builtin_amdgcn_s_sleep(1);
    if (++spins > kGangSpinLimit) return false;
  }
}

// The leader's evaluator (see lane_line_search for the interface): publish the pass, sweep own tiles, collect.
template <int DP, int G, int W, int TPW, int COV>
struct GangLeader {
  GangTiles<DP, G, W, TPW, COV>* tiles;
  GangShared sh;
  int lane;
  double mean;
  mutable int seq;    // passes published so far
  mutable bool dead;  // a follower did not answer within kGangSpinLimit polls: results are poisoned, no further waiting
  int sample;         // what the followers need to find the sample's weights

  __device__ __forceinline__ void publish(int kind, int T) const {
    seq += 1;
    if (lane == 0) {
      sh.ctl[1] = kind;
      sh.ctl[2] = T;
      sh.ctl[3] = sample;
      sh.ctl[0] = seq;  // (LDS operations of a wave complete in order: whoever sees the sequence number sees the command)
    }
  }
  __device__ __forceinline__ void collect() const {
    if (dead) return;
#pragma unroll
    for (int p = 1; p < W; ++p)
      if (!gang_wait(sh.ctl, 8 + p, seq)) dead = true;
  }
  // the gang's sum of entry `idx` (may differ from lane to lane), added in wave order
  __device__ __forceinline__ double total(int idx) const {
    double t = 0.0;
#pragma unroll
    for (int p = 0; p < W; ++p) t += sh.slots[p * kGangSlot + idx];
    return t;
  }

  __device__ __forceinline__ double grad(const double (&xq)[DP], double xq_l, lds_rw_ptr, double& g_l) const {
    sh.cmd[4 + (lane < DP ? lane : DP)] = xq_l;
    publish(kGangGrad, 0);
    tiles->sweep_grad(xq, sh.slots);
    collect();
    const double f = total(0);
    double v = total(1 + (lane < DP ? lane : 0));
    if (G > 0) {
      const double vd = total(1 + DP + (lane < G ? lane : 0));
      if (lane < G) v -= vd;
    }
    g_l = dead ? __builtin_nan("") : -v;
    return -(mean + uniform(f));
  }

  template <int T>
  __device__ __forceinline__ bool multi(const double (&x2)[DP], const double (&d2)[DP], double x2_l, double d2_l, double sxx, double sxd,
                                        double sdd, double alpha0, double (&f)[T]) const {
    // (the far-field test of eval_multi_loop_s: |q(alpha)|^2 is convex in alpha, the two ends bound every trial)
    const double qq0 = fma(0.25, fma(alpha0, fma(alpha0, sdd, 2.0 * sxd), sxx), 1.0e-300);
    if (!(uniform(fmax(qq0, 0.25 * sxx)) <= kFarRadius * kFarRadius)) return false;
    sh.cmd[4 + (lane < DP ? lane : DP)] = x2_l;
    sh.cmd[kGangVecB + (lane < DP ? lane : DP)] = d2_l;
    if (lane < 4) sh.cmd[lane] = (lane == 0) ? alpha0 : (lane == 1) ? sxx : (lane == 2) ? sxd : sdd;
    publish(kGangMulti, T);
    tiles->template sweep_multi<T>(x2, d2, sxx, sxd, sdd, alpha0, sh.slots);
    collect();
#pragma unroll
    for (int t = 0; t < T; ++t) f[t] = dead ? __builtin_nan("") : -(mean + total(t));
    return true;
  }

  __device__ __forceinline__ double value(const double (&q2)[DP], double q2_l) const {
    double ss = 0.0;
#pragma unroll
    for (int k = 0; k < DP; ++k) ss = fma(q2[k], q2[k], ss);
    if (!(uniform(fma(ss, 0.25, 1.0e-300)) <= kFarRadius * kFarRadius)) return -mean;
    sh.cmd[4 + (lane < DP ? lane : DP)] = q2_l;
    publish(kGangValue, 0);
    tiles->sweep_value(q2, sh.slots);
    collect();
    return dead ? __builtin_nan("") : -(mean + total(0));
  }
};

// wave-uniform vector of a command as scalar operands
template <int DP>
__device__ __forceinline__ void gang_read_vec(lds_rw_ptr p, double (&out)[DP]) {
#pragma unroll
  for (int k = 0; k < DP; ++k) out[k] = p[k];
  make_scalar<DP>(out);
}

// A follower's life for one evaluation: serve the leader's passes until it says done.
template <int DP, int G, int W, int TPW, int COV>
__device__ __forceinline__ void gang_follow(const KgMcParams& P, int e, GangTiles<DP, G, W, TPW, COV>& tiles, const GangShared& sh, int& seq,
                                            int lane) {
  int cur = -1;
  lds_rw_ptr slot = sh.slots + tiles.w * kGangSlot;
  while (true) {
    if (!gang_wait(sh.ctl, 0, seq + 1)) return;  // (the leader is gone: leave; it poisons its own results)
    seq += 1;
    const int kind = __builtin_amdgcn_readfirstlane(sh.ctl[1]);
    if (kind == kGangDone) return;
    const int T = __builtin_amdgcn_readfirstlane(sh.ctl[2]);
    const int sl = __builtin_amdgcn_readfirstlane(sh.ctl[3]);
    if (sl != cur) {
      cur = sl;
      tiles.load_weights(P.V + ((long)e * P.num_local + sl) * P.v_stride);
    }
    if (kind == kGangGrad) {
      double xq[DP];
      gang_read_vec<DP>(sh.cmd + 4, xq);
      tiles.sweep_grad(xq, slot);
    } else if (kind == kGangMulti) {
      double x2[DP], d2[DP];
      gang_read_vec<DP>(sh.cmd + 4, x2);
      gang_read_vec<DP>(sh.cmd + kGangVecB, d2);
      const double alpha0 = uniform(sh.cmd[0]), sxx = uniform(sh.cmd[1]), sxd = uniform(sh.cmd[2]), sdd = uniform(sh.cmd[3]);
      switch (T) {
        case 2: tiles.template sweep_multi<2>(x2, d2, sxx, sxd, sdd, alpha0, slot); break;
        case 3: tiles.template sweep_multi<3>(x2, d2, sxx, sxd, sdd, alpha0, slot); break;
        case 4: tiles.template sweep_multi<4>(x2, d2, sxx, sxd, sdd, alpha0, slot); break;
        default: tiles.template sweep_multi<5>(x2, d2, sxx, sxd, sdd, alpha0, slot); break;
      }
    } else {
      double q2[DP];
      gang_read_vec<DP>(sh.cmd + 4, q2);
      tiles.sweep_value(q2, slot);
    }
    if (lane == 0) sh.ctl[8 + tiles.w] = seq;  // (after the partial sums: in-order LDS)
  }
}

// LDS: [64] exp table | [kLaneCstRows x DP] lane constants | partial-sum slots | per-wave scratch | per gang: command + control | the
// leading `lds_tiles` tiles of the coordinate table [tile][DP + 1][64].  Needs the sample pre-pass (P.best_j, P.beta) and the weight
// table P.V (v_slots1 == 1 + G, whole tiles); the coordinate table P.XsTab in the plain [tile][DP][64] layout.
#ifndef MOE_GANG_THREADS
#define MOE_GANG_THREADS 512
#endif
template <int DP, int G, int W, int TPW, int COV>
__global__ __launch_bounds__(MOE_GANG_THREADS) void kg_mc_gang_kernel(KgMcParams P, int lds_tiles) {
  static_assert(kGangMaxWaves % W == 0, "gangs tile the workgroup");
  static_assert(8 + W <= 2 * kGangCtl, "control block");
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int gang = wave / W, w = wave % W;
  const int ntiles = P.ntiles;
  double* cst = smem + kExpTabLen;
  double* all_slots = cst + kLaneCstRows * DP;                   // [8][kGangSlot]
  double* all_scratch = all_slots + kGangMaxWaves * kGangSlot;   // [8][2 kMaxM]
  double* per_gang = all_scratch + kGangMaxWaves * 2 * kMaxM;    // [8][kGangCmd + kGangCtl]
  double* coords = smem + gang_fixed_doubles(DP);
  GangShared sh;
  sh.cmd = (lds_rw_ptr)(per_gang + gang * (kGangCmd + kGangCtl));
  sh.ctl = (lds_flag_ptr) reinterpret_cast<int*>(per_gang + gang * (kGangCmd + kGangCtl) + kGangCmd);
  sh.slots = (lds_rw_ptr)(all_slots + gang * W * kGangSlot);
  double* scratch = all_scratch + wave * 2 * kMaxM;
  if (threadIdx.x < kExpTabLen) smem[threadIdx.x] = kExp2Tab64[threadIdx.x];
  if (threadIdx.x < kGangMaxWaves * 2 * kGangCtl) {  // every gang's control ints
    const int gi = threadIdx.x / (2 * kGangCtl), ii = threadIdx.x % (2 * kGangCtl);
    reinterpret_cast<int*>(per_gang + gi * (kGangCmd + kGangCtl) + kGangCmd)[ii] = 0;
  }
  fill_lane_constants<DP>(P, cst);
  GangTiles<DP, G, W, TPW, COV> tiles;
  tiles.etab = smem;
  tiles.w = w;
  tiles.lane = lane;
  tiles.nt = (w < ntiles) ? min((ntiles - w + W - 1) / W, TPW) : 0;
  tiles.ntl = (w < lds_tiles) ? min((lds_tiles - w + W - 1) / W, tiles.nt) : 0;
  int seq = 0;  // this wave's count of the gang's passes (leader: published; follower: served)
  bool dead = false;
  const int size = P.dim - P.f;
  for (int e = blockIdx.x % P.E; e < P.E; e += (gridDim.x < (unsigned)P.E ? gridDim.x : P.E)) {
    const double* xg = P.XsTab + (long)e * P.tab_stride;
    __syncthreads();  // previous evaluation's readers are done (and the control blocks are cleared)
    for (int pt = threadIdx.x; pt < lds_tiles * 64; pt += blockDim.x) {  // one point per thread and step
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
    __syncthreads();
    tiles.xl = coords + lane;
    tiles.xg = xg + lane;
    if (w != 0) {
      gang_follow<DP, G, W, TPW, COV>(P, e, tiles, sh, seq, lane);
    } else {
      typedef const volatile __attribute__((address_space(3))) double* lds_ro_ptr;
      lds_ro_ptr C = (lds_ro_ptr)cst;
      const int lk = lane < DP ? lane : 0;
      const bool in_l = lane < DP;
      const int perm_l = (int)C[6 * DP + lk];
      const double pin_l = C[3 * DP + lk];
      GangLeader<DP, G, W, TPW, COV> ps{&tiles, sh, lane, P.mean, seq, dead, 0};
      // sample tickets drawn ONE AHEAD: the atomic's round trip overlaps the current sample
      unsigned int ticket = 0;
      unsigned int* next = P.next_sample + (long)e * kTicketStride;
      if (lane == 0) ticket = atomicAdd(next, 1u);
      unsigned int tot_val = 0, tot_grad = 0;
      while (true) {
        const unsigned int sl = (unsigned int)__builtin_amdgcn_readfirstlane((int)ticket);
        if (sl >= (unsigned int)P.num_local) break;
        if (lane == 0) ticket = atomicAdd(next, 1u);
        const long so = (long)e * P.num_local + sl;
        ps.sample = (int)sl;
        tiles.load_weights(P.V + so * P.v_stride);
        const int best_j = P.best_j[so];
        const double* disc = P.blob + (long)e * P.rec.stride + P.rec.disc;
        // start: the discretised point's coordinates on the optimised rows, 1 on fidelity rows, 0 on pads (.cpp:353-357)
        double xo_l = (perm_l < size) ? disc[(long)best_j * size + min(perm_l, size - 1)] : pin_l;
        double fcur = 0.0;
        unsigned int n_val = 0, n_grad = 0;
        lane_line_search<DP>(P, ps, C, (lds_rw_ptr)scratch, lane, xo_l, fcur, n_val, n_grad);
        if (lane == 0) P.best_value[so] = ps.dead ? __builtin_nan("") : fcur;
        if (in_l) P.best_point[so * DP + perm_l] = xo_l;  // original dimension order
        tot_val += n_val;
        tot_grad += n_grad;
      }
      ps.publish(kGangDone, 0);
      seq = ps.seq;
      dead = ps.dead;
      if (lane == 0 && (tot_val | tot_grad) != 0) {
        atomicAdd(&P.counters[2 * e], (unsigned long long)tot_val);
        atomicAdd(&P.counters[2 * e + 1], (unsigned long long)tot_grad);
      }
    }
    if (gridDim.x >= (unsigned)P.E) break;
  }
}

template <int DP, int G, int W, int TPW>
inline void launch_gang_inst(const KgMcParams& P, int lds_tiles, int blocks, size_t shm, hipStream_t s) {
  if (P.cov_type == MOE_COV_SQUARE_EXPONENTIAL) {
    auto kern = kg_mc_gang_kernel<DP, G, W, TPW, MOE_COV_SQUARE_EXPONENTIAL>;
    MOE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(kGangMaxWaves * 64), shm, s, P, lds_tiles);
  } else {
    auto kern = kg_mc_gang_kernel<DP, G, W, TPW, MOE_COV_MATERN_NU_2P5>;
    MOE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(kGangMaxWaves * 64), shm, s, P, lds_tiles);
  }
  MOE_HIP_CHECK(hipGetLastError());
}

// Built for: up to 4 derivative slots with W = 4 (n + u <= 2048 points) or W = 8 (<= 4096), TPW = 8; 8 and 12 slots with W = 8, TPW = 4
// (<= 2048 points).  kg.hip picks W from the evaluation's shape alone.
template <int DP>
inline void launch_gang_dp(const KgMcParams& P, int G, int W, int lds_tiles, int blocks, size_t shm, hipStream_t s) {
  static_assert(DP <= kMaxLaneDP, "lane-parked line search: one scratch row holds kMaxLaneDP doubles");
  if (W == 4) {
    switch (G) {
      case 0: launch_gang_inst<DP, 0, 4, 8>(P, lds_tiles, blocks, shm, s); return;
      case 1: launch_gang_inst<DP, 1, 4, 8>(P, lds_tiles, blocks, shm, s); return;
      case 2: launch_gang_inst<DP, 2, 4, 8>(P, lds_tiles, blocks, shm, s); return;
      case 3: launch_gang_inst<DP, 3, 4, 8>(P, lds_tiles, blocks, shm, s); return;
      case 4: launch_gang_inst<DP, 4, 4, 8>(P, lds_tiles, blocks, shm, s); return;
      default: break;
    }
  } else if (W == 8) {
    switch (G) {
      case 0: launch_gang_inst<DP, 0, 8, 8>(P, lds_tiles, blocks, shm, s); return;
      case 1: launch_gang_inst<DP, 1, 8, 8>(P, lds_tiles, blocks, shm, s); return;
      case 2: launch_gang_inst<DP, 2, 8, 8>(P, lds_tiles, blocks, shm, s); return;
      case 3: launch_gang_inst<DP, 3, 8, 8>(P, lds_tiles, blocks, shm, s); return;
      case 4: launch_gang_inst<DP, 4, 8, 8>(P, lds_tiles, blocks, shm, s); return;
      case 8:
        if constexpr (DP >= 8) {
          launch_gang_inst<DP, 8, 8, 4>(P, lds_tiles, blocks, shm, s);
          return;
        }
        break;
      case 12:
        if constexpr (DP >= 12) {
          launch_gang_inst<DP, 12, 8, 4>(P, lds_tiles, blocks, shm, s);
          return;
        }
        break;
      default: break;
    }
  }
  throw Error(MOE_ERR_RUNTIME, "unsupported shape in the gang MC kernel");
}
