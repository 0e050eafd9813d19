// cornell_moe_amd/csrc/mcmc.hip -- MCMC-averaged acquisition evaluators: the Bayesian treatment of the GP
// hyper-parameters the reference's examples actually run -- num_mcmc GPs over the same data, one per hyper-parameter sample
// (GaussianProcessMCMC, gpp_knowledge_gradient_mcmc_optimization.cpp:24-49), acquisition = the average of the per-GP
// acquisitions, KG additionally divided by the fidelity cost:
//   * KnowledgeGradientMCMCEvaluator::Compute[Grad]KnowledgeGradient, ComputeCost, ComputeGradCost (.cpp:84-180);
//   * ExpectedImprovementMCMCEvaluator / OnePotentialSampleExpectedImprovementMCMCEvaluator
//     (gpp_expected_improvement_mcmc_optimization.cpp:48-88, 136-176);
//   * ComputeKGMCMCOptimalPointsToSampleViaMultistartGradientDescent / EvaluateKGMCMCAtPointList
//     (gpp_knowledge_gradient_mcmc_optimization.hpp:665-862) and the EI twins (gpp_expected_improvement_mcmc_optimization.hpp).
// Every per-GP evaluator replays the same normal stream (each rewinds the shared RNG before it draws,
// gpp_knowledge_gradient_optimization.cpp:78, 139), so one table serves all GPs.  The GP index is a third independent shard
// axis: kg_mcmc_sums returns plain sums over the GPs it was given so ranks holding disjoint GP subsets can all-reduce them
// before kg_mcmc_finalize (cornell_moe_amd/dist.py).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>

#include "gp.hpp"
#include "kg.hpp"

#include <cstdlib>
#include <map>
#include <thread>

namespace moe {

namespace {

void check_ensemble(const std::vector<GpDev*>& gps) {
  if (gps.empty()) throw Error(MOE_ERR_BOUNDS, "num_mcmc must be positive", 0, 1, 1e9);
  for (const GpDev* g : gps) {
    if (g == nullptr) throw Error(MOE_ERR_RUNTIME, "NULL GP handle in the MCMC ensemble");
    if (g->d != gps[0]->d || g->g != gps[0]->g)
      throw Error(MOE_ERR_INVALID_VALUE, "MCMC ensemble members must share dim and the observed-derivative list", g->d, gps[0]->d, 0);
  }
}

std::atomic<int> g_ens_launch{-1};  // -1: follow the environment
std::atomic<long long> g_ens_stats[4];
}  // namespace

bool ensemble_launches() {
  const int v = g_ens_launch.load();
  if (v >= 0) return v != 0;
  const char* e = std::getenv("MOE_ENS_LAUNCH");
  return !(e != nullptr && *e != 0 && std::atoi(e) == 0);
}
void set_ensemble_launches(int on) { g_ens_launch.store(on < 0 ? -1 : (on != 0 ? 1 : 0)); }
void ensemble_launch_stats(long long* out4) {
  for (int i = 0; i < 4; ++i) out4[i] = g_ens_stats[i].load();
}

namespace {
// ---- ensemble-wide launches (r6; launch.hpp) ----
// The recordings of the members' evaluation chains, zipped: position by position ONE launch of a kernel's ensemble twin over all
// members (their argument records side by side in a device table), one copy kernel for their small pinned copies, everything else
// member after member -- on ONE stream.  Returns false (nothing launched) when the recordings do not line up or too few positions
// merge to be worth the members' concurrency on their own streams.
struct EnsArena {
  PinnedBuf<unsigned char>& host;
  DevBuf<unsigned char>& dev;
};
constexpr size_t kEnsCopyMaxBytes = (size_t)4 << 20;

bool ens_copy_mergeable(const std::vector<Recorder>& recs, size_t pos) {
  static const bool on = !(std::getenv("MOE_ENS_COPY") != nullptr && std::atoi(std::getenv("MOE_ENS_COPY")) == 0);
  if (!on) return false;
  for (const Recorder& r : recs) {
    const LaunchOp& o = r.ops[pos];
    if (!o.host_pinned || o.copy.bytes % 8 != 0 || o.copy.bytes > kEnsCopyMaxBytes) return false;
  }
  return true;
}

bool replay_ensemble(const std::vector<Recorder>& recs, hipStream_t z, EnsArena& arena, int* merged_out) {
  const size_t M = recs.size(), L = recs[0].ops.size();
  std::vector<char> how(L, 0);  // 0: member after member, 1: ensemble twin, 2: copy kernel
  std::vector<size_t> off(L, 0);
  size_t total = 0, merged = 0;
  for (const Recorder& r : recs)
    if (r.ops.size() != L) return false;
  for (size_t k = 0; k < L; ++k) {
    const LaunchOp& a = recs[0].ops[k];
    bool same = true;
    for (size_t i = 1; i < M && same; ++i) {
      const LaunchOp& b = recs[i].ops[k];
      same = b.kind == a.kind && b.ens_launch == a.ens_launch && b.args.size() == a.args.size() && b.shm == a.shm &&
             b.grid.x == a.grid.x && b.grid.y == a.grid.y && b.grid.z == a.grid.z && b.block.x == a.block.x &&
             b.block.y == a.block.y && b.block.z == a.block.z;
    }
    if (!same) {
      if (recs[0].ops[k].kind != LaunchOp::kKernel) return false;  // (copies and the rest must line up; kernels may differ in shape)
      for (size_t i = 1; i < M; ++i)
        if (recs[i].ops[k].kind != LaunchOp::kKernel) return false;
      continue;
    }
    if (a.kind == LaunchOp::kKernel && a.ens_launch != nullptr && (size_t)a.grid.z * M <= 65535) {
      how[k] = 1;
      off[k] = total;
      total += (a.args.size() * M + 255) / 256 * 256;
      ++merged;
    } else if (a.kind == LaunchOp::kCopy && ens_copy_mergeable(recs, k)) {
      how[k] = 2;
      off[k] = total;
      total += (sizeof(CopyEntry) * M + 255) / 256 * 256;
      ++merged;
    }
  }
  if (merged_out) *merged_out = (int)merged;
  if (merged * 10 < L * 6) return false;
  arena.host.reserve(total);
  arena.dev.reserve(total);
  for (size_t k = 0; k < L; ++k) {
    if (how[k] == 1) {
      const size_t sz = recs[0].ops[k].args.size();
      for (size_t i = 0; i < M; ++i) std::memcpy(arena.host.p + off[k] + i * sz, recs[i].ops[k].args.data(), sz);
    } else if (how[k] == 2) {
      for (size_t i = 0; i < M; ++i) std::memcpy(arena.host.p + off[k] + i * sizeof(CopyEntry), &recs[i].ops[k].copy, sizeof(CopyEntry));
    }
  }
  MOE_HIP_CHECK(hipMemcpyAsync(arena.dev.p, arena.host.p, total, hipMemcpyHostToDevice, z));
  for (size_t k = 0; k < L; ++k) {
    const LaunchOp& a = recs[0].ops[k];
    if (how[k] == 1) {
      a.ens_launch(arena.dev.p + off[k], (int)M, a.grid, a.block, a.shm, z);
    } else if (how[k] == 2) {
      size_t most = 0;
      for (size_t i = 0; i < M; ++i) most = std::max(most, recs[i].ops[k].copy.bytes);
      const unsigned bx = (unsigned)std::max<size_t>(1, std::min<size_t>(64, (most / 8 + 1023) / 1024));
      ens_copy_kernel<0><<<dim3(bx, (unsigned)M), dim3(256), 0, z>>>((const CopyEntry*)(arena.dev.p + off[k]));
    } else {
      for (size_t i = 0; i < M; ++i) recs[i].ops[k].run(z);
    }
  }
  MOE_HIP_CHECK(hipGetLastError());
  return true;
}

void release_retired(std::vector<Recorder>& recs) {
  for (Recorder& r : recs) {
    for (auto& b : r.retired_dev) DevicePool::get().give(b.first, b.second);
    for (auto& b : r.retired_host) DevicePool::get().give_host(b.first, b.second);
    r.retired_dev.clear();
    r.retired_host.clear();
  }
}

struct RetireOnExit {  // (blocks the members' buffers outgrew while their launches were pending: back to the pool on every path)
  std::vector<Recorder>& r;
  ~RetireOnExit() { release_retired(r); }
};

}  // namespace

// The per-member values of a batch, not yet added up (r5): what the ranks of a member-sharded optimisation exchange, so that every
// rank can add the members up in GLOBAL order -- the order kg_mcmc_sums adds them in on one rank.
void kg_mcmc_members(const std::vector<GpDev*>& gps, int num_fidelity, const moe_gd_params_t& inner, const double* bounds,
                     const double* discrete_all, int P, const double* Xq_all, int num_evals, const double* Xp, int q, int p,
                     int num_mc, const double* best_so_far, const double* normals, bool want_grad, double* kg_mem, double* grad_mem,
                     const double* disc_head) {
  check_ensemble(gps);
  const int d = gps[0]->d, qd = q * d, E = num_evals;
  std::vector<double> ks(E), gs(want_grad ? (size_t)E * qd : 0);
  const size_t disc_stride = (size_t)P * (d - num_fidelity);
  // (every member owns its workspaces: the batch each of them may carry is the budget divided by the ensemble size)
  const char* bg = std::getenv("MOE_KG_BATCH_GB");
  const double budget = ((bg && *bg) ? std::atof(bg) : 48.0) / (double)gps.size();
  const int max_e = kg_max_batch(*gps[0], P, q, p, num_mc, want_grad, budget);
  for (int e0 = 0; e0 < E; e0 += max_e) {
    const int ne = std::min(max_e, E - e0);
    // Every member's evaluation is ~30 small launches on its own stream: at BO sizes (n = tens of points, M = 2^7 -- the regime of a
    // whole suggestion, bench.py --config suggest) issuing them costs the host more than running them costs the device, so the
    // members are issued by host threads side by side (r5: 2.0 -> ms per optimiser step of 16 members x 20 restarts; MOE_MCMC_THREADS=1:
    // one after another, as in round 4).  Each member owns its stream, workspaces and staging buffers; results are collected in order.
    std::vector<KgPending> pending(gps.size());
    const auto t_call = std::chrono::steady_clock::now();
    // r6: ensemble-wide launches -- the members' chains are RECORDED (same host threads), then replayed as one chain of launches over all
    // members on the first member's stream (replay_ensemble); recordings that do not line up are replayed per member on the members'
    // own streams, and a shape that did not merge twice is launched directly from then on.  MOE_ENS_LAUNCH=0: always direct.
    static thread_local std::map<long, int> ens_misses;
    const long ens_key = ((long)gps[0]->N * 4096 + (long)(q + p) * 64 + (want_grad ? 1 : 0)) * 64 + (long)gps.size();
    bool ens = ensemble_launches() && gps.size() > 1 && ens_misses[ens_key] < 2;
    for (size_t i = 1; i < gps.size() && ens; ++i) ens = gps[i]->device == gps[0]->device;
    std::vector<Recorder> recs(ens ? gps.size() : 0);
    struct Hint {
      explicit Hint(int m) { set_ensemble_members_hint(m); }
      ~Hint() { set_ensemble_members_hint(1); }
    } hint(ens ? (int)gps.size() : 1);
    auto issue = [&](size_t i) {
      Recorder::Scope scope(ens ? &recs[i] : Recorder::current());
      pending[i] = kg_launch(*gps[i], num_fidelity, inner, bounds, discrete_all + i * disc_stride, P, Xq_all + (size_t)e0 * qd, ne, Xp, q,
                             p, num_mc, best_so_far[i], normals, 0, num_mc, want_grad, false, budget, disc_head);
    };
    const char* mt = std::getenv("MOE_MCMC_THREADS");
    // (recording is cheap -- no launch is issued -- and starting 16 host threads per evaluation costs more than it: 0.156 s per
    //  suggestion with them, 0.113 s without; MOE_MCMC_THREADS still sets the count for member-by-member launches)
    const size_t nthreads = ens ? 1 : std::min(gps.size(), (size_t)std::max(1, (mt && *mt) ? std::atoi(mt) : 16));
    RetireOnExit retire{recs};
    if (nthreads <= 1) {
      for (size_t i = 0; i < gps.size(); ++i) issue(i);
    } else {
      std::vector<Error> errors(nthreads, Error(MOE_OK, ""));
      std::vector<char> failed(nthreads, 0);
      std::vector<std::thread> threads;
      threads.reserve(nthreads);
      struct JoinAll {
        std::vector<std::thread>& t;
        ~JoinAll() {
          for (std::thread& th : t)
            if (th.joinable()) th.join();
        }
      } join_all{threads};
      for (size_t k = 0; k < nthreads; ++k)
        threads.emplace_back([&, k] {
          try {
            for (size_t i = k; i < gps.size(); i += nthreads) issue(i);
          } catch (const Error& e) {
            errors[k] = e;
            failed[k] = 1;
          } catch (const std::exception& e) {
            errors[k] = Error(MOE_ERR_RUNTIME, e.what());
            failed[k] = 1;
          }
        });
      for (std::thread& t : threads) t.join();
      for (size_t k = 0; k < nthreads; ++k)
        if (failed[k]) {
          // (members already in flight finish on their own streams; their workspaces are not touched again before the next call's
          //  stream-ordered work.  With recorded launches nothing is in flight.)
          for (size_t i = 0; i < gps.size() && !ens; ++i)
            if (pending[i].collect) {
              try {
                pending[i].collect(ks.data(), want_grad ? gs.data() : nullptr, nullptr, nullptr);
              } catch (...) {
              }
            }
          throw errors[k];
        }
    }
    const auto t_rec = std::chrono::steady_clock::now();
    auto t_issue = t_rec, t_sync = t_rec;
    if (ens) {
      EnsArena arena{gps[0]->hEns, gps[0]->dEns};
      gps[0]->use_device();
      hipStream_t z = gps[0]->stream;
      int merged = 0;
      if (replay_ensemble(recs, z, arena, &merged)) {
        t_issue = std::chrono::steady_clock::now();
        MOE_HIP_CHECK(hipStreamSynchronize(z));
        t_sync = std::chrono::steady_clock::now();
        ens_misses[ens_key] = 0;
        g_ens_stats[0] += 1;
        g_ens_stats[2] += 1 + merged + ((long long)recs[0].ops.size() - merged) * (long long)gps.size();  // (1: the table's copy)
        g_ens_stats[3] += (long long)recs[0].ops.size() * (long long)gps.size();
      } else {
        ++ens_misses[ens_key];
        g_ens_stats[1] += 1;
        for (size_t i = 0; i < gps.size(); ++i) {
          gps[i]->use_device();
          for (const LaunchOp& op : recs[i].ops) op.run(gps[i]->stream);
        }
      }
      if (std::getenv("MOE_ENS_TRACE") != nullptr) {
        auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
          return std::chrono::duration<double, std::micro>(b - a).count();
        };
        std::fprintf(stderr, "[moe ens] members %zu, ops %zu, merged positions %d, misses %d; us: record %.0f, zip + issue %.0f, wait %.0f\n",
                     gps.size(), recs[0].ops.size(), merged, ens_misses[ens_key], us(t_call, t_rec), us(t_rec, t_issue), us(t_issue, t_sync));
      }
    }
    for (size_t i = 0; i < gps.size(); ++i) {
      pending[i].collect(ks.data(), want_grad ? gs.data() : nullptr, nullptr, nullptr);
      for (int e = 0; e < ne; ++e) kg_mem[i * (size_t)E + e0 + e] = ks[e] / (double)num_mc;
      if (want_grad)
        for (size_t j = 0; j < (size_t)ne * qd; ++j) grad_mem[(i * (size_t)E + e0) * qd + j] = gs[j] / (double)num_mc;
    }
  }
}

void kg_mcmc_sums(const std::vector<GpDev*>& gps, int num_fidelity, const moe_gd_params_t& inner, const double* bounds,
                  const double* discrete_all, int P, const double* Xq_all, int num_evals, const double* Xp, int q, int p,
                  int num_mc, const double* best_so_far, const double* normals, bool want_grad, double* kg_sum, double* grad_sum,
                  const double* disc_head) {
  check_ensemble(gps);
  const int d = gps[0]->d, qd = q * d, E = num_evals;
  // launch every member (own stream, own workspaces), then collect: member i's kernels run while member i+1's state set-up
  // and host algebra are being prepared (kg_mcmc_members); the members are added up in their order
  std::vector<double> km(gps.size() * (size_t)E), gm(want_grad ? gps.size() * (size_t)E * qd : 0);
  kg_mcmc_members(gps, num_fidelity, inner, bounds, discrete_all, P, Xq_all, E, Xp, q, p, num_mc, best_so_far, normals, want_grad,
                  km.data(), want_grad ? gm.data() : nullptr, disc_head);
  std::fill(kg_sum, kg_sum + E, 0.0);
  if (want_grad) std::fill(grad_sum, grad_sum + (size_t)E * qd, 0.0);
  for (size_t i = 0; i < gps.size(); ++i) {
    for (int e = 0; e < E; ++e) kg_sum[e] += km[i * (size_t)E + e];
    if (want_grad)
      for (size_t j = 0; j < (size_t)E * qd; ++j) grad_sum[j] += gm[i * (size_t)E * qd + j];
  }
}

void kg_mcmc_finalize(double* kg, double* grad, const double* Xq_all, int num_evals, int q, int d, int num_fidelity, int num_mcmc) {
  const int qd = q * d;
  for (int e = 0; e < num_evals; ++e) {
    const double* x = Xq_all + (size_t)e * qd;
    double cost = 1.0;
    int index = -1;
    if (num_fidelity > 0) {  // ComputeCost (.cpp:84-102): the largest product of the fidelity coordinates over the q points
      cost = 0.0;
      for (int i = 0; i < q; ++i) {
        double pc = 1.0;
        for (int j = d - num_fidelity; j < d; ++j) pc *= x[i * d + j];
        if (cost < pc) {
          cost = pc;
          index = i;
        }
      }
    }
    const double mean_kg = kg[e] / (double)num_mcmc;
    kg[e] = mean_kg / cost;
    if (grad != nullptr) {
      double* g = grad + (size_t)e * qd;
      for (int k = 0; k < qd; ++k) {
        double gc = 0.0;  // ComputeGradCost (.cpp:107-127)
        if (num_fidelity > 0 && index >= 0 && k / d == index && k % d >= d - num_fidelity) gc = cost / x[k];
        g[k] = (g[k] / (double)num_mcmc * cost - mean_kg * gc) / (cost * cost);
      }
    }
  }
}

void ei_mcmc_batch(const std::vector<GpDev*>& gps, const double* Xq_all, int num_evals, const double* Xp, int q, int p, int num_mc,
                   const double* best_so_far, const double* normals, bool analytic, double* ei, double* grad_ei) {
  check_ensemble(gps);
  const int d = gps[0]->d, qd = q * d, E = num_evals;
  if (analytic && !(q == 1 && p == 0)) throw Error(MOE_ERR_RUNTIME, "analytic EI needs num_to_sample == 1 and num_being_sampled == 0");
  if (ei) std::fill(ei, ei + E, 0.0);
  if (grad_ei) std::fill(grad_ei, grad_ei + (size_t)E * qd, 0.0);
  std::vector<double> es(ei ? E : 0), gs(grad_ei ? (size_t)E * qd : 0);
  // r6: ensemble-wide launches for the Monte-Carlo EI as for KG (kg_mcmc_members): the members' chains recorded, then every kernel issued
  // once for all of them; the same bits as the loop below.
  static thread_local std::map<long, int> ens_misses;
  const long ens_key = ((long)gps[0]->N * 4096 + (long)(q + p) * 64 + (grad_ei ? 1 : 0)) * 64 + (long)gps.size();
  bool ens = !analytic && ei_device_algebra() && q + p <= 16 && ensemble_launches() && gps.size() > 1 && ens_misses[ens_key] < 2;  // (wider unions wait on the host mid-way: ei.hip)
  for (size_t i = 1; i < gps.size() && ens; ++i) ens = gps[i]->device == gps[0]->device;
  if (ens) {
    std::vector<Recorder> recs(gps.size());
    std::vector<EiPending> pending(gps.size());
    RetireOnExit retire{recs};
    for (size_t i = 0; i < gps.size(); ++i) {
      Recorder::Scope scope(&recs[i]);
      pending[i] = ei_launch(*gps[i], Xq_all, E, Xp, q, p, num_mc, best_so_far[i], normals, ei != nullptr, grad_ei != nullptr);
    }
    EnsArena arena{gps[0]->hEns, gps[0]->dEns};
    gps[0]->use_device();
    hipStream_t z = gps[0]->stream;
    int merged = 0;
    if (replay_ensemble(recs, z, arena, &merged)) {
      MOE_HIP_CHECK(hipStreamSynchronize(z));
      ens_misses[ens_key] = 0;
      g_ens_stats[0] += 1;
      g_ens_stats[2] += 1 + merged + ((long long)recs[0].ops.size() - merged) * (long long)gps.size();
      g_ens_stats[3] += (long long)recs[0].ops.size() * (long long)gps.size();
    } else {
      ++ens_misses[ens_key];
      g_ens_stats[1] += 1;
      for (size_t i = 0; i < gps.size(); ++i) {
        gps[i]->use_device();
        for (const LaunchOp& op : recs[i].ops) op.run(gps[i]->stream);
      }
    }
    for (size_t i = 0; i < gps.size(); ++i) {
      pending[i].collect(ei ? es.data() : nullptr, grad_ei ? gs.data() : nullptr);
      if (ei)
        for (int e = 0; e < E; ++e) ei[e] += es[e];
      if (grad_ei)
        for (size_t j = 0; j < (size_t)E * qd; ++j) grad_ei[j] += gs[j];
    }
  }
  for (size_t i = 0; i < gps.size() && !ens; ++i) {
    if (analytic)
      ei_analytic_batch(*gps[i], Xq_all, E, best_so_far[i], ei ? es.data() : nullptr, grad_ei ? gs.data() : nullptr);
    else
      ei_evaluate_batch(*gps[i], Xq_all, E, Xp, q, p, num_mc, best_so_far[i], normals, ei ? es.data() : nullptr,
                        grad_ei ? gs.data() : nullptr);
    if (ei)
      for (int e = 0; e < E; ++e) ei[e] += es[e];
    if (grad_ei)
      for (size_t j = 0; j < (size_t)E * qd; ++j) grad_ei[j] += gs[j];
  }
  const double inv = 1.0 / (double)gps.size();
  if (ei)
    for (int e = 0; e < E; ++e) ei[e] *= inv;
  if (grad_ei)
    for (size_t j = 0; j < (size_t)E * qd; ++j) grad_ei[j] *= inv;
}

namespace {

// TensorProductDomain::LimitUpdate (gpp_domain.cpp:64-105) on one coordinate.
double limit_update_coord(double lo, double hi, double max_relative_change, double x, double desired) {
  double dist = std::fmin(x - lo, hi - x);
  if (std::fabs(desired) > max_relative_change * dist) desired = std::copysign(max_relative_change * dist, desired);
  const double next = x + desired;
  if (next < lo) {
    desired = (x + desired * 0.5 < lo) ? (lo - x) * 0.5 : desired * 0.5;
  } else if (next > hi) {
    desired = (x + desired * 0.5 > hi) ? (hi - x) * 0.5 : desired * 0.5;
  }
  return desired;
}

// ComputeCost / ComputeGradCost (.cpp:84-127) on the points the MCMC state holds; gcost[qd] (may be NULL).
double fidelity_cost(const double* x, int q, int d, int num_fidelity, double* gcost) {
  if (gcost) std::fill(gcost, gcost + (size_t)q * d, 0.0);
  if (num_fidelity == 0) return 1.0;
  double cost = 0.0;
  int index = -1;
  for (int i = 0; i < q; ++i) {
    double pc = 1.0;
    for (int j = d - num_fidelity; j < d; ++j) pc *= x[i * d + j];
    if (cost < pc) {
      cost = pc;
      index = i;
    }
  }
  if (gcost && index >= 0)
    for (int j = d - num_fidelity; j < d; ++j) gcost[index * d + j] = cost / x[index * d + j];
  return cost;
}

}  // namespace

// ComputeKGMCMCOptimalPointsToSampleViaMultistartGradientDescent / EvaluateKGMCMCAtPointList
// (gpp_knowledge_gradient_mcmc_optimization.hpp:665-862) AS THE REFERENCE EXECUTES THEM.  Its KnowledgeGradientMCMCState
//   * forwards all q points to the per-GP states in SetCurrentPoint but copies only the FIRST point into its own
//     union_of_points (.cpp:186-195: `points_to_sample_in + dim`), which is what GetCurrentPoint returns (.hpp:439-441) and what
//     the fidelity cost reads: the optimiser steps from, and reports, [moved first point ; the other points of the state's
//     construction point = starts[0]], while the objective is evaluated where the per-GP states really are;
//   * the evaluator's gradient accumulates into its output (.cpp:163-166), a vector GradientDescentOptimization allocates once
//     per restart (gpp_optimization.hpp:626): step i sees G_i = ((G_{i-1} + sum_i) / num_mcmc * cost - KG * gradcost) / cost^2;
//   * the per-GP states keep the discretised set of starts[0] (kg.hpp: disc_head).
// With q = 1 and no fidelity dimension only the last two are visible.  Pinned to the reference's end point
// (tests/golden/ref_kg_multistart.npz).  Every step evaluates all live restarts of all ensemble members in batched passes.
void kg_mcmc_multistart(const std::vector<GpDev*>& gps, int num_fidelity, const moe_gd_params_t& outer,
                        const moe_gd_params_t& inner, const double* bounds, const double* discrete_all, int P,
                        const double* starts, int num_starts, const double* Xp, int q, int p, int num_mc,
                        const double* best_so_far, const double* normals, int do_gradient_ascent, double* best_points,
                        double* best_kg, int* found, int total_num_mcmc, const Comm* comm) {
  check_ensemble(gps);
  if (num_starts <= 0) throw Error(MOE_ERR_BOUNDS, "num_multistarts must be > 1", num_starts, 1, 1e9);
  const bool shard = comm != nullptr && comm->world > 1;
  const int d = gps[0]->d, qd = q * d, nm = shard ? total_num_mcmc : (int)gps.size();
  multistart_trace_begin(comm == nullptr || comm->rank == 0);
  if (shard) {
    // member g lives on rank g % world at local index g / world
    const int W = comm->world, mine = (total_num_mcmc - comm->rank + W - 1) / W;
    if (total_num_mcmc < W || mine != (int)gps.size())
      throw Error(MOE_ERR_INVALID_VALUE, "member-sharded KG-MCMC optimisation: this rank must hold members rank, rank + world, ... "
                  "of total_num_mcmc >= world", (double)gps.size(), (double)mine, 0);
  }
  // Sums over ALL members of a batch of evaluations.  One rank: kg_mcmc_sums.  Sharded (r5): every rank evaluates its members,
  // ONE exchange of the per-member values, and every rank adds them up in global member order -- the bits of the single-rank sum.
  auto all_sums = [&](const double* x_all, int n, bool want_grad, double* ks, double* gs, const double* head_pts) {
    struct Timed {  // (the quirk-free branch goes through multistart(), which records its own rows: only the direct calls below count)
      bool on;
      int kind, items;
      std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
      ~Timed() {
        if (on) multistart_trace_add(kind, items, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
      }
    } timed{head_pts != nullptr, want_grad ? 1 : 0, n};
    if (!shard) {
      kg_mcmc_sums(gps, num_fidelity, inner, bounds, discrete_all, P, x_all, n, Xp, q, p, num_mc, best_so_far, normals, want_grad, ks, gs,
                   head_pts);
      return;
    }
    const int W = comm->world, per = (total_num_mcmc + W - 1) / W, width = n * (1 + (want_grad ? qd : 0));
    std::vector<double> all((size_t)W * per * width);
    // items = (rank, slot) pairs: item k * per + j is member j * W + k (or a pad); dealt so that rank k evaluates exactly its own
    sharded_items(Comm{comm->rank, W, comm->allgather}, W, per * width, [&](const std::vector<int>&, double* out_local) {
      std::vector<double> km(gps.size() * (size_t)n), gm(want_grad ? gps.size() * (size_t)n * qd : 0);
      kg_mcmc_members(gps, num_fidelity, inner, bounds, discrete_all, P, x_all, n, Xp, q, p, num_mc, best_so_far, normals, want_grad,
                      km.data(), want_grad ? gm.data() : nullptr, head_pts);
      std::fill(out_local, out_local + (size_t)per * width, 0.0);
      for (size_t j = 0; j < gps.size(); ++j) {
        double* slot = out_local + j * (size_t)width;
        std::copy(&km[j * (size_t)n], &km[(j + 1) * (size_t)n], slot);
        if (want_grad) std::copy(&gm[j * (size_t)n * qd], &gm[(j + 1) * (size_t)n * qd], slot + n);
      }
    }, all.data());
    std::fill(ks, ks + n, 0.0);
    if (want_grad) std::fill(gs, gs + (size_t)n * qd, 0.0);
    for (int g = 0; g < total_num_mcmc; ++g) {
      const double* slot = &all[((size_t)(g % W) * per + (size_t)(g / W)) * width];
      for (int e = 0; e < n; ++e) ks[e] += slot[e];
      if (want_grad)
        for (size_t j = 0; j < (size_t)n * qd; ++j) gs[j] += slot[n + j];
    }
  };
  if (!reference_quirks()) {
    // The driver as the reference INTENDS it (MOE_REFERENCE_QUIRKS=0 / moe_set_reference_quirks(0)): the generic multistart over
    // the MCMC-averaged objective KG(x) = mean_i KG_i(x) / cost(x) -- fresh per-GP states at every evaluation, all q points move
    // and are returned, the plain gradient ((mean grad) cost - KG grad cost) / cost^2 at every step.
    BatchObjective f;
    f.values = [&](const double* x_all, int n, double* values) {
      all_sums(x_all, n, false, values, nullptr, nullptr);
      kg_mcmc_finalize(values, nullptr, x_all, n, q, d, num_fidelity, nm);
    };
    f.grads = [&](const double* x_all, int n, double* grads) {
      std::vector<double> ks(n);
      all_sums(x_all, n, true, ks.data(), grads, nullptr);
      kg_mcmc_finalize(ks.data(), grads, x_all, n, q, d, num_fidelity, nm);
    };
    multistart(f, outer, bounds, d, qd, starts, num_starts, do_gradient_ascent, -INFINITY, best_points, best_kg, found);
    return;
  }
  const double* head = starts;
  auto seen_of = [&](const double* actual, double* seen) {  // what GetCurrentPoint returns for a state moved to `actual`
    std::copy(head, head + qd, seen);
    std::copy(actual, actual + d, seen);
  };
  auto sums = [&](const double* x_all, int n, bool want_grad, double* ks, double* gs) { all_sums(x_all, n, want_grad, ks, gs, head); };
  *found = 0;
  *best_kg = -INFINITY;
  std::vector<double> seen(qd);
  // value at every start: per-GP states at the start itself, cost from what the MCMC state holds
  std::vector<double> vals(num_starts);
  sums(starts, num_starts, false, vals.data(), nullptr);
  for (int s = 0; s < num_starts; ++s) {
    seen_of(starts + (size_t)s * qd, seen.data());
    vals[s] /= (double)nm * fidelity_cost(seen.data(), q, d, num_fidelity, nullptr);
  }
  std::vector<int> order;
  if (do_gradient_ascent) {
    order = top_k_order(vals.data(), num_starts);
  } else {
    order.resize(num_starts);
    for (int s = 0; s < num_starts; ++s) order[s] = s;
  }
  const int S = (int)order.size();
  std::vector<double> actual((size_t)S * qd), seen_all((size_t)S * qd);
  for (int s = 0; s < S; ++s) {
    std::copy(starts + (size_t)order[s] * qd, starts + (size_t)(order[s] + 1) * qd, &actual[(size_t)s * qd]);
    seen_of(&actual[(size_t)s * qd], &seen_all[(size_t)s * qd]);
  }
  std::copy(seen_all.begin(), seen_all.begin() + qd, best_points);  // the IO container's seed (.hpp:733 / point-list: first start)
  std::vector<double> end_vals(S);
  if (do_gradient_ascent && outer.max_num_restarts > 0) {
    const double step_tol = outer.tolerance / (double)outer.max_num_steps;
    const DomainLimiter limiter(outer, bounds, d);
    std::vector<char> alive(S, 1), running(S);
    std::vector<double> cur((size_t)S * qd), nxt((size_t)S * qd), G((size_t)S * qd), xs((size_t)S * qd), ks(S), gs((size_t)S * qd),
        gcost(qd), step(qd);
    std::vector<int> idx;
    for (int r = 0; r < outer.max_num_restarts; ++r) {
      if (std::none_of(alive.begin(), alive.end(), [](char c) { return c != 0; })) break;
      cur = seen_all;
      nxt = seen_all;
      std::fill(G.begin(), G.end(), 0.0);
      running = alive;
      for (int i = 0; i < outer.max_num_steps; ++i) {
        idx.clear();
        for (int s = 0; s < S; ++s)
          if (running[s]) idx.push_back(s);
        if (idx.empty()) break;
        const double alpha = outer.pre_mult * std::pow((double)(i + 1), -outer.gamma);
        for (size_t k = 0; k < idx.size(); ++k)
          std::copy(&actual[(size_t)idx[k] * qd], &actual[(size_t)(idx[k] + 1) * qd], &xs[k * qd]);
        sums(xs.data(), (int)idx.size(), true, ks.data(), gs.data());
        for (size_t k = 0; k < idx.size(); ++k) {
          const int s = idx[k];
          double* Gs = &G[(size_t)s * qd];
          double* ns = &nxt[(size_t)s * qd];
          const double cost = fidelity_cost(&seen_all[(size_t)s * qd], q, d, num_fidelity, gcost.data());
          const double mean_kg = ks[k] / (double)nm;
          for (int j = 0; j < qd; ++j) {
            Gs[j] = ((Gs[j] + gs[k * qd + j]) / (double)nm * cost - mean_kg * gcost[j]) / (cost * cost);
            step[j] = alpha * Gs[j];
          }
          limiter.apply(outer.max_relative_change, ns, step.data(), qd);
          for (int j = 0; j < qd; ++j) ns[j] += step[j];
          std::copy(ns, ns + qd, &actual[(size_t)s * qd]);  // SetCurrentPoint: the per-GP states follow all q points ...
          seen_of(ns, &seen_all[(size_t)s * qd]);            // ... the MCMC state only the first
          double n2 = 0.0;
          for (int j = 0; j < qd; ++j) n2 += step[j] * step[j];
          if (std::sqrt(n2) < step_tol) running[s] = 0;
        }
      }
      for (int s = 0; s < S; ++s) {
        if (!alive[s]) continue;
        double n2 = 0.0;
        for (int j = 0; j < qd; ++j) {
          const double dlt = cur[(size_t)s * qd + j] - seen_all[(size_t)s * qd + j];
          n2 += dlt * dlt;
        }
        if (!(std::sqrt(n2) > outer.tolerance)) alive[s] = 0;
      }
    }
  }
  sums(actual.data(), S, false, end_vals.data(), nullptr);
  for (int s = 0; s < S; ++s) {
    const double v = end_vals[s] / ((double)nm * fidelity_cost(&seen_all[(size_t)s * qd], q, d, num_fidelity, nullptr));
    if (v > *best_kg) {  // strict, like MultistartOptimizer's compare (gpp_optimization.hpp:1512)
      *best_kg = v;
      std::copy(&seen_all[(size_t)s * qd], &seen_all[(size_t)(s + 1) * qd], best_points);
      *found = 1;
    }
  }
}

void ei_mcmc_multistart(const std::vector<GpDev*>& gps, const moe_gd_params_t& outer, const double* bounds, const double* starts,
                        int num_starts, const double* Xp, int q, int p, int num_mc, const double* best_so_far,
                        const double* normals, int do_gradient_ascent, double* best_points, double* best_ei, int* found) {
  check_ensemble(gps);
  const int d = gps[0]->d, qd = q * d;
  const bool analytic = (q == 1 && p == 0);  // gpp_expected_improvement_mcmc_optimization.hpp:871
  if (!analytic && normals == nullptr) throw Error(MOE_ERR_RUNTIME, "q,p-EI by Monte Carlo needs a normal table");
  BatchObjective f;
  f.values = [&](const double* x_all, int n, double* values) {
    ei_mcmc_batch(gps, x_all, n, Xp, q, p, num_mc, best_so_far, normals, analytic, values, nullptr);
  };
  f.grads = [&](const double* x_all, int n, double* grads) {
    ei_mcmc_batch(gps, x_all, n, Xp, q, p, num_mc, best_so_far, normals, analytic, nullptr, grads);
  };
  // the MCMC drivers seed their IO container with 0.0, not the -1.0 of the single-GP ones
  // (gpp_expected_improvement_mcmc_optimization.hpp:914, 968; .cpp:268, 295): an all-zero EI surface reports found = 0
  multistart(f, outer, bounds, d, qd, starts, num_starts, do_gradient_ascent, 0.0, best_points, best_ei, found);
}

}  // namespace moe
