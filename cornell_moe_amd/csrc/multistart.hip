// cornell_moe_amd/csrc/multistart.hip -- the callers of the hot path, host C++ above the evaluators:
//   * ComputeKGOptimalPointsToSampleViaMultistartGradientDescent (gpp_knowledge_gradient_optimization.hpp:860-935):
//     KG value at every start, the best 20 kept (:895-921), restarted gradient ASCENT on each
//     (GradientDescentOptimizer, gpp_optimization.hpp:619-705, 1144-1185), KG value at every end point, best one returned
//     (MultistartOptimizer, gpp_optimization.hpp:1472-1546);
//   * ComputeKGOptimalPointsToSampleViaLatinHypercubeSearch / EvaluateKGAtPointList (:1090-1141): value search;
//   * ComputeOptimalPosteriorMean from one initial guess (gpp_knowledge_gradient_optimization.cpp:420-472);
//   * ComputeLatinHypercubePointsInDomain (gpp_random.cpp:173-194) for the start sets.
// Where the reference runs one OpenMP thread per restart, every step here evaluates ALL live restarts in one batched
// device pass (kg_evaluate_batch).  Every evaluation replays the same normal table (the reference rewinds its RNG before
// each evaluation), so the ascent sees common random numbers.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <chrono>
#include <cstdlib>
#include <functional>
#include <memory>
#include <mutex>
#include <numeric>
#include <queue>
#include <random>
#include <utility>

#include "gp.hpp"
#include "kg.hpp"

namespace moe {

namespace {
std::atomic<int> g_reference_quirks{-1};  // -1: follow the environment
}

bool reference_quirks() {
  const int v = g_reference_quirks.load();
  if (v >= 0) return v != 0;
  const char* e = std::getenv("MOE_REFERENCE_QUIRKS");
  return !(e && *e && std::atoi(e) == 0);
}

void set_reference_quirks(int on) { g_reference_quirks.store(on < 0 ? -1 : (on != 0 ? 1 : 0)); }

namespace {
std::mutex g_trace_mu;
std::vector<double> g_trace;
thread_local bool tl_trace = false;
}  // namespace

void multistart_trace_begin(bool enabled) {
  tl_trace = enabled;
  if (enabled) {
    std::lock_guard<std::mutex> lk(g_trace_mu);
    g_trace.clear();
  }
}
void multistart_trace_add(int kind, int items, double ms) {
  if (!tl_trace) return;
  std::lock_guard<std::mutex> lk(g_trace_mu);
  g_trace.push_back((double)kind);
  g_trace.push_back((double)items);
  g_trace.push_back(ms);
}
int multistart_trace_get(double* out, int cap) {
  std::lock_guard<std::mutex> lk(g_trace_mu);
  const int rows = (int)(g_trace.size() / 3);
  for (int i = 0; out != nullptr && i < std::min(rows, cap); ++i)
    for (int j = 0; j < 3; ++j) out[3 * i + j] = g_trace[3 * (size_t)i + j];
  return rows;
}

namespace {

struct TraceTimer {
  int kind, items;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  TraceTimer(int k, int n) : kind(k), items(n) {}
  ~TraceTimer() { multistart_trace_add(kind, items, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); }
};

constexpr int kTopK = 20;  // gpp_knowledge_gradient_optimization.hpp:901

// TensorProductDomain::LimitUpdate (gpp_domain.cpp:64-105) on one coordinate.
double limit_update_1d(double lo, double hi, double max_relative_change, double x, double desired) {
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

double norm2(const double* v, int n) {
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += v[i] * v[i];
  return std::sqrt(s);
}

// VectorNorm (gpp_linear_algebra.cpp:53-72): the scaled, overflow-safe recurrence -- the simplex update divides by it, so it is
// restated operation for operation.
double vector_norm_scaled(const double* v, int n) {
  if (n == 1) return std::fabs(v[0]);
  double scale = 0.0, scaled = 1.0;
  for (int i = 0; i < n; ++i) {
    if (v[i] != 0.0) {
      const double a = std::fabs(v[i]);
      if (scale < a) {
        const double t = scale / a;
        scaled = 1.0 + scaled * (t * t);
        scale = a;
      } else {
        const double t = a / scale;
        scaled += t * t;
      }
    }
  }
  return scale * std::sqrt(scaled);
}

// CheckPointInUnitSimplex (gpp_geometry.hpp:313-325)
bool in_unit_simplex(const double* pt, int d) {
  double sum = 0.0;
  for (int i = 0; i < d; ++i) {
    if (pt[i] < 0.0) return false;
    sum += pt[i];
  }
  return (sum - 4.0 * 2.220446049250313e-16) <= 1.0;
}

// SimplexIntersectTensorProductDomain (gpp_domain.hpp:215-349, gpp_domain.cpp:107-141): the box clipped to the unit hypercube, the
// emptiness test of its constructor, and its LimitUpdate (:234-290) on ONE point -- the tensor-product limit first, then, if the
// proposed point leaves the simplex, half the distance to the diagonal face along the (limited) direction.
struct SimplexDomain {
  int d;
  std::vector<double> box;  // clipped bounds
  double inv_sqrt_d;
  SimplexDomain(const double* bounds, int dim) : d(dim), box(2 * (size_t)dim), inv_sqrt_d(1.0 / std::sqrt((double)dim)) {
    double corner_sum = 0.0;
    bool empty = false;
    for (int i = 0; i < d; ++i) {
      box[2 * i] = std::fmax(bounds[2 * i], 0.0);
      box[2 * i + 1] = std::fmin(bounds[2 * i + 1], 1.0);
      empty = empty || box[2 * i] > box[2 * i + 1];
      corner_sum += box[2 * i];
    }
    if (corner_sum >= 1.0 || empty)
      throw Error(MOE_ERR_BOUNDS,
                  "Simplex/Tensor product intersection is EMPTY; 'lower left' corner coordinate sum out of bounds or bounding "
                  "boxes do not intersect.",
                  corner_sum, 0.0, 1.0);
  }
  void limit_update(double max_relative_change, const double* x, double* step) const {
    if (max_relative_change == 1.0) max_relative_change -= 4.0 * 2.220446049250313e-16;  // kRelativeChangeEpsilonTweak
    for (int j = 0; j < d; ++j) step[j] = limit_update_1d(box[2 * j], box[2 * j + 1], max_relative_change, x[j], step[j]);
    double norm = vector_norm_scaled(step, d);
    if (norm == 0.0) norm = 2.2250738585072014e-308;
    std::vector<double> dir(d), next(d);
    for (int j = 0; j < d; ++j) {
      dir[j] = step[j] / norm;
      next[j] = x[j] + step[j];
    }
    if (!in_unit_simplex(next.data(), d)) {
      // Plane::DistanceToPlaneAlongVector (gpp_geometry.hpp:252-272) for the plane -1/sqrt(d) + sum x_i / sqrt(d) = 0
      double xn = 0.0, vn = 0.0;
      for (int j = 0; j < d; ++j) {
        xn += x[j] * inv_sqrt_d;
        vn += dir[j] * inv_sqrt_d;
      }
      const double numerator = inv_sqrt_d - xn;  // -offset - x . n, offset = -1/sqrt(d)
      double dist = (vn == 0.0) ? (numerator == 0.0 ? 0.0 : INFINITY) : numerator / vn;
      if (dist < 0.0) dist = 0.0;
      const double relaxed = 0.5 * dist;  // kInvalidStepScaleFactor
      for (int j = 0; j < d; ++j) step[j] = relaxed * dir[j];
    }
  }
};

}  // namespace

struct DomainLimiter::Impl {
  std::unique_ptr<SimplexDomain> simplex;
};

DomainLimiter::DomainLimiter(const moe_gd_params_t& outer, const double* bounds_in, int dim) : d(dim), bounds(bounds_in), impl(new Impl) {
  if (outer.domain_type == MOE_DOMAIN_SIMPLEX)
    impl->simplex.reset(new SimplexDomain(bounds, d));
  else if (outer.domain_type != MOE_DOMAIN_TENSOR_PRODUCT)
    throw Error(MOE_ERR_INVALID_VALUE, "unknown domain_type (0 = tensor product, 1 = simplex)", outer.domain_type, 0, 1);
}
DomainLimiter::~DomainLimiter() = default;

void DomainLimiter::apply(double max_relative_change, const double* x, double* step, int qd) const {
  if (impl->simplex) {
    for (int pt = 0; pt < qd / d; ++pt) impl->simplex->limit_update(max_relative_change, x + (size_t)pt * d, step + (size_t)pt * d);
  } else {
    for (int j = 0; j < qd; ++j) {
      const int dd = j % d;
      step[j] = limit_update_1d(bounds[2 * dd], bounds[2 * dd + 1], max_relative_change, x[j], step[j]);
    }
  }
}

void latin_hypercube(unsigned int seed, const double* bounds, int dim, int num_points, double* out) {
  std::mt19937 eng(seed);
  std::vector<int> index(num_points);
  for (int i = 0; i < dim; ++i) {
    const double lo = bounds[2 * i], edge = (bounds[2 * i + 1] - bounds[2 * i]) / (double)num_points;
    std::iota(index.begin(), index.end(), 0);
    std::shuffle(index.begin(), index.end(), eng);
    std::uniform_real_distribution<double> uni(0.0, edge);
    for (int j = 0; j < num_points; ++j) out[(size_t)j * dim + i] = lo + edge * index[j] + uni(eng);
  }
}

void kg_values(GpDev& gp, int num_fidelity, const moe_gd_params_t& inner, const double* inner_bounds, const double* discrete,
               int P, const double* Xq_all, int num_evals, const double* Xp, int q, int p, int num_mc, double best_so_far,
               const double* normals, double* values, const double* disc_head) {
  if (num_evals <= 0) return;
  std::vector<double> sums(num_evals);
  kg_evaluate_batch(gp, num_fidelity, inner, inner_bounds, discrete, P, Xq_all, num_evals, Xp, q, p, num_mc, best_so_far,
                    normals, 0, num_mc, false, sums.data(), nullptr, nullptr, nullptr, disc_head);
  for (int e = 0; e < num_evals; ++e) values[e] = sums[e] / (double)num_mc;
}

namespace {

// GradientDescentOptimizer::Optimize (gpp_optimization.hpp:619-705, 1144-1185) for every start at once; x [S][qd] in place.
// bounds are per coordinate of ONE point (RepeatedDomain applies them to each of the q points, gpp_domain.hpp:509-520).
void gradient_ascent(const BatchObjective& f, const moe_gd_params_t& outer, const double* bounds, int d, int qd, double* x,
                     int S) {
  if (outer.max_num_restarts <= 0 || S <= 0) return;
  const double step_tol = outer.tolerance / (double)outer.max_num_steps;
  // (r4) RepeatedDomain<DomainType>::LimitUpdate: the tensor-product or simplex update, point by point (gpp_domain.hpp:509-520)
  const DomainLimiter limiter(outer, bounds, d);
  std::vector<char> alive(S, 1), running(S);
  std::vector<double> x_begin((size_t)S * qd), xs((size_t)S * qd), grad((size_t)S * qd), step(qd);
  std::vector<int> idx;
  for (int r = 0; r < outer.max_num_restarts; ++r) {
    if (std::none_of(alive.begin(), alive.end(), [](char c) { return c != 0; })) break;
    std::copy(x, x + (size_t)S * qd, x_begin.begin());
    running = alive;
    for (int i = 0; i < outer.max_num_steps; ++i) {
      idx.clear();
      for (int s = 0; s < S; ++s)
        if (running[s]) idx.push_back(s);
      if (idx.empty()) break;
      const double alpha = outer.pre_mult * std::pow((double)(i + 1), -outer.gamma);
      for (size_t k = 0; k < idx.size(); ++k) std::copy(x + (size_t)idx[k] * qd, x + (size_t)(idx[k] + 1) * qd, &xs[k * qd]);
      {
        TraceTimer tt(1, (int)idx.size());
        f.grads(xs.data(), (int)idx.size(), grad.data());
      }
      for (size_t k = 0; k < idx.size(); ++k) {
        double* xk = x + (size_t)idx[k] * qd;
        for (int j = 0; j < qd; ++j) step[j] = alpha * grad[k * qd + j];
        limiter.apply(outer.max_relative_change, xk, step.data(), qd);
        for (int j = 0; j < qd; ++j) xk[j] += step[j];
        if (norm2(step.data(), qd) < step_tol) running[idx[k]] = 0;
      }
    }
    for (int s = 0; s < S; ++s) {
      if (!alive[s]) continue;
      for (int j = 0; j < qd; ++j) step[j] = x_begin[(size_t)s * qd + j] - x[(size_t)s * qd + j];
      if (!(norm2(step.data(), qd) > outer.tolerance)) alive[s] = 0;
    }
  }
}

}  // namespace

void gradient_ascent_batch(const BatchObjective& f, const moe_gd_params_t& outer, const double* bounds, int d, int qd, double* x, int S) {
  gradient_ascent(f, outer, bounds, d, qd, x, S);
}

std::vector<int> top_k_order(const double* vals, int num_starts) {
  // std::priority_queue<std::pair<double, int>> is a max-heap on (-value, index): its top is the lowest-valued kept start and,
  // among equal values, the one with the larger index
  std::priority_queue<std::pair<double, int>> pq;
  for (int i = 0; i < num_starts; ++i) {
    if (i < kTopK) {
      pq.push(std::pair<double, int>(-vals[i], i));
    } else if (pq.top().first > -vals[i]) {
      pq.pop();
      pq.push(std::pair<double, int>(-vals[i], i));
    }
  }
  std::vector<int> order;
  while (!pq.empty()) {
    order.push_back(pq.top().second);
    pq.pop();
  }
  return order;
}

// Value at every start, the best 20 kept, restarted ascent on each, value at every end point, best one returned if it
// beats `floor_value` (MultistartOptimizer, gpp_optimization.hpp:1472-1546; the reference seeds its IO container with
// -inf for KG, gpp_knowledge_gradient_optimization.hpp:924, and -1.0 for EI, gpp_math.hpp:1728).
void multistart(const BatchObjective& f, const moe_gd_params_t& outer, const double* bounds, int d, int qd, const double* starts,
                int num_starts, int do_gradient_ascent, double floor_value, double* best_points, double* best_value,
                int* found) {
  if (num_starts <= 0) throw Error(MOE_ERR_BOUNDS, "num_multistarts must be > 1", num_starts, 1, 1e9);
  *found = 0;
  *best_value = floor_value;
  std::vector<double> vals(num_starts);
  {
    TraceTimer tt(0, num_starts);
    f.values(starts, num_starts, vals.data());
  }
  std::vector<double> ends, end_vals;
  int S = num_starts;
  if (do_gradient_ascent) {
    // the kept starts in the reference's own order (lowest kept value first): with MultistartOptimizer's strict compare the
    // FIRST of equal end values in this order wins
    const std::vector<int> order = top_k_order(vals.data(), num_starts);
    S = (int)order.size();
    ends.resize((size_t)S * qd);
    for (int s = 0; s < S; ++s) std::copy(starts + (size_t)order[s] * qd, starts + (size_t)(order[s] + 1) * qd, &ends[(size_t)s * qd]);
    // what the reference returns when nothing beats floor_value: the point its IO container was seeded with, i.e. the first
    // entry popped from its top-20 queue = the lowest-valued kept start (gpp_math.hpp:1717-1728)
    std::copy(&ends[0], &ends[(size_t)qd], best_points);
    gradient_ascent(f, outer, bounds, d, qd, ends.data(), S);
    end_vals.resize(S);
    TraceTimer tt(0, S);
    f.values(ends.data(), S, end_vals.data());
  } else {
    ends.assign(starts, starts + (size_t)num_starts * qd);
    end_vals = vals;
    std::copy(starts, starts + qd, best_points);  // seeded with the first point of the list (gpp_math.cpp:2326)
  }
  for (int s = 0; s < S; ++s) {
    if (end_vals[s] > *best_value) {  // strict, like MultistartOptimizer's compare (gpp_optimization.hpp:1512)
      *best_value = end_vals[s];
      std::copy(&ends[(size_t)s * qd], &ends[(size_t)(s + 1) * qd], best_points);
      *found = 1;
    }
  }
}

void sharded_items(const Comm& comm, int n, int width,
                   const std::function<void(const std::vector<int>& idx, double* out_local)>& eval_local, double* out) {
  const int W = std::max(comm.world, 1), r = comm.rank;
  std::vector<int> idx;
  for (int i = r; i < n; i += W) idx.push_back(i);
  const int per = (n + W - 1) / W;  // items per rank, padded: ONE fixed-size exchange
  // payload of a rank: [status | value | lo | hi | per x width results]; a failed rank's status travels with zeros
  const int count = 4 + per * width;
  std::vector<double> send((size_t)count, 0.0), recv((size_t)count * W, 0.0);
  Error mine(MOE_OK, "");
  try {
    if (!idx.empty()) eval_local(idx, send.data() + 4);
  } catch (const Error& e) {
    mine = e;
  } catch (const std::exception& e) {  // (bad_alloc, ...: this rank still joins the exchange -- no rank is left waiting in it)
    mine = Error(MOE_ERR_RUNTIME, std::string("exception in a rank's share of a multi-rank evaluation: ") + e.what());
  } catch (...) {
    mine = Error(MOE_ERR_RUNTIME, "unknown exception in a rank's share of a multi-rank evaluation");
  }
  if (mine.code != MOE_OK) {
    send[0] = (double)mine.code;
    send[1] = mine.payload[0];
    send[2] = mine.payload[1];
    send[3] = mine.payload[2];
    std::fill(send.begin() + 4, send.end(), 0.0);
  }
  if (W > 1) {
    if (!comm.allgather) throw Error(MOE_ERR_RUNTIME, "multi-rank optimisation without an exchange function");
    comm.allgather(send.data(), recv.data(), count);
  } else {
    recv = send;
  }
  for (int k = 0; k < W; ++k) {
    const double* rk = &recv[(size_t)k * count];
    if (rk[0] != 0.0) {
      if (k == r) throw mine;  // (this rank's own message)
      throw Error((int)rk[0], "an evaluation failed on another rank of the multi-rank optimisation (its message is on that rank)",
                  rk[1], rk[2], rk[3]);
    }
  }
  for (int i = 0; i < n; ++i) {
    const double* src = &recv[(size_t)(i % W) * count + 4 + (size_t)(i / W) * width];
    std::copy(src, src + width, out + (size_t)i * width);
  }
}

void kg_multistart(GpDev& gp, int num_fidelity, const moe_gd_params_t& outer, const moe_gd_params_t& inner, const double* bounds,
                   const double* discrete, int P, const double* starts, int num_starts, const double* Xp, int q, int p,
                   int num_mc, double best_so_far, const double* normals, int do_gradient_ascent, double* best_points,
                   double* best_kg, int* found, const Comm* comm) {
  const int d = gp.d, qd = q * d;
  // (the reference builds outer AND inner domain of one type, gpp_python_knowledge_gradient.cpp:288-296; here each parameter struct
  //  carries its own: the MC kernels' line search takes the inner one -- kg.hip kg_launch)
  // The reference builds its evaluation states at the FIRST start and moves them with SetCurrentPoint, which leaves the
  // discretised set behind (kg.hpp: disc_head): every evaluation of the run scores / starts its inner optimisation from the
  // first start's q points.  Reproduced: the end point is pinned to the reference's (tests/golden/ref_kg_multistart.npz).
  // With the quirks switched off every evaluation runs on a fresh state (its own q points in the discretised set), as the
  // single-evaluation entry points do.
  const double* head = (num_starts > 0 && reference_quirks()) ? starts : nullptr;
  multistart_trace_begin(comm == nullptr || comm->rank == 0);
  BatchObjective f;
  f.values = [&](const double* x_all, int n, double* values) {
    kg_values(gp, num_fidelity, inner, bounds, discrete, P, x_all, n, Xp, q, p, num_mc, best_so_far, normals, values, head);
  };
  f.grads = [&](const double* x_all, int n, double* grads) {
    std::vector<double> ksum(n);
    kg_evaluate_batch(gp, num_fidelity, inner, bounds, discrete, P, x_all, n, Xp, q, p, num_mc, best_so_far, normals, 0, num_mc,
                      true, ksum.data(), grads, nullptr, nullptr, head);
    for (size_t j = 0; j < (size_t)n * qd; ++j) grads[j] /= (double)num_mc;
  };
  if (comm != nullptr && comm->world > 1) {
    // r5: the restarts dealt to the ranks -- each rank's share in one batched device pass, one exchange per evaluation of the
    // optimiser (the reference's omp-parallel restarts with their critical-section merge, gpp_optimization.hpp:1472-1546)
    const BatchObjective local = f;
    auto gather = [&](const double* x_all, const std::vector<int>& idx, std::vector<double>& xl) {
      xl.resize(idx.size() * (size_t)qd);
      for (size_t k = 0; k < idx.size(); ++k) std::copy(x_all + (size_t)idx[k] * qd, x_all + (size_t)(idx[k] + 1) * qd, &xl[k * qd]);
    };
    f.values = [&, local](const double* x_all, int n, double* values) {
      sharded_items(*comm, n, 1, [&](const std::vector<int>& idx, double* out_local) {
        std::vector<double> xl;
        gather(x_all, idx, xl);
        local.values(xl.data(), (int)idx.size(), out_local);
      }, values);
    };
    f.grads = [&, local](const double* x_all, int n, double* grads) {
      sharded_items(*comm, n, qd, [&](const std::vector<int>& idx, double* out_local) {
        std::vector<double> xl;
        gather(x_all, idx, xl);
        local.grads(xl.data(), (int)idx.size(), out_local);
      }, grads);
    };
  }
  multistart(f, outer, bounds, d, qd, starts, num_starts, do_gradient_ascent, -INFINITY, best_points, best_kg, found);
}

void ei_multistart(GpDev& gp, const moe_gd_params_t& outer, const double* bounds, const double* starts, int num_starts,
                   const double* Xp, int q, int p, int num_mc, double best_so_far, const double* normals,
                   int do_gradient_ascent, double* best_points, double* best_ei, int* found) {
  const int d = gp.d, qd = q * d;
  BatchObjective f;
  if (q == 1 && p == 0) {  // special analytic case (gpp_math.hpp:1703, gpp_math.cpp:2317)
    f.values = [&](const double* x_all, int n, double* values) { ei_analytic_batch(gp, x_all, n, best_so_far, values, nullptr); };
    f.grads = [&](const double* x_all, int n, double* grads) { ei_analytic_batch(gp, x_all, n, best_so_far, nullptr, grads); };
  } else {
    if (normals == nullptr) throw Error(MOE_ERR_RUNTIME, "q,p-EI by Monte Carlo needs a normal table");
    f.values = [&](const double* x_all, int n, double* values) {
      ei_evaluate_batch(gp, x_all, n, Xp, q, p, num_mc, best_so_far, normals, values, nullptr);
    };
    f.grads = [&](const double* x_all, int n, double* grads) {
      ei_evaluate_batch(gp, x_all, n, Xp, q, p, num_mc, best_so_far, normals, nullptr, grads);
    };
  }
  multistart(f, outer, bounds, d, qd, starts, num_starts, do_gradient_ascent, -1.0, best_points, best_ei, found);
}

// ComputeOptimalPosteriorMean from ONE start (gpp_knowledge_gradient_optimization.cpp:420-472): back-tracking line-search
// ascent on f = -mu (fidelity coordinates pinned to 1), gpp_optimization.hpp:708-828 / 1242-1283.
void posterior_mean_optimize(GpDev& gp, int num_fidelity, const moe_gd_params_t& gd, const double* bounds, const double* x0,
                             double* best_point, double* best_value) {
  const int d = gp.d, size = d - num_fidelity;
  if (num_fidelity < 0 || num_fidelity >= d) throw Error(MOE_ERR_BOUNDS, "num_fidelity out of range", num_fidelity, 0, d - 1);
  DerivList none;
  none.g = 0;
  for (int i = 0; i < kMaxDerivs; ++i) none.idx[i] = 0;
  std::vector<double> x(x0, x0 + size), pt(d, 1.0), g(size), trial(size), step(size), x_begin(size), gm(d);
  auto f = [&](const double* p, double* grad) {
    std::copy(p, p + size, pt.begin());
    double mu;
    gp.mean_of_points(pt.data(), 1, &mu, grad ? gm.data() : nullptr);
    if (grad)
      for (int k = 0; k < size; ++k) grad[k] = -gm[k];
    return -mu;
  };
  double fcur = f(x.data(), nullptr);
  const DomainLimiter limiter(gd, bounds, size);
  const double step_tol = gd.tolerance / (double)std::max(gd.max_num_steps, 1);
  for (int r = 0; r < gd.max_num_restarts; ++r) {
    x_begin = x;
    for (int i = 0; i < gd.max_num_steps; ++i) {
      const double f0 = f(x.data(), g.data());
      fcur = f0;
      double alpha = gd.pre_mult * std::pow((double)(i + 1), -gd.gamma);
      const double n2 = std::inner_product(g.begin(), g.end(), g.begin(), 0.0);
      int search = 0;
      double ftrial = f0;
      for (; search < 30; ++search) {
        for (int k = 0; k < size; ++k) trial[k] = x[k] + alpha * g[k];
        ftrial = f(trial.data(), nullptr);
        if (ftrial - f0 > 0.5 * alpha * n2) break;
        alpha *= 0.5;
      }
      bool changed = false, nonzero = false;
      for (int k = 0; k < size; ++k) step[k] = alpha * g[k];
      limiter.apply(gd.max_relative_change, x.data(), step.data(), size);  // (tensor product or simplex: gd.domain_type)
      for (int k = 0; k < size; ++k) {
        changed = changed || step[k] != alpha * g[k];
        nonzero = nonzero || step[k] != 0.0;
      }
      if (search == 30 || !nonzero) break;
      double obj2 = ftrial;
      if (changed) {
        for (int k = 0; k < size; ++k) trial[k] = x[k] + step[k];
        obj2 = f(trial.data(), nullptr);
      }
      if (obj2 <= f0) break;
      for (int k = 0; k < size; ++k) x[k] += step[k];
      fcur = obj2;
      if (norm2(step.data(), size) < step_tol) break;
    }
    for (int k = 0; k < size; ++k) step[k] = x_begin[k] - x[k];
    if (!(norm2(step.data(), size) > gd.tolerance)) break;
  }
  std::copy(x.begin(), x.end(), best_point);
  if (best_value) *best_value = fcur;
}

}  // namespace moe
