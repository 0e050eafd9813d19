"""Outer multistart optimisation of q-KG: thin drivers over the C ABI (moe_kg_multistart,
moe_posterior_mean_optimize, moe_latin_hypercube -- csrc/multistart.hip), where every restart's KG value / gradient is
evaluated in ONE batched device pass per step instead of one OpenMP thread per restart.  The numpy functions below
(latin_hypercube, limit_update, kg_gradient_ascent) restate the same logic on top of moe_kg_batch; tests use them as the
independent check of the C++ drivers.

Follows ComputeKGOptimalPointsToSample (gpp_knowledge_gradient_optimization.cpp:490-551):
  * Latin-hypercube starts in the repeated domain (gpp_random.cpp:173-194, gpp_domain.hpp:490-504);
  * KG value at every start, the best 20 kept (gpp_knowledge_gradient_optimization.hpp:895-921; the reference pops an
    under-filled queue when there are fewer than 20 starts -- here all starts are kept in that case);
  * restarted gradient ASCENT on each kept start: x += LimitUpdate(pre_mult (i+1)^-gamma grad KG), no function evaluations,
    stop when |step| < tolerance / max_num_steps (gpp_optimization.hpp:619-705, 1144-1185);
  * KG value at every end point, the best one returned if it beats -inf (MultistartOptimizer, gpp_optimization.hpp:1472-1546);
  * Latin-hypercube value search as the fall-back / null-optimiser path (.cpp:514-541).
Every evaluation replays the same normal table (the reference rewinds its RNG before each evaluation), so the descent
sees common random numbers.  Also: posterior_mean_optimization (ComputeOptimalPosteriorMean from one initial guess,
gpp_knowledge_gradient_optimization.cpp:420-472 / gpp_python_knowledge_gradient.cpp:315-342).
"""
import numpy as np

TOP_K = 20  # gpp_knowledge_gradient_optimization.hpp:901


def latin_hypercube(bounds, num_samples, uniform):
    """ComputeLatinHypercubePointsInDomain: bounds [dim][2]; uniform(size) -> U[0,1) draws; returns [num_samples][dim]."""
    bounds = np.asarray(bounds, dtype=np.float64).reshape(-1, 2)
    dim = bounds.shape[0]
    pts = np.empty((num_samples, dim))
    for i in range(dim):
        edge = (bounds[i, 1] - bounds[i, 0]) / float(num_samples)
        order = np.argsort(uniform(num_samples), kind="stable")  # a uniform random ordering of the slices
        pts[:, i] = bounds[i, 0] + edge * order + edge * uniform(num_samples)
    return pts


def repeated_domain_starts(bounds, num_points, num_repeats, uniform):
    """RepeatedDomain::GenerateUniformPointsInDomain: [num_points][num_repeats][dim], one hypercube per repeat."""
    dim = np.asarray(bounds).size // 2
    out = np.empty((num_points, num_repeats, dim))
    for r in range(num_repeats):
        out[:, r, :] = latin_hypercube(bounds, num_points, uniform)
    return out


def limit_update(bounds, max_relative_change, x, step):
    """TensorProductDomain::LimitUpdate (gpp_domain.cpp:64-105), vectorised over leading axes; x, step [..., dim]."""
    b = np.asarray(bounds, dtype=np.float64).reshape(-1, 2)
    lo, hi = b[:, 0], b[:, 1]
    step = np.array(step, dtype=np.float64, copy=True)
    dist = np.minimum(x - lo, hi - x)
    big = np.abs(step) > max_relative_change * dist
    step = np.where(big, np.copysign(max_relative_change * dist, step), step)
    nxt = x + step
    below, above = nxt < lo, nxt > hi
    half = 0.5 * step
    step = np.where(below, np.where(x + half < lo, 0.5 * (lo - x), half), step)
    step = np.where(above, np.where(x + half > hi, 0.5 * (hi - x), half), step)
    return step


def _gd(params):
    p = params.optimizer_parameters
    return (int(p.num_multistarts), int(p.max_num_steps), int(p.max_num_restarts), int(p.num_steps_averaged), float(p.gamma),
            float(p.pre_mult), float(p.max_relative_change), float(p.tolerance))


def kg_values(dev_gp, num_fidelity, inner_gd, inner_bounds, discrete, Xq_all, Xp, num_mc, best_so_far, normals):
    r = dev_gp.kg_batch(inner_gd, inner_bounds, discrete, Xq_all, Xp, num_mc, best_so_far, normals, want_grad=False,
                        num_fidelity=num_fidelity)
    return r["kg_sum"] / num_mc


def gradient_ascent(grad_fn, gd, bounds, starts, on_step=None):
    """GradientDescentOptimizer::Optimize (gpp_optimization.hpp:619-705, 1144-1185) for every start at once.
    grad_fn(x [k][q][dim]) -> gradient [k][q][dim] of the objective being MAXIMISED.  starts [S][q][dim] -> end points."""
    _, max_steps, max_restarts, _, gamma, pre_mult, max_rel, tol = gd
    max_steps, max_restarts = int(max_steps), int(max_restarts)
    x = np.array(starts, dtype=np.float64, copy=True)
    S = x.shape[0]
    if max_restarts <= 0:
        return x
    step_tol = tol / float(max_steps)
    alive = np.ones(S, dtype=bool)          # restart loop still running
    for _ in range(max_restarts):
        if not alive.any():
            break
        x_begin = x.copy()
        running = alive.copy()              # inner GD loop still running
        for i in range(max_steps):
            idx = np.flatnonzero(running)
            if idx.size == 0:
                break
            alpha = pre_mult * float(i + 1) ** (-gamma)
            grad = grad_fn(x[idx])
            step = limit_update(bounds, max_rel, x[idx], alpha * grad)
            x[idx] += step
            norm = np.sqrt((step.reshape(idx.size, -1) ** 2).sum(axis=1))
            running[idx[norm < step_tol]] = False
            if on_step is not None:
                on_step(i, idx)
        delta = np.sqrt(((x_begin - x).reshape(S, -1) ** 2).sum(axis=1))
        alive &= delta > tol
    return x


def multistart_best(value_fn, grad_fn, gd, bounds, starts, floor_value=-np.inf, do_gradient_ascent=True):
    """Value at every start, the best TOP_K kept, restarted ascent on each, best end point by value (strict compare against
    floor_value): MultistartOptimizer (gpp_optimization.hpp:1472-1546) as driven by gpp_math.hpp:1683-1800 /
    gpp_knowledge_gradient_optimization.hpp:860-935.  Returns (best_point, best_value, found)."""
    starts = np.asarray(starts, dtype=np.float64)
    vals = np.asarray(value_fn(starts))
    if do_gradient_ascent:
        order = np.argsort(-vals, kind="stable")[:TOP_K]
        ends = gradient_ascent(grad_fn, gd, bounds, starts[order])
        end_vals = np.asarray(value_fn(ends))
    else:
        ends, end_vals = starts, vals
    best, best_val, found = np.zeros_like(starts[0]), floor_value, False
    for s in range(ends.shape[0]):
        if end_vals[s] > best_val:
            best, best_val, found = ends[s].copy(), float(end_vals[s]), True
    return best, best_val, found


def kg_gradient_ascent(dev_gp, num_fidelity, gd, inner_gd, bounds, inner_bounds, discrete, starts, Xp, num_mc, best_so_far,
                       normals, on_step=None):
    """gradient_ascent on q-KG: every live restart's gradient comes from ONE moe_kg_batch call per step."""
    def grad_fn(x):
        r = dev_gp.kg_batch(inner_gd, inner_bounds, discrete, x, Xp, num_mc, best_so_far, normals, want_grad=True,
                            num_fidelity=num_fidelity)
        return r["grad_sum"] / num_mc
    return gradient_ascent(grad_fn, gd, bounds, starts, on_step)


def kg_optimal_points(dev_gp, num_fidelity, optimizer_parameters, optimizer_parameters_inner, bounds, discrete, Xp,
                      num_to_sample, best_so_far, num_mc, randomness, starts=None):
    """ComputeKGOptimalPointsToSample (gpp_knowledge_gradient_optimization.cpp:490-551): multistart gradient ascent from
    Latin-hypercube starts, Latin-hypercube value search as the fall-back / null-optimiser path.
    Returns (best_points [q][dim], found_flag)."""
    from . import GPP, api
    d = dev_gp.d
    q = int(num_to_sample)
    bounds = np.asarray(bounds, dtype=np.float64).reshape(-1)[:2 * d]
    inner_gd = _gd(optimizer_parameters_inner)
    p = 0 if Xp is None else np.asarray(Xp).reshape(-1, d).shape[0]
    m = (q + p) * (1 + dev_gp.g)
    normals = randomness.normal_rng_vec[0].table(((num_mc + 1) // 2) * m)

    def lhc(count):  # RepeatedDomain::GenerateUniformPointsInDomain: one hypercube per repeat (gpp_domain.hpp:490-504)
        out = np.empty((count, q, d))
        for r in range(q):
            out[:, r, :] = api.latin_hypercube(randomness._next_uniform_seed(), bounds, count)
        return out

    use_gd = int(optimizer_parameters.optimizer_type) == int(GPP.OptimizerTypes.gradient_descent)
    best, found = np.zeros((q, d)), False
    if use_gd:
        gd = _gd(optimizer_parameters)
        if starts is None:
            starts = lhc(gd[0])
        starts = np.asarray(starts, dtype=np.float64).reshape(-1, q, d)
        best, _, found = dev_gp.kg_multistart(gd, inner_gd, bounds, discrete, starts, Xp, num_mc, best_so_far, normals,
                                              gradient_ascent=True, num_fidelity=num_fidelity)
    if not found:
        n_lhc = int(optimizer_parameters.num_random_samples or 0)
        if n_lhc > 0:
            best, _, found = dev_gp.kg_multistart(_gd(optimizer_parameters_inner), inner_gd, bounds, discrete, lhc(n_lhc), Xp,
                                                  num_mc, best_so_far, normals, gradient_ascent=False,
                                                  num_fidelity=num_fidelity)
    return best, found


def ei_optimal_points(dev_gp, optimizer_parameters, bounds, Xp, num_to_sample, best_so_far, num_mc, randomness):
    """ComputeOptimalPointsToSample (gpp_math.cpp:2369-2425): multistart gradient ascent on q,p-EI from Latin-hypercube
    starts (ComputeOptimalPointsToSampleWithRandomStarts, gpp_math.hpp:1836-1866), Latin-hypercube value search as the
    fall-back / null-optimiser path (gpp_python_expected_improvement.cpp:148-206).  Returns (best_points [q][dim], found)."""
    from . import GPP, api
    d = dev_gp.d
    q = int(num_to_sample)
    bounds = np.asarray(bounds, dtype=np.float64).reshape(-1)[:2 * d]
    p = 0 if Xp is None else np.asarray(Xp).reshape(-1, d).shape[0]
    normals = None if (q == 1 and p == 0) else randomness.normal_rng_vec[0].table(int(num_mc) * (q + p))

    def lhc(count):
        out = np.empty((count, q, d))
        for r in range(q):
            out[:, r, :] = api.latin_hypercube(randomness._next_uniform_seed(), bounds, count)
        return out

    use_gd = int(optimizer_parameters.optimizer_type) == int(GPP.OptimizerTypes.gradient_descent)
    best, found = np.zeros((q, d)), False
    if use_gd:
        gd = _gd(optimizer_parameters)
        best, _, found = dev_gp.ei_multistart(gd, bounds, lhc(gd[0]), Xp, num_mc, best_so_far, normals, gradient_ascent=True)
    if not found:
        n_lhc = int(optimizer_parameters.num_random_samples or 0)
        if n_lhc > 0:
            null_gd = (1, 1, 0, 0, 1.0, 1.0, 1.0, 0.0)
            best, _, found = dev_gp.ei_multistart(null_gd, bounds, lhc(n_lhc), Xp, num_mc, best_so_far, normals,
                                                  gradient_ascent=False)
    return best, found


def _lhc_starts(randomness, bounds, count, q, d):
    """RepeatedDomain::GenerateUniformPointsInDomain: one Latin hypercube per repeat (gpp_domain.hpp:490-504)."""
    from . import api
    out = np.empty((count, q, d))
    for r in range(q):
        out[:, r, :] = api.latin_hypercube(randomness._next_uniform_seed(), bounds, count)
    return out


def kg_mcmc_optimal_points(dev_mcmc, num_fidelity, optimizer_parameters, optimizer_parameters_inner, bounds, discrete_all, Xp,
                           num_to_sample, best_so_far, num_mc, randomness):
    """ComputeKGMCMCOptimalPointsToSample (gpp_knowledge_gradient_mcmc_optimization.cpp:236-296): kg_optimal_points on the
    MCMC-averaged, cost-scaled objective.  Returns (best_points [q][dim], found)."""
    from . import GPP
    d, q = dev_mcmc.d, int(num_to_sample)
    bounds = np.asarray(bounds, dtype=np.float64).reshape(-1)[:2 * d]
    inner_gd = _gd(optimizer_parameters_inner)
    p = 0 if Xp is None else np.asarray(Xp).reshape(-1, d).shape[0]
    normals = randomness.normal_rng_vec[0].table(((num_mc + 1) // 2) * (q + p) * (1 + dev_mcmc.g))
    use_gd = int(optimizer_parameters.optimizer_type) == int(GPP.OptimizerTypes.gradient_descent)
    best, found = np.zeros((q, d)), False
    if use_gd:
        gd = _gd(optimizer_parameters)
        best, _, found = dev_mcmc.kg_multistart(gd, inner_gd, bounds, discrete_all, _lhc_starts(randomness, bounds, gd[0], q, d), Xp,
                                                num_mc, best_so_far, normals, gradient_ascent=True, num_fidelity=num_fidelity)
    if not found:
        n_lhc = int(optimizer_parameters.num_random_samples or 0)
        if n_lhc > 0:
            best, _, found = dev_mcmc.kg_multistart(inner_gd, inner_gd, bounds, discrete_all,
                                                    _lhc_starts(randomness, bounds, n_lhc, q, d), Xp, num_mc, best_so_far, normals,
                                                    gradient_ascent=False, num_fidelity=num_fidelity)
    return best, found


def ei_mcmc_optimal_points(dev_mcmc, optimizer_parameters, bounds, Xp, num_to_sample, best_so_far, num_mc, randomness):
    """ComputeMCMCOptimalPointsToSample (gpp_expected_improvement_mcmc_optimization.cpp:300-388): ei_optimal_points on the
    MCMC-averaged EI.  Returns (best_points [q][dim], found)."""
    from . import GPP
    d, q = dev_mcmc.d, int(num_to_sample)
    bounds = np.asarray(bounds, dtype=np.float64).reshape(-1)[:2 * d]
    p = 0 if Xp is None else np.asarray(Xp).reshape(-1, d).shape[0]
    normals = None if (q == 1 and p == 0) else randomness.normal_rng_vec[0].table(int(num_mc) * (q + p))
    use_gd = int(optimizer_parameters.optimizer_type) == int(GPP.OptimizerTypes.gradient_descent)
    best, found = np.zeros((q, d)), False
    if use_gd:
        gd = _gd(optimizer_parameters)
        best, _, found = dev_mcmc.ei_multistart(gd, bounds, _lhc_starts(randomness, bounds, gd[0], q, d), Xp, num_mc, best_so_far,
                                                normals, gradient_ascent=True)
    if not found:
        n_lhc = int(optimizer_parameters.num_random_samples or 0)
        if n_lhc > 0:
            best, _, found = dev_mcmc.ei_multistart((1, 1, 0, 0, 1.0, 1.0, 1.0, 0.0), bounds,
                                                    _lhc_starts(randomness, bounds, n_lhc, q, d), Xp, num_mc, best_so_far, normals,
                                                    gradient_ascent=False)
    return best, found


def posterior_mean_optimization(dev_gp, num_fidelity, optimizer_parameters, bounds, initial_guess):
    """ComputeOptimalPosteriorMean from ONE start (moe_posterior_mean_optimize).  Returns (best_point, found_flag)."""
    size = dev_gp.d - num_fidelity
    best, _ = dev_gp.posterior_mean_optimize(_gd(optimizer_parameters), np.asarray(bounds, dtype=np.float64).reshape(-1)[:2 * size],
                                             np.asarray(initial_guess, dtype=np.float64).reshape(-1)[:size], num_fidelity)
    return best, True
