"""Outer multistart optimisation of q-KG / q-EI: thin host drivers over the C ABI (moe_kg_multistart,
moe_ei_multistart, moe_posterior_mean_optimize, moe_latin_hypercube -- csrc/multistart.hip), where every restart's value /
gradient is evaluated in ONE batched device pass per step instead of one OpenMP thread per restart.

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

The device drivers are pinned to the reference's own end points (tests/golden/ref_kg_multistart.npz, ref_fixtures.npz); a
numpy restatement of the same algorithm lives with the tests (tests/ms_restatement.py).
"""
import numpy as np


def _gd(params, domain_type=0):
    p = params.optimizer_parameters
    return (int(p.num_multistarts), int(p.max_num_steps), int(p.max_num_restarts), int(p.num_steps_averaged), float(p.gamma),
            float(p.pre_mult), float(p.max_relative_change), float(p.tolerance), int(domain_type))


def _domain_type(optimizer_parameters):
    """0 = tensor product, 1 = simplex intersection (GPP.DomainTypes; the dispatch of gpp_python_knowledge_gradient.cpp:288-296)."""
    return int(getattr(optimizer_parameters, "domain_type", 0))


def _simplex_box(bounds, d):
    """SimplexIntersectTensorProductDomain's constructor (gpp_domain.cpp:107-141): the box clipped to the unit hypercube; an empty
    intersection -- an empty clipped interval, or a 'lower left' corner whose coordinates sum to >= 1 -- is the reference's
    BoundsException, raised HERE, before any point is drawn (the rejection loop below would otherwise grow its draw
    5 x per attempt and never keep a point)."""
    from . import api
    b = np.asarray(bounds, dtype=np.float64).reshape(d, 2).copy()
    b[:, 0] = np.maximum(b[:, 0], 0.0)
    b[:, 1] = np.minimum(b[:, 1], 1.0)
    corner_sum = float(b[:, 0].sum())
    if corner_sum >= 1.0 or bool(np.any(b[:, 0] > b[:, 1])):
        raise api.BoundsException("Simplex/Tensor product intersection is EMPTY; 'lower left' corner coordinate sum out of bounds or "
                                  "bounding boxes do not intersect.", corner_sum, 0.0, 1.0)
    return b.reshape(-1)


def _in_unit_simplex(pts):
    """CheckPointInUnitSimplex (gpp_geometry.hpp:313-325), rows of pts."""
    return np.all(pts >= 0.0, axis=1) & (pts.sum(axis=1) - 4.0 * np.finfo(float).eps <= 1.0)


def _domain_points(randomness, bounds, count, d, domain_type):
    """DomainType::GenerateUniformPointsInDomain: a Latin hypercube in the box; for the simplex intersection the points outside the
    unit simplex are rejected and the draw is repeated with more points, the reference's retry rule (gpp_domain.cpp:179-232: at most
    ten attempts, content with 90 % of the request, growth 1 / ratio capped at 5 x).  May return fewer than `count` points."""
    from . import api
    if domain_type == 0:
        return api.latin_hypercube(randomness._next_uniform_seed(), bounds, count)
    box = _simplex_box(bounds, d)
    n_local = max(10, count)
    kept = np.empty((0, d))
    for _ in range(10):
        pts = api.latin_hypercube(randomness._next_uniform_seed(), box, n_local)
        kept = pts[_in_unit_simplex(pts)][:count]
        if len(kept) >= 0.9 * count:
            break
        ratio = len(kept) / float(count)
        n_local = n_local * 5 if ratio < 0.2 else int(np.ceil(n_local / ratio))
        if n_local * d > _MAX_DRAW_DOUBLES:  # (a sliver of an intersection: the reference would keep multiplying; bounded here)
            break
    return kept


_MAX_DRAW_DOUBLES = 1 << 26  # 512 MB per Latin-hypercube draw of the simplex rejection loop


def _starts(randomness, bounds, count, q, d, domain_type=0):
    """RepeatedDomain::GenerateUniformPointsInDomain (gpp_domain.hpp:490-510): one point set per repeat, transposed; when a repeat
    comes up short (simplex rejection) every start set is cut to the shortest.  NO surviving start is not handled here: the multistart
    entry points refuse num_starts = 0 with the reference's BoundsException ("num_multistarts must be > 1", gpp_optimization.hpp:1478),
    which is what its own drivers do with the count GenerateUniformPointsInDomain returns."""
    sets = []
    n = count
    for _ in range(q):
        pts = _domain_points(randomness, bounds, n, d, domain_type)
        n = min(n, len(pts))
        sets.append(pts)
    out = np.empty((n, q, d))
    for r in range(q):
        out[:, r, :] = sets[r][:n]
    return out


def kg_optimal_points(dev_gp, num_fidelity, optimizer_parameters, optimizer_parameters_inner, bounds, discrete, Xp,
                      num_to_sample, best_so_far, num_mc, randomness, starts=None, comm=None):
    """ComputeKGOptimalPointsToSample (gpp_knowledge_gradient_optimization.cpp:490-551): multistart gradient ascent from
    Latin-hypercube starts, Latin-hypercube value search as the fall-back / null-optimiser path.
    Returns (best_points [q][dim], found_flag).
    comm (r5: dist.Exchange): one process per GPU, every rank calling this with the same arguments and the same `randomness` seeds
    (the start sets and the normal table are drawn identically on every rank) and a GP over the same data on its own device: the
    restarts of every optimiser step are dealt to the ranks, one all-gather per step (moe_kg_multistart_comm), and every rank
    returns the single-rank answer bit for bit."""
    from . import GPP, api
    d = dev_gp.d
    q = int(num_to_sample)
    bounds = np.asarray(bounds, dtype=np.float64).reshape(-1)[:2 * d]
    # (the reference builds the inner domain -- every sample's posterior-mean optimisation -- of the OUTER parameters' type,
    #  gpp_python_knowledge_gradient.cpp:279-296)
    dom = _domain_type(optimizer_parameters)
    inner_gd = _gd(optimizer_parameters_inner, dom)
    p = 0 if Xp is None else np.asarray(Xp).reshape(-1, d).shape[0]
    m = (q + p) * (1 + dev_gp.g)
    normals = randomness.normal_rng_vec[0].table(((num_mc + 1) // 2) * m)

    def lhc(count):
        return _starts(randomness, bounds, count, q, d, dom)

    use_gd = int(optimizer_parameters.optimizer_type) == int(GPP.OptimizerTypes.gradient_descent)
    best, found = np.zeros((q, d)), False
    if use_gd:
        gd = _gd(optimizer_parameters, dom)
        if starts is None:
            starts = lhc(gd[0])
        starts = np.asarray(starts, dtype=np.float64).reshape(-1, q, d)
        best, _, found = dev_gp.kg_multistart(gd, inner_gd, bounds, discrete, starts, Xp, num_mc, best_so_far, normals,
                                              gradient_ascent=True, num_fidelity=num_fidelity, comm=comm)
    if not found:
        n_lhc = int(optimizer_parameters.num_random_samples or 0)
        if n_lhc > 0:
            best, _, found = dev_gp.kg_multistart(inner_gd, inner_gd, bounds, discrete, lhc(n_lhc), Xp,
                                                  num_mc, best_so_far, normals, gradient_ascent=False,
                                                  num_fidelity=num_fidelity, comm=comm)
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

    dom = _domain_type(optimizer_parameters)

    def lhc(count):
        return _starts(randomness, bounds, count, q, d, dom)

    use_gd = int(optimizer_parameters.optimizer_type) == int(GPP.OptimizerTypes.gradient_descent)
    best, found = np.zeros((q, d)), False
    if use_gd:
        gd = _gd(optimizer_parameters, dom)
        best, _, found = dev_gp.ei_multistart(gd, bounds, lhc(gd[0]), Xp, num_mc, best_so_far, normals, gradient_ascent=True)
    if not found:
        n_lhc = int(optimizer_parameters.num_random_samples or 0)
        if n_lhc > 0:
            null_gd = (1, 1, 0, 0, 1.0, 1.0, 1.0, 0.0)
            best, _, found = dev_gp.ei_multistart(null_gd, bounds, lhc(n_lhc), Xp, num_mc, best_so_far, normals,
                                                  gradient_ascent=False)
    return best, found




def kg_mcmc_optimal_points(dev_mcmc, num_fidelity, optimizer_parameters, optimizer_parameters_inner, bounds, discrete_all, Xp,
                           num_to_sample, best_so_far, num_mc, randomness, comm=None):
    """ComputeKGMCMCOptimalPointsToSample (gpp_knowledge_gradient_mcmc_optimization.cpp:236-296): kg_optimal_points on the
    MCMC-averaged, cost-scaled objective.  Returns (best_points [q][dim], found).
    comm (r5: dist.Exchange): the ensemble's MEMBERS are dealt to the ranks -- dev_mcmc = DeviceGPMCMC(..., members=
    dist.shard_members(num_mcmc, rank, world)) on this rank's device, discrete_all / best_so_far for the whole ensemble, the same
    `randomness` seeds everywhere; one all-gather of the per-member values per optimiser step (moe_kg_mcmc_multistart_comm), the
    single-rank answer bit for bit on every rank."""
    from . import GPP
    d, q = dev_mcmc.d, int(num_to_sample)
    bounds = np.asarray(bounds, dtype=np.float64).reshape(-1)[:2 * d]
    dom = _domain_type(optimizer_parameters)
    inner_gd = _gd(optimizer_parameters_inner, dom)
    p = 0 if Xp is None else np.asarray(Xp).reshape(-1, d).shape[0]
    normals = randomness.normal_rng_vec[0].table(((num_mc + 1) // 2) * (q + p) * (1 + dev_mcmc.g))
    use_gd = int(optimizer_parameters.optimizer_type) == int(GPP.OptimizerTypes.gradient_descent)
    best, found = np.zeros((q, d)), False
    if use_gd:
        gd = _gd(optimizer_parameters, dom)
        best, _, found = dev_mcmc.kg_multistart(gd, inner_gd, bounds, discrete_all, _starts(randomness, bounds, gd[0], q, d, dom), Xp,
                                                num_mc, best_so_far, normals, gradient_ascent=True, num_fidelity=num_fidelity,
                                                comm=comm)
    if not found:
        n_lhc = int(optimizer_parameters.num_random_samples or 0)
        if n_lhc > 0:
            best, _, found = dev_mcmc.kg_multistart(inner_gd, inner_gd, bounds, discrete_all,
                                                    _starts(randomness, bounds, n_lhc, q, d, dom), Xp, num_mc, best_so_far, normals,
                                                    gradient_ascent=False, num_fidelity=num_fidelity, comm=comm)
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
    dom = _domain_type(optimizer_parameters)
    if use_gd:
        gd = _gd(optimizer_parameters, dom)
        best, _, found = dev_mcmc.ei_multistart(gd, bounds, _starts(randomness, bounds, gd[0], q, d, dom), Xp, num_mc, best_so_far,
                                                normals, gradient_ascent=True)
    if not found:
        n_lhc = int(optimizer_parameters.num_random_samples or 0)
        if n_lhc > 0:
            best, _, found = dev_mcmc.ei_multistart((1, 1, 0, 0, 1.0, 1.0, 1.0, 0.0), bounds,
                                                    _starts(randomness, bounds, n_lhc, q, d, dom), Xp, num_mc, best_so_far, normals,
                                                    gradient_ascent=False)
    return best, found


def posterior_mean_optimization(dev_gp, num_fidelity, optimizer_parameters, bounds, initial_guess):
    """ComputeOptimalPosteriorMean from ONE start (moe_posterior_mean_optimize).  Returns (best_point, found_flag)."""
    size = dev_gp.d - num_fidelity
    gd = _gd(optimizer_parameters, _domain_type(optimizer_parameters))  # (gpp_python_knowledge_gradient.cpp:327-341)
    best, _ = dev_gp.posterior_mean_optimize(gd, np.asarray(bounds, dtype=np.float64).reshape(-1)[:2 * size],
                                             np.asarray(initial_guess, dtype=np.float64).reshape(-1)[:size], num_fidelity)
    return best, True
