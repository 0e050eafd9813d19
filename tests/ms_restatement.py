"""TEST INFRASTRUCTURE: numpy restatement of the reference's outer multistart optimiser -- Latin hypercube
(gpp_random.cpp:173-194), TensorProductDomain::LimitUpdate (gpp_domain.cpp:64-105), GradientDescentOptimizer::Optimize
(gpp_optimization.hpp:619-705, 1144-1185) and MultistartOptimizer (gpp_optimization.hpp:1472-1546) as driven by
gpp_math.hpp:1683-1800 / gpp_knowledge_gradient_optimization.hpp:860-935 -- over caller-supplied value / gradient callables.
tests/test_oracle.py pins it to the reference's end points (golden fixtures); the GPU tests use it as a second check of the
C++ drivers in csrc/multistart.hip.  Not part of the product package.
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


def kg_values(dev_gp, num_fidelity, inner_gd, inner_bounds, discrete, Xq_all, Xp, num_mc, best_so_far, normals):
    r = dev_gp.kg_batch(inner_gd, inner_bounds, discrete, Xq_all, Xp, num_mc, best_so_far, normals, want_grad=False,
                        num_fidelity=num_fidelity)
    return r["kg_sum"] / num_mc


def gradient_ascent(grad_fn, gd, bounds, starts, on_step=None, carry_div=None):
    """GradientDescentOptimizer::Optimize (gpp_optimization.hpp:619-705, 1144-1185) for every start at once.
    grad_fn(x [k][q][dim]) -> gradient [k][q][dim] of the objective being MAXIMISED.  starts [S][q][dim] -> end points.
    carry_div = num_mcmc reproduces KnowledgeGradientMCMCEvaluator::ComputeGradKnowledgeGradient's use of its OUTPUT as an
    accumulator (gpp_knowledge_gradient_mcmc_optimization.cpp:163-166: ``grad_KG[k] += temp[k]`` into the vector that
    GradientDescentOptimization allocates once per restart and reuses every step, gpp_optimization.hpp:626, 646): without
    fidelity cost, step i of a restart sees G_i = g_i + G_{i-1} / num_mcmc instead of the gradient g_i."""
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
        carried = np.zeros_like(x)
        for i in range(max_steps):
            idx = np.flatnonzero(running)
            if idx.size == 0:
                break
            alpha = pre_mult * float(i + 1) ** (-gamma)
            grad = grad_fn(x[idx])
            if carry_div is not None:
                grad = grad + carried[idx] / float(carry_div)
                carried[idx] = grad
            step = limit_update(bounds, max_rel, x[idx], alpha * grad)
            x[idx] += step
            norm = np.sqrt((step.reshape(idx.size, -1) ** 2).sum(axis=1))
            running[idx[norm < step_tol]] = False
            if on_step is not None:
                on_step(i, idx)
        delta = np.sqrt(((x_begin - x).reshape(S, -1) ** 2).sum(axis=1))
        alive &= delta > tol
    return x


def top_k_order(vals, k=TOP_K):
    """The reference's top-20 selection, tie for tie (gpp_knowledge_gradient_optimization.hpp:895-921, gpp_math.hpp:1717-1738): a
    std::priority_queue of (-value, index) pairs -- a max-heap, so its top is the LOWEST-valued kept start and, among equal
    values, the one with the larger index -- is filled with the first k starts; a later start replaces the top only if its
    value is strictly larger; the kept starts are then popped (lowest value first, equal values by descending index) into the
    list the optimiser walks.  MultistartOptimizer's strict compare makes the first of equal end values in that list win."""
    import heapq
    heap = []  # python's heapq is a min-heap: store the negated pair (value, -index)
    for i, v in enumerate(vals):
        if i < k:
            heapq.heappush(heap, (float(v), -i))
        elif -heap[0][0] > -float(v):
            heapq.heapreplace(heap, (float(v), -i))
    return np.array([-heapq.heappop(heap)[1] for _ in range(len(heap))], dtype=int)


def multistart_best(value_fn, grad_fn, gd, bounds, starts, floor_value=-np.inf, do_gradient_ascent=True, carry_div=None):
    """Value at every start, the best TOP_K kept, restarted ascent on each, best end point by value (strict compare against
    floor_value): MultistartOptimizer (gpp_optimization.hpp:1472-1546) as driven by gpp_math.hpp:1683-1800 /
    gpp_knowledge_gradient_optimization.hpp:860-935.  Returns (best_point, best_value, found)."""
    starts = np.asarray(starts, dtype=np.float64)
    vals = np.asarray(value_fn(starts))
    if do_gradient_ascent:
        order = top_k_order(vals)
        ends = gradient_ascent(grad_fn, gd, bounds, starts[order], carry_div=carry_div)
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


def kg_mcmc_multistart_reference(kg_sum_fn, grad_sum_fn, num_mcmc, num_fidelity, gd, bounds, starts):
    """ComputeKGMCMCOptimalPointsToSampleViaMultistartGradientDescent (gpp_knowledge_gradient_mcmc_optimization.hpp:665-760) as the
    reference EXECUTES it, including what its state object does (gpp_knowledge_gradient_mcmc_optimization.cpp:163-166, 186-195;
    .hpp:439-441):
      * KnowledgeGradientMCMCState::SetCurrentPoint forwards all q points to the per-GP states but copies only the FIRST point
        into its own union_of_points, which is what GetCurrentPoint returns and what the fidelity cost reads: the optimiser
        sees [moved first point, the other points of the state's construction point = starts[0]];
      * ComputeGradKnowledgeGradient accumulates into its output, which GradientDescentOptimization allocates once per restart:
        G_i = ((G_{i-1} + sum_i) / num_mcmc * cost - KG * gradcost) / cost^2;
      * the per-GP states keep the discretised set of starts[0] (as in the single-GP driver).
    kg_sum_fn(x [q][d]) -> sum over the GPs of KG_i at x; grad_sum_fn(x) -> (that sum, sum of the per-GP gradients [q][d]); both
    must evaluate on states built at starts[0] (head).  Returns (best_point [q][d], best_value, found)."""
    _, max_steps, max_restarts, _, gamma, pre_mult, max_rel, tol = gd
    max_steps, max_restarts = int(max_steps), int(max_restarts)
    starts = np.asarray(starts, dtype=np.float64)
    S, q, d = starts.shape
    head = starts[0]

    def seen_of(actual_first):
        out = head.copy()
        out[0] = actual_first
        return out

    def cost_of(seen):
        if num_fidelity == 0:
            return 1.0, np.zeros_like(seen)
        pc = np.prod(seen[:, d - num_fidelity:], axis=1)
        cost, index = 0.0, -1
        for k in range(q):
            if cost < pc[k]:
                cost, index = float(pc[k]), k
        g = np.zeros_like(seen)
        g[index, d - num_fidelity:] = cost / seen[index, d - num_fidelity:]
        return cost, g

    vals = np.array([kg_sum_fn(starts[s]) / (num_mcmc * cost_of(seen_of(starts[s][0]))[0]) for s in range(S)])
    order = top_k_order(vals)
    best, best_val, found = seen_of(starts[order[0]][0]), -np.inf, False
    for s in order:
        actual = starts[s].copy()          # what the per-GP states hold
        seen = seen_of(actual[0])          # what GetCurrentPoint returns
        if max_restarts > 0:
            for _ in range(max_restarts):
                cur = seen.copy()
                nxt = seen.copy()
                G = np.zeros_like(nxt)
                for i in range(max_steps):
                    alpha = pre_mult * float(i + 1) ** (-gamma)
                    kg_sum, gsum = grad_sum_fn(actual)
                    cost, gcost = cost_of(seen)
                    G = ((G + gsum) / num_mcmc * cost - (kg_sum / num_mcmc) * gcost) / (cost * cost)
                    step = limit_update(bounds, max_rel, nxt, alpha * G)
                    nxt = nxt + step
                    actual = nxt.copy()
                    seen = seen_of(nxt[0])
                    if np.sqrt((step ** 2).sum()) < tol / float(max_steps):
                        break
                if not (np.sqrt(((cur - seen) ** 2).sum()) > tol):
                    break
        v = kg_sum_fn(actual) / (num_mcmc * cost_of(seen)[0])
        if v > best_val:
            best, best_val, found = seen.copy(), float(v), True
    return best, best_val, found
