"""Device self-test behind `GPP.run_cpp_tests()`.

The reference's `run_cpp_tests` (cpp/gpp_python_test.cpp:60-301) runs its C++ unit suites and returns the number of failures:
finite-difference "ping" tests of every gradient (RunGPTests, RunKGTests, RunLogLikelihoodPingTests; cpp/gpp_test_utils.hpp:
PingDerivative), analytic-vs-Monte-Carlo EI consistency (RunEIConsistencyTests), the random-number and point-generator checks
(RandomNumberGeneratorContainerTest, RunRandomPointGeneratorTests), linear algebra (RunLinearAlgebraTests) and optimiser end-to-end
tests.  This module runs the same KINDS of check against the HIP library -- no oracle, no reference, no CPU fallback: every quantity
is computed on the GPU and checked against itself through an identity (central differences of the value entry point against the
gradient entry point; L L^T = A; analytic = Monte Carlo within its standard error; an optimiser's end point is no worse than its
start).  It is a smoke-level acceptance test for an installation; parity with the reference is the job of tests/ (DESIGN section 3).

    from cornell_moe_amd import selftest; failures = selftest.run(verbose=True)
"""
import numpy as np

from . import api

_H = 1.0e-5      # central-difference step (inputs are O(1)): truncation ~1e-10, rounding ~1e-11 relative
_TOL = 2.0e-6    # relative to the largest gradient entry


def _problem(seed=11, n=40, d=3):
    rng = np.random.default_rng(seed)
    X = rng.uniform(size=(n, d))
    y = np.sin(3.0 * X).sum(1, keepdims=True) + 0.05 * rng.standard_normal((n, 1))
    hyper = np.concatenate([[1.1], 0.45 + 0.1 * np.arange(d)])
    return rng, X, y, hyper


def _fd(f, x):
    """Central differences of array-valued f at x (any shape): out[..., *x.shape]."""
    x = np.array(x, dtype=np.float64)
    f0 = np.asarray(f(x))
    out = np.zeros(f0.shape + x.shape)
    for idx in np.ndindex(*x.shape):
        xp, xm = x.copy(), x.copy()
        xp[idx] += _H
        xm[idx] -= _H
        out[(Ellipsis,) + idx] = (np.asarray(f(xp)) - np.asarray(f(xm))) / (2.0 * _H)
    return out


def _close(a, b, tol=_TOL):
    a, b = np.asarray(a), np.asarray(b)
    return bool(np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max()))


def check_linear_algebra():
    """RunLinearAlgebraTests' core: L L^T = A, L^-1 L = I on a random SPD matrix (device factorisation + explicit inverse)."""
    rng = np.random.default_rng(3)
    B = rng.standard_normal((70, 70))
    A = B @ B.T + 70.0 * np.eye(70)
    L, Linv = api.debug_cholesky(A)
    L = np.tril(L)
    return _close(L @ L.T, A, 1e-13) and _close(np.tril(Linv) @ L, np.eye(70), 1e-12)


def check_ping_gp_mean():
    """PingGPMean (cpp/gpp_math_test.cpp): d mu(x_i) / d x_i."""
    rng, X, y, hyper = _problem()
    G = api.DeviceGP(hyper, X, y, [0.01])
    pts = rng.uniform(size=(3, 3))
    grad = G.grad_mean(pts).reshape(3, 3)             # [point][dim]
    fd = _fd(lambda p: G.mean(p), pts)                # [point][point][dim]
    own = np.stack([fd[i, i] for i in range(3)])
    cross = max(np.abs(fd[i, j]).max() for i in range(3) for j in range(3) if i != j)
    return _close(grad, own) and cross <= 1e-9


def _variance_matrix(G, pts, chol):
    m = pts.shape[0]
    v = (G.cholesky_variance(pts) if chol else G.variance(pts)).reshape(m, m).T   # col-major -> [row][col]
    return np.tril(v)


def check_ping_gp_variance(chol=False):
    """PingGPVariance / PingGPCholeskyVariance: d Var[i][j] / d x_p[k] and the same of its Cholesky factor (lower triangles)."""
    rng, X, y, hyper = _problem(seed=12)
    G = api.DeviceGP(hyper, X, y, [0.01])
    m, d = 3, 3
    pts = rng.uniform(size=(m, d))
    raw = (G.grad_cholesky_variance(pts, m) if chol else G.grad_variance(pts, m)).reshape(m, m, m, d)   # [p][row][col][dim]
    fd = _fd(lambda p: _variance_matrix(G, p, chol), pts)                                                # [row][col][p][dim]
    ok = True
    for p in range(m):
        for i in range(m):
            for j in range(i + 1):
                ok = ok and _close(raw[p, i, j], fd[i, j, p], 5e-6)
    return ok


def check_ei_consistency():
    """RunEIConsistencyTests: analytic 1-EI = Monte-Carlo EI with q = 1 within its standard error; both gradients agree."""
    rng, X, y, hyper = _problem(seed=13)
    G = api.DeviceGP(hyper, X, y, [0.01])
    x = rng.uniform(size=(1, 3))
    best = float(G.mean(x)[0])  # an improvement in half of the samples: the Monte-Carlo error is that of a bulk, not a tail, estimate
    ea, ga = G.ei_analytic_batch(x, best)
    M = 200000
    em, gm = G.ei(x, None, M, best, api.normal_draws(5, M))
    sd = np.sqrt(max(G.variance(x)[0], 1e-30))
    return abs(em - ea[0]) <= 5.0 * sd / np.sqrt(M) + 1e-12 and np.abs(gm[0] - ga[0]).max() <= 0.03 * max(np.abs(ga[0]).max(), 1e-6)


def check_ping_ei():
    """PingEIGeneral: Monte-Carlo q,p-EI on a fixed normal table is differentiable almost everywhere; analytic 1-EI everywhere."""
    rng, X, y, hyper = _problem(seed=14)
    G = api.DeviceGP(hyper, X, y, [0.01])
    best = float(y.min()) + 0.3
    Xq, Xp = rng.uniform(size=(2, 3)), rng.uniform(size=(1, 3))
    Z = rng.standard_normal((512, 3))
    _, g = G.ei(Xq, Xp, 512, best, Z)
    fd = _fd(lambda q: G.ei(q, Xp, 512, best, Z, want_grad=False)[0], Xq)
    x1 = rng.uniform(size=(1, 3))
    _, ga = G.ei_analytic_batch(x1, best)
    fda = _fd(lambda q: G.ei_analytic_batch(q, best, want_grad=False)[0][0], x1)
    return _close(g, fd, 1e-5) and _close(ga, fda, 1e-5)


def check_ping_kg():
    """RunKGTests: (a) value-only evaluation = the value of the gradient evaluation, bit for bit; (b) with the inner optimiser held in
    place (one step of relative length 1e-13: the ABI rejects zero steps) KG is the discrete knowledge gradient E[best - min_j mu_j] on
    a fixed normal table, differentiable almost everywhere: ping it; (c) the optimised KG is at least the discrete one (the inner
    optimisation can only lower every sample's minimum)."""
    rng, X, y, hyper = _problem(seed=15)
    G = api.DeviceGP(hyper, X, y, [0.01])
    disc = rng.uniform(size=(6, 3))
    best = float(G.additional_mean(disc).min())
    Xq = rng.uniform(size=(2, 3))
    bounds = np.tile([0.0, 1.0], 3)
    M = 64
    Z = rng.standard_normal((M // 2, 2))
    gd_on = (1, 6, 1, 3, 0.0, 1.0, 0.1, 1e-10)
    a = G.kg(gd_on, bounds, disc, Xq, None, M, best, Z)
    b = G.kg(gd_on, bounds, disc, Xq, None, M, best, Z, want_grad=False)
    ok = a["kg_sum"] == b["kg_sum"] and np.all(np.isfinite(a["grad_sum"]))
    gd_off = (1, 1, 1, 3, 0.0, 1e-13, 1e-13, 1e-10)
    c = G.kg(gd_off, bounds, disc, Xq, None, M, best, Z)
    fd = _fd(lambda q: G.kg(gd_off, bounds, disc, q, None, M, best, Z, want_grad=False)["kg_sum"], Xq)
    ok = ok and _close(np.asarray(c["grad_sum"]).reshape(2, 3), fd, 1e-5)
    return bool(ok and a["kg_sum"] >= c["kg_sum"] - 1e-9 * abs(c["kg_sum"]))


def check_ping_log_likelihood():
    """RunLogLikelihoodPingTests: d log p(y | X, theta) / d theta."""
    rng, X, y, hyper = _problem(seed=16)
    LL = api.LogLikelihood(X, y)
    th = np.concatenate([hyper, [0.02]])
    g = LL.grad(th)
    fd = _fd(lambda t: LL.evaluate(t[None])[0], th)
    return _close(g, fd, 1e-5)


def check_random_sources():
    """RandomNumberGeneratorContainerTest / RunRandomPointGeneratorTests: seeded normal streams are reproducible, distinct across
    seeds, standard normal; a Latin hypercube puts exactly one point in every slice of every edge, inside the domain."""
    a, b, c = api.normal_draws(7, 100000), api.normal_draws(7, 100000), api.normal_draws(8, 100000)
    ok = np.array_equal(a, b) and not np.array_equal(a, c) and abs(a.mean()) < 0.02 and abs(a.var() - 1.0) < 0.02
    bounds = np.array([[-1.0, 2.0], [0.5, 0.75], [3.0, 9.0]])
    P = api.latin_hypercube(21, bounds, 16)
    for k in range(3):
        lo, hi = bounds[k]
        ok = ok and np.all(P[:, k] >= lo) and np.all(P[:, k] <= hi)
        ok = ok and sorted(np.minimum(((P[:, k] - lo) / (hi - lo) * 16).astype(int), 15)) == list(range(16))
    return bool(ok)


def check_optimisers():
    """ExpectedImprovementOptimizationTest / KnowledgeGradientOptimizationTest in spirit: the posterior-mean optimiser ends no worse than
    it starts and inside the domain; the best of a q-KG multistart is no worse than its best start."""
    rng, X, y, hyper = _problem(seed=17)
    G = api.DeviceGP(hyper, X, y, [0.01])
    bounds = np.tile([0.0, 1.0], 3)
    x0 = rng.uniform(size=3)
    x1, v1 = G.posterior_mean_optimize((1, 40, 2, 3, 0.0, 1.0, 0.2, 1e-9), bounds, x0)
    # (the objective is -mu, as in the reference's PosteriorMeanEvaluator: the optimiser ascends it)
    ok = v1 >= G.posterior_mean(x0, want_grad=False)[0] - 1e-12 and np.all(x1 >= 0.0) and np.all(x1 <= 1.0)
    disc = rng.uniform(size=(60, 3))
    best = float(G.additional_mean(disc).min())
    starts = rng.uniform(0.2, 0.8, size=(6, 2, 3))
    M = 32
    Z = rng.standard_normal((M // 2, 2))
    inner = (1, 4, 1, 3, 0.0, 1.0, 0.1, 1e-10)
    quirks = api.get_reference_quirks()
    api.set_reference_quirks(0)  # every evaluation of the driver on a fresh state: its values are comparable with single evaluations
    try:
        # short steps: every restart climbs
        pt, val, found = G.kg_multistart((6, 6, 1, 3, 0.7, 0.02, 0.05, 1e-12), inner, bounds, disc, starts, None, M, best, Z)
    finally:
        api.set_reference_quirks(1 if quirks else 0)
    pt = np.asarray(pt)
    at_starts = G.kg_batch(inner, bounds, disc, starts, None, M, best, Z, want_grad=False)["kg_sum"] / M
    at_end = G.kg_batch(inner, bounds, disc, pt.reshape(1, 2, 3), None, M, best, Z, want_grad=False)["kg_sum"][0] / M
    ok = ok and found and np.all(pt >= 0.0) and np.all(pt <= 1.0)
    return bool(ok and abs(at_end - val) <= 1e-9 * max(1.0, abs(val)) and at_end >= at_starts.max() - 1e-9)


CHECKS = (
    ("linear algebra: Cholesky factor and inverse factor", check_linear_algebra),
    ("ping GP mean", check_ping_gp_mean),
    ("ping GP variance", lambda: check_ping_gp_variance(False)),
    ("ping GP cholesky variance", lambda: check_ping_gp_variance(True)),
    ("EI consistency: analytic vs Monte Carlo", check_ei_consistency),
    ("ping EI (Monte Carlo and analytic)", check_ping_ei),
    ("KG: value-only = value of the gradient call, ping of the discrete KG", check_ping_kg),
    ("ping log marginal likelihood", check_ping_log_likelihood),
    ("random number sources and Latin hypercube", check_random_sources),
    ("optimisers: posterior-mean descent, q-KG multistart", check_optimisers),
)


def run(verbose=False, checks=CHECKS):
    """Number of failed checks (0 = all passed), as the reference's run_cpp_tests returns its failure count.  A check that raises
    counts as failed.  Needs the GPU: the library has no CPU path (the first call raises the library's own error otherwise)."""
    from . import _lib
    _lib.require_gpu()
    failures = 0
    for name, fn in checks:
        try:
            ok = bool(fn())
            note = ""
        except Exception as e:  # noqa: BLE001 -- a raising check is a failing check; its message is reported
            ok, note = False, " (%s: %s)" % (type(e).__name__, e)
        failures += 0 if ok else 1
        if verbose:
            print("%s: %s%s" % ("SUCCESS" if ok else "FAILURE", name, note))
    return failures
