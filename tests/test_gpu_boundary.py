"""GPU flow through the drop-in boundary: the reference's Python call sequence on the wrapper mirror
-> cornell_moe_amd.GPP -> C ABI -> HIP kernels, checked against the oracle fed the very same normal table."""
import numpy as np
import pytest

from helpers import TOL, rel

pytestmark = pytest.mark.gpu


def _setup(seed=31, n=120, d=3, q=2, p=1, P=7, M=300, derivs=()):
    import wrappers_mirror as cw
    from cornell_moe_amd.workloads import make_workload
    w = make_workload(seed=seed, n=n, d=d, q=q, M=M, P=P, derivs=derivs, p=p)
    hd = cw.HistoricalData(dim=d, num_derivatives=len(derivs))
    hd.append_sample_points([cw.SamplePoint(w.X[i], w.y[i], 0.01) for i in range(n)])
    gp = cw.GaussianProcess(cw.SquareExponential(w.hyperparameters), w.noise, hd, list(derivs))
    return cw, w, gp


def test_gaussian_process_wrapper_layouts():
    from oracle import orc
    cw, w, gp = _setup()
    O = orc.OrcGP(1, w.alpha, w.lengths, w.X, w.y, w.noise, ())  # Matern-5/2 whatever the covariance class is called
    pts = w.query[:5]
    assert gp.dim == 3 and gp.num_sampled == 120
    assert rel(gp.compute_mean_of_points(pts), O.mean(pts)) < TOL["q_mean"]
    assert rel(gp.compute_mean_of_additional_points(pts), O.mean(pts)) < TOL["q_mean"]
    var = gp.compute_variance_of_points(pts)
    assert var.shape == (5, 5) and np.array_equal(var, var.T)
    assert rel(var, O.var(pts).reshape(5, 5)) < TOL["q_var"]
    chol = gp.compute_cholesky_variance_of_points(pts)
    assert np.allclose(np.triu(chol, 1), 0.0) and rel(chol @ chol.T, var) < 1e-10
    gm = gp.compute_grad_mean_of_points(pts)
    assert gm.shape == (5, 1, 3) and rel(gm.ravel(), O.grad_mean(pts)) < TOL["q_grad_mean"]
    gv = gp.compute_grad_variance_of_points(pts, 2)
    assert gv.shape == (2, 5, 5, 3) and rel(gv.ravel(), O.grad_var(pts, 2)) < TOL["q_grad_var"]
    gc = gp.compute_grad_cholesky_variance_of_points(pts, 2)
    assert rel(gc.ravel(), O.grad_chol_var(pts, 2)) < TOL["q_grad_chol_var"]
    gp.add_sampled_points([cw.SamplePoint(w.query[50], [0.3], 0.01), cw.SamplePoint(w.query[51], [0.1], 0.01)])
    O2 = orc.OrcGP(1, w.alpha, w.lengths, np.vstack([w.X, w.query[50:52]]), np.vstack([w.y, [[0.3], [0.1]]]), w.noise, ())
    assert gp.num_sampled == 122 and rel(gp.compute_mean_of_points(pts), O2.mean(pts)) < TOL["q_mean"]
    s = gp.sample_point_from_gp(pts[0])
    assert s.shape == (1,) and np.isfinite(s[0])


def test_knowledge_gradient_wrapper_flow():
    from cornell_moe_amd import GPP
    from cornell_moe_amd.api import normal_draws
    from oracle import orc
    cw, w, gp = _setup()
    O = orc.OrcGP(1, w.alpha, w.lengths, w.X, w.y, w.noise, ())
    ps = cw.PosteriorMean(gp, 0)
    inner = cw.GradientDescentOptimizer(cw.TensorProductDomain([[0.0, 1.0]] * 3), ps,
                                        cw.GradientDescentParameters(1, 6, 1, 3, 0.0, 1.0, 0.1, 1e-10))
    rnd = GPP.RandomnessSourceContainer(1)
    rnd.SetExplicitNormalRNGSeed(2718)
    kg = cw.KnowledgeGradient(gp, 0, inner, w.discrete, points_to_sample=w.Xq, points_being_sampled=w.Xp,
                              num_mc_iterations=w.M, randomness=rnd)
    best = float(O.additional_mean(w.discrete).min())
    assert abs(kg._best_so_far - best) < 1e-12 * max(1.0, abs(best))
    m = (w.q + w.p)
    table = normal_draws(2718, ((w.M + 1) // 2) * m).reshape(-1, m)
    ro = O.kg(w.inner_gd, w.bounds, w.discrete, w.Xq, w.Xp, w.M, kg._best_so_far, table)
    val = kg.compute_knowledge_gradient()
    grad = kg.compute_grad_knowledge_gradient()
    assert grad.shape == (w.q, 3)
    assert abs(val - ro["kg"]) <= TOL["kg"] * abs(ro["kg"])
    assert np.abs(grad - ro["grad"]).max() <= TOL["grad_kg"] * max(np.abs(ro["grad"]).max(), abs(ro["kg"]))
    assert kg.compute_knowledge_gradient() == val  # common random numbers: the stream is rewound on every evaluation
    # evaluate_at_point_list == one compute_knowledge_gradient per point set
    pts = np.stack([w.Xq, w.Xq[::-1] * 0.9 + 0.05])
    vals = kg.evaluate_at_point_list(pts, max_num_threads=1)
    assert vals.shape == (2,) and vals[0] == val
    kg.set_current_point(pts[1])
    assert abs(kg.compute_objective_function() - vals[1]) <= 1e-12 * abs(vals[1])
    # posterior mean objective: value and gradient
    ps.set_current_point(w.query[3])
    assert abs(ps.compute_posterior_mean() + O.mean(w.query[3:4])[0]) < 1e-11
    assert rel(ps.compute_grad_posterior_mean().ravel(), -O.grad_mean(w.query[3:4])) < 1e-10


def test_expected_improvement_wrapper_flow():
    from cornell_moe_amd import GPP
    from cornell_moe_amd.api import normal_draws
    from oracle import orc
    cw, w, gp = _setup(seed=32, q=3, p=0)
    O = orc.OrcGP(1, w.alpha, w.lengths, w.X, w.y, w.noise, ())
    rnd = GPP.RandomnessSourceContainer(1)
    rnd.SetExplicitNormalRNGSeed(99)
    ei = cw.ExpectedImprovement(gp, points_to_sample=w.Xq, num_mc_iterations=w.M, randomness=rnd)
    assert ei._best_so_far == w.y[:, 0].min()
    # lift best_so_far so a healthy fraction of samples improve
    ei._best_so_far = float(np.median(w.y[:, 0]))
    table = normal_draws(99, w.M * w.q).reshape(w.M, w.q)
    eo, go = O.ei(w.Xq, None, w.M, ei._best_so_far, table)
    v = ei.compute_expected_improvement()
    g = ei.compute_grad_expected_improvement()
    assert g.shape == (3, 3)
    assert abs(v - eo) <= TOL["ei"] * max(abs(eo), 1e-3)
    assert np.abs(g - go).max() <= TOL["grad_ei"] * max(np.abs(go).max(), 1e-3)


def test_multistart_expected_improvement_optimization():
    """q,p-EI optimisation through the boundary: the C++ driver (moe_ei_multistart, Monte-Carlo evaluator at q = 2) against
    the numpy restatement on the same starts and table; the 1,0 case takes the analytic evaluator; point-list evaluation."""
    from cornell_moe_amd import GPP, api
    import ms_restatement as ms
    cw, w, gp = _setup(seed=33, n=80, d=3, q=2, p=1, M=400)
    dom = cw.TensorProductDomain([[0.0, 1.0]] * 3)
    rnd = GPP.RandomnessSourceContainer(1)
    rnd.SetExplicitNormalRNGSeed(5)
    rnd.SetExplicitUniformGeneratorSeed(9)
    ei = cw.ExpectedImprovement(gp, points_to_sample=w.Xq, points_being_sampled=w.Xp, num_mc_iterations=w.M, randomness=rnd)
    ei._best_so_far = float(np.median(w.y[:, 0]))
    outer_params = cw.GradientDescentParameters(24, 15, 2, 4, 0.7, 0.05, 0.2, 1e-8)
    opt = cw.GradientDescentOptimizer(dom, ei, outer_params, 10)
    status = {}
    best = cw.multistart_expected_improvement_optimization(opt, 24, 2, randomness=rnd, max_num_threads=1, status=status)
    assert best.shape == (2, 3) and best.min() >= 0.0 and best.max() <= 1.0
    assert status == {"gradient_descent_tensor_product_domain_found_update": True}
    # C++ driver vs numpy restatement, same starts, same normal table
    dev = gp._gaussian_process._dev
    bounds = np.tile([0.0, 1.0], 3)
    starts = np.stack([api.latin_hypercube(200 + r, bounds, 24) for r in range(2)], axis=1)
    table = rnd.normal_rng_vec[0].table(w.M * 3)
    gd = (24, 15, 2, 4, 0.7, 0.05, 0.2, 1e-8)
    cbest, cval, cfound = dev.ei_multistart(gd, bounds, starts, w.Xp, w.M, ei._best_so_far, table)
    value_fn = lambda x: dev.ei_batch(x, w.Xp, w.M, ei._best_so_far, table, want_grad=False)[0]  # noqa: E731
    grad_fn = lambda x: dev.ei_batch(x, w.Xp, w.M, ei._best_so_far, table)[1]  # noqa: E731
    nbest, nval, nfound = ms.multistart_best(value_fn, grad_fn, gd, bounds, starts, floor_value=-1.0)
    assert cfound and nfound and abs(cval - nval) <= 1e-10 * abs(nval) and np.abs(cbest - nbest).max() <= 1e-10
    assert cval >= value_fn(starts).max() * (1 - 1e-12)
    ei.set_current_point(cbest)
    assert abs(ei.compute_expected_improvement() - cval) <= 1e-12 * abs(cval)
    # 1,0-EI: analytic evaluator, no Monte Carlo (gpp_math.hpp:1703)
    ei1 = cw.ExpectedImprovement(gp, num_mc_iterations=w.M, randomness=rnd)
    ei1._best_so_far = ei._best_so_far
    opt1 = cw.GradientDescentOptimizer(dom, ei1, outer_params, 10)
    b1 = cw.multistart_expected_improvement_optimization(opt1, 24, 1, randomness=rnd, max_num_threads=1)
    assert b1.shape == (1, 3) and b1.min() >= 0.0 and b1.max() <= 1.0
    from oracle import orc
    O = orc.OrcGP(1, w.alpha, w.lengths, w.X, w.y, w.noise, ())  # the boundary always builds Matern-5/2
    pts = np.random.default_rng(1).uniform(size=(30, 1, 3))
    vals = ei1.evaluate_at_point_list(pts)
    want = np.array([O.ei_analytic(p.ravel(), ei._best_so_far, want_grad=False)[0] for p in pts])
    assert np.abs(vals - want).max() <= 1e-11 * max(want.max(), 1e-6)
    assert O.ei_analytic(b1.ravel(), ei._best_so_far, want_grad=False)[0] >= want.max() * 0.5  # a sane optimum
    # q = 2 point list: Monte Carlo
    pts2 = np.random.default_rng(2).uniform(size=(5, 2, 3))
    v2 = ei.evaluate_at_point_list(pts2, randomness=rnd, max_num_threads=1)
    for k in range(5):
        eo, _ = O.ei(pts2[k], w.Xp, w.M, ei._best_so_far, table.reshape(w.M, 3))
        assert abs(v2[k] - eo) <= TOL["ei"] * max(abs(eo), 1e-3)


def test_exceptions_cross_the_boundary_as_reference_classes():
    from cornell_moe_amd import GPP
    rng = np.random.default_rng(4)
    X = rng.uniform(size=(10, 2))
    X[3] = X[2]
    with pytest.raises(GPP.SingularMatrixException):  # gaussian_process_test.py:68 relies on this class
        GPP.GaussianProcess([1.0, [0.5, 0.5]], list(X.ravel()), list(np.zeros(10)), [0.0], [], 0, 2, 10)
    with pytest.raises(GPP.OptimalLearningException):
        GPP.GaussianProcess([1.0, [0.5, -0.5]], list(X.ravel()), list(np.zeros(10)), [0.1], [], 0, 2, 10)


def test_multistart_knowledge_gradient_optimization():
    """The outer optimiser: restarted gradient ascent over Latin-hypercube starts returns a point inside
    the domain whose KG is at least the best starting KG; deterministic for fixed seeds; status dict filled like the reference."""
    from cornell_moe_amd import GPP
    cw, w, gp = _setup(seed=41, n=60, d=2, q=2, p=0, P=6, M=200)
    ps = cw.PosteriorMean(gp, 0)
    dom = cw.TensorProductDomain([[0.0, 1.0]] * 2)
    inner = cw.GradientDescentOptimizer(dom, ps, cw.GradientDescentParameters(1, 6, 1, 3, 0.0, 1.0, 0.1, 1e-10))

    def run():
        rnd = GPP.RandomnessSourceContainer(1)
        rnd.SetExplicitNormalRNGSeed(7)
        rnd.SetExplicitUniformGeneratorSeed(11)
        kg = cw.KnowledgeGradient(gp, 0, inner, w.discrete, num_mc_iterations=w.M, randomness=rnd)
        outer = cw.GradientDescentOptimizer(dom, kg, cw.GradientDescentParameters(12, 8, 2, 4, 0.7, 0.3, 0.2, 1e-7), 10)
        status = {}
        best = cw.multistart_knowledge_gradient_optimization(outer, inner, 12, w.discrete, 2, w.discrete.shape[0],
                                                             randomness=rnd, max_num_threads=1, status=status)
        return kg, rnd, best, status

    kg, rnd, best, status = run()
    assert best.shape == (2, 2) and best.min() >= 0.0 and best.max() <= 1.0
    assert status == {"gradient_descent_tensor_product_domain_found_update": True}
    kg.set_current_point(best)
    v_best = kg.compute_knowledge_gradient()
    # (the C++ driver itself, moe_kg_multistart, is pinned to the reference's end points in tests/test_gpu_multistart.py)
    from cornell_moe_amd import api
    bounds = np.array([0.0, 1.0, 0.0, 1.0])
    starts = np.stack([api.latin_hypercube(100 + r, bounds, 12) for r in range(2)], axis=1)
    assert starts.shape == (12, 2, 2) and starts.min() >= 0.0 and starts.max() <= 1.0
    cells = np.floor(starts[:, 0, 1] * 12).astype(int)
    assert sorted(cells) == list(range(12))  # one point per slice of every edge
    assert np.isfinite(v_best)
    _, _, best2, _ = run()
    assert np.array_equal(best, best2)
    # recommendation step of the BO loop: posterior-mean optimisation from the best discrete point
    x0 = w.discrete[int(np.argmin(gp.compute_mean_of_additional_points(w.discrete)))]
    xs = np.array(GPP.posterior_mean_optimization(gp._gaussian_process, 0, inner.optimizer_parameters, [0.0, 1.0, 0.0, 1.0],
                                                  list(x0), {}))
    assert xs.shape == (2,) and xs.min() >= 0.0 and xs.max() <= 1.0
    assert gp.compute_mean_of_points(xs[None, :])[0] <= gp.compute_mean_of_points(x0[None, :])[0] + 1e-12


def test_log_likelihood_wrapper_flow():
    """cpp_wrappers.GaussianProcessLogLikelihood (log_likelihood.py:230-400 in the reference): value, hyper-parameter
    gradient and the list evaluator through the GPP stand-in, against the restatement; the gradient against central
    differences of the device value (no derivative observations: it IS the gradient of the value)."""
    import numpy as np
    import wrappers_mirror as cw
    from oracle import orc
    rng = np.random.default_rng(12)
    n, d = 80, 3
    X = rng.uniform(size=(n, d))
    y = np.sin(3 * X).sum(1) + 0.05 * rng.standard_normal(n)
    hd = cw.HistoricalData(dim=d, num_derivatives=0)
    hd.append_sample_points([cw.SamplePoint(X[i], [y[i]], 0.01) for i in range(n)])
    cov = cw.SquareExponential(np.array([1.3, 0.5, 0.7, 0.9]))
    ll = cw.GaussianProcessLogMarginalLikelihood(cov, hd, np.array([0.01]), [])
    assert ll.num_hyperparameters == 5 and ll.dim == d
    v = ll.compute_log_likelihood()
    g = ll.compute_grad_log_likelihood()
    theta = ll.hyperparameters
    vo = orc.log_likelihood(1, theta[0], theta[1:4], X, y[:, None], theta[4:], ())
    go = orc.log_likelihood_grad(1, theta[0], theta[1:4], X, y[:, None], theta[4:], ())
    assert abs(v - vo) <= 1e-10 * abs(vo) and np.abs(g - go).max() <= 1e-9 * np.abs(go).max()
    H = theta * np.linspace(0.8, 1.25, 7)[:, None]
    vals = cw.evaluate_log_likelihood_at_hyperparameter_list(ll, H)
    for k in range(7):
        ll.hyperparameters = H[k]
        assert abs(ll.compute_log_likelihood() - vals[k]) <= 1e-12 * abs(vals[k])
    ll.hyperparameters = theta
    for k in range(5):
        e = np.zeros(5)
        e[k] = 1e-6 * theta[k]
        fd = np.diff(cw.evaluate_log_likelihood_at_hyperparameter_list(ll, np.array([theta - e, theta + e])))[0] / (2 * e[k])
        assert abs(fd - g[k]) <= 1e-5 * max(1.0, abs(g[k]))



def test_run_cpp_tests_is_a_real_self_test():
    """GPP.run_cpp_tests() (r5): the reference returns its C++ suites' failure count (gpp_python_test.cpp:60-314); here the device
    self-test (cornell_moe_amd/selftest.py: gradient pings, EI consistency, linear algebra, random sources, optimisers).  0 failures on
    a healthy installation -- and a failing or raising check IS counted."""
    from cornell_moe_amd import GPP, selftest
    assert selftest.run(verbose=True) == 0
    assert GPP.run_cpp_tests() == 0

    def boom():
        raise RuntimeError("x")
    assert selftest.run(checks=(("always fails", lambda: False), ("raises", boom), ("passes", lambda: True))) == 2
