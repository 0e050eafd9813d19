"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against
 (a) the committed golden fixtures from the unmodified reference,
 (b) the plain-C oracle restatement (and oracle/_ref when its prebuilt .so travelled) on seeded inputs,
 (c) size-independent properties at BASELINE.json's full sizes (shard-sum invariance, antithetic structure,
     batch == single, re-evaluation determinism).
Tolerances are the stated FP64 ones in helpers.TOL / helpers.kg_tolerances."""
import numpy as np
import pytest

from helpers import TOL, kg_tolerances, rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from cornell_moe_amd import _lib, api as moe_api
    _lib.load()
    assert _lib.device_count() > 0, "no GPU visible"
    arch = _lib.C.create_string_buffer(64)
    _lib.load().moe_device_arch(0, arch, 64)
    assert b"gfx950" in arch.value, arch.value
    return moe_api


def _dev_gp(api, i):
    hyper = np.concatenate([[float(i["alpha"])], i["lengths"]])
    return api.DeviceGP(hyper, i["X"], i["y"], i["noise"], list(i["derivs"]), cov_type=int(i["cov_type"]))


def test_device_cholesky_known_answers(api, golden):
    _, la = golden
    for name in ("la_A", "la_B"):  # gpp_linear_algebra_test.cpp:237-262, exact small-integer factors
        L, Linv = api.debug_cholesky(la[name])
        assert np.array_equal(L, la[name + "_chol"])
        assert np.allclose(Linv @ la[name + "_chol"], np.eye(L.shape[0]), atol=1e-15)
    with pytest.raises(api.SingularMatrixException) as e:
        api.debug_cholesky(np.array([[4.0, 2.0], [2.0, 1.0]]))
    assert e.value.leading_minor_index == 2
    rng = np.random.default_rng(0)
    for n in (5, 64, 65, 200, 333):  # block-edge sizes of the NB=64 blocked factorisation
        B = rng.standard_normal((n, n))
        A = B @ B.T + n * np.eye(n)
        L, Linv = api.debug_cholesky(A)
        assert rel(L @ L.T, A) < 1e-13 and np.allclose(np.triu(L, 1), 0.0)
        assert rel(Linv @ L, np.eye(n)) < 1e-12
    # big enough that the recursive inversion's top GEMMs take the FP64-MFMA kernel (>= 512 output tiles of 64 x 64),
    # with a ragged edge (3111 = 48 * 64 + 39)
    n = 3111
    B = rng.standard_normal((n, 64))
    A = B @ B.T + np.diag(rng.uniform(1.0, 2.0, n))
    L, Linv = api.debug_cholesky(A)
    assert rel(L, np.linalg.cholesky(A)) < 1e-12
    assert rel(Linv @ L, np.eye(n)) < 1e-11 and np.allclose(np.triu(Linv, 1), 0.0)


def test_golden_gp_and_posterior(api, golden):
    cases, _ = golden
    for c in cases:
        gp = _dev_gp(api, c.inp)
        K, kiy, mean = gp.get_factor()
        assert rel(np.tril(K), c.out["K_chol"]) < TOL["K_chol"]
        assert rel(kiy, c.out["K_inv_y"]) < TOL["K_inv_y"]
        assert abs(mean - float(c.out["mean"])) < 1e-13
        pts = c.inp["query"]
        m4 = 4 * (1 + gp.g)
        assert rel(gp.mean(pts), c.out["q_mean"]) < TOL["q_mean"]
        assert rel(gp.additional_mean(pts), c.out["q_mean"]) < TOL["q_mean"]
        assert rel(gp.grad_mean(pts), c.out["q_grad_mean"]) < TOL["q_grad_mean"]
        assert rel(gp.variance(pts), c.out["q_var"]) < TOL["q_var"]
        assert rel(np.tril(gp.cholesky_variance(pts).reshape(m4, m4).T), c.out["q_chol_var"]) < TOL["q_chol_var"]
        assert rel(gp.grad_variance(pts, 2), c.out["q_grad_var"]) < TOL["q_grad_var"]
        assert rel(gp.grad_cholesky_variance(pts, 2), c.out["q_grad_chol_var"]) < TOL["q_grad_chol_var"]
        assert rel(gp.mix_covariance(pts, list(c.inp["derivs"])), c.out["q_mix_cov"]) < TOL["q_mix_cov"]
        v, g = gp.posterior_mean(pts[0])
        assert abs(v - float(c.out["post_mean"])) < TOL["post"] * max(1.0, abs(v)) and rel(g, c.out["post_grad"]) < TOL["post"]


def test_golden_ei(api, golden):
    cases, _ = golden
    for c in cases:
        i = c.inp
        gp = _dev_gp(api, i)
        Xp = i["Xp"] if int(i["p"]) > 0 else None
        ei, gei = gp.ei(i["Xq"], Xp, int(i["M"]), float(i["ei_best"]), i["ei_normals"])
        assert abs(ei - float(c.out["ei"])) <= TOL["ei"] * max(abs(float(c.out["ei"])), 1e-3)
        assert rel(gei, c.out["grad_ei"]) < TOL["grad_ei"]
        ei2, _ = gp.ei(i["Xq"], Xp, int(i["M"]), float(i["ei_best"]), i["ei_normals"], want_grad=False)
        assert ei2 == ei


def test_golden_analytic_ei_and_ei_multistart(api, golden):
    """a20 + its caller: analytic 1,0-EI (moe_ei_analytic_batch) and the EI multistart driver (moe_ei_multistart) against the
    reference's OnePotentialSampleExpectedImprovementEvaluator / ComputeOptimalPointsToSampleViaMultistartGradientDescent."""
    cases, _ = golden
    seen = 0
    for c in cases:
        i = c.inp
        G = _dev_gp(api, i)
        best = float(i["ei_best"])
        ei, grad = G.ei_analytic_batch(i["query"], best)
        ref_ei, ref_grad = c.out["ei_analytic"], c.out["grad_ei_analytic"]
        assert np.abs(ei - ref_ei).max() <= 1e-11 * max(np.abs(ref_ei).max(), 1e-6)
        assert np.abs(grad - ref_grad).max() <= 1e-9 * max(np.abs(ref_grad).max(), 1e-6)
        if "ms_starts" not in i:
            continue
        seen += 1
        d = int(i["d"])
        pt, val, found = G.ei_multistart(tuple(i["ms_gd"]), i["bounds"], i["ms_starts"].reshape(-1, 1, d), None, 1, best, None)
        assert found == bool(c.out["ms_found"])
        assert np.abs(pt.ravel() - c.out["ms_best_point"]).max() <= 1e-8
        assert abs(val - float(c.out["ms_best_ei"])) <= 1e-10 * abs(float(c.out["ms_best_ei"]))
        # value-only search (EvaluateEIAtPointList): the best start by analytic EI
        pt0, val0, found0 = G.ei_multistart(tuple(i["ms_gd"]), i["bounds"], i["ms_starts"].reshape(-1, 1, d), None, 1, best, None,
                                            gradient_ascent=False)
        all_ei = G.ei_analytic_batch(i["ms_starts"], best, want_grad=False)[0]
        assert found0 and val0 == all_ei.max() and np.array_equal(pt0.ravel(), i["ms_starts"][int(np.argmax(all_ei))])
        assert val >= val0
    assert seen == 3


def test_golden_log_likelihood(api, golden):
    """moe_ll_evaluate against the reference's LogMarginalLikelihoodEvaluator (three hyper-parameter sets per case, one
    handle per data set: the factorisation is redone in place per set); a singular K + noise gives -inf."""
    cases, _ = golden
    for c in cases:
        i = c.inp
        LL = api.LogLikelihood(i["X"], i["y"], list(i["derivs"]), cov_type=int(i["cov_type"]))
        sets = np.array([np.r_[float(i["alpha"]) * s, i["lengths"] * s, i["noise"] * s] for s in (1.0, 0.7, 1.6)])
        vals = LL.evaluate(sets)
        assert np.abs(vals - c.out["log_likelihood"]).max() <= 1e-10 * np.abs(c.out["log_likelihood"]).max()
        assert LL.evaluate(sets[1:2])[0] == vals[1]  # re-evaluation in place reproduces the value bit for bit
    rng = np.random.default_rng(0)
    X = rng.uniform(size=(70, 3))
    X[5] = X[4]                       # duplicate point, (almost) no noise: singular even with the 1e-6 jitter? no -- the
    y = rng.uniform(size=(70, 1))     # jitter keeps it factorable, like the reference; a NEGATIVE noise makes it singular
    LL = api.LogLikelihood(X, y)
    ok = LL.evaluate(np.array([[1.0, 0.5, 0.5, 0.5, 0.0]]))[0]
    assert np.isfinite(ok)
    assert LL.evaluate(np.array([[1.0, 0.5, 0.5, 0.5, -2.0]]))[0] == -np.inf
    from oracle import orc
    assert abs(ok - orc.log_likelihood(1, 1.0, [0.5, 0.5, 0.5], X, y, [0.0], ())) <= 1e-9 * abs(ok)


def test_golden_log_likelihood_grad(api):
    """moe_ll_grad (one fused N x N pass over alpha alpha^T - K^-1 on the device) against the reference's
    ComputeGradLogLikelihood fixtures and, at a larger size, against the restatement."""
    from helpers import load_golden_ll_grad
    from oracle import orc
    for c in load_golden_ll_grad():
        LL = api.LogLikelihood(c["X"], c["y"], list(c["derivs"]), cov_type=int(c["cov_type"]))
        theta = np.r_[float(c["alpha"]), c["lengths"], c["noise"]]
        g = LL.grad(theta)
        assert np.abs(g - c["grad"]).max() <= 1e-9 * np.abs(c["grad"]).max(), (c["X"].shape, np.abs(g - c["grad"]).max())
        assert abs(LL.evaluate(theta[None, :])[0] - float(c["value"])) <= 1e-10 * abs(float(c["value"]))
    rng = np.random.default_rng(8)
    n, d = 400, 8
    X = rng.uniform(size=(n, d))
    y = np.sin(3 * X).sum(1, keepdims=True) + 0.1 * rng.uniform(size=(n, 1))
    theta = np.r_[1.2, rng.uniform(0.4, 1.0, size=d), 0.02]
    g = api.LogLikelihood(X, y).grad(theta)
    o = orc.log_likelihood_grad(1, theta[0], theta[1:1 + d], X, y, theta[1 + d:], ())
    assert np.abs(g - o).max() <= 1e-9 * np.abs(o).max()
    with pytest.raises(api.InvalidValueException):
        api.LogLikelihood(X[:20], np.zeros((20, 2)), [0], cov_type=0).grad(np.r_[1.0, np.full(d, 0.5), 0.1, 0.1])
    with pytest.raises(api.SingularMatrixException):
        api.LogLikelihood(X[:20], np.zeros((20, 1))).grad(np.r_[1.0, np.full(d, 0.5), -2.0])


def test_log_likelihood_and_append_edges(api):
    """Sizes around the blocking of the factorisations (1, 2, 64, 65, 129 rows, with and without a derivative row per point),
    an empty hyper-parameter list, and a GP grown one observation at a time from a single point."""
    from oracle import orc
    rng = np.random.default_rng(0)
    for n, d, derivs in ((1, 1, ()), (2, 3, ()), (3, 2, (0,)), (65, 2, ()), (64, 2, (1,)), (129, 1, ())):
        g = len(derivs)
        X, y = rng.uniform(size=(n, d)), rng.uniform(size=(n, 1 + g))
        th = np.r_[1.1, np.full(d, 0.6), np.full(1 + g, 0.05)]
        LL = api.LogLikelihood(X, y, derivs)
        v = LL.evaluate(np.array([th, th * 1.1]))[0]
        gr = LL.grad(th)
        vo = orc.log_likelihood(1, th[0], th[1:1 + d], X, y, th[1 + d:], derivs)
        go = orc.log_likelihood_grad(1, th[0], th[1:1 + d], X, y, th[1 + d:], derivs)
        assert abs(v - vo) <= 1e-10 * max(1.0, abs(vo)) and np.abs(gr - go).max() <= 1e-8 * max(1.0, np.abs(go).max())
    assert LL.evaluate(np.zeros((0, th.size))).size == 0
    X, y = rng.uniform(size=(40, 2)), rng.uniform(size=(40, 1))
    a = api.DeviceGP([1.0, 0.5, 0.5], X[:1], y[:1], [0.01])
    for i in range(1, 40):
        a.add_points(X[i:i + 1], y[i:i + 1])
    b = api.DeviceGP([1.0, 0.5, 0.5], X, y, [0.01])
    q = rng.uniform(size=(5, 2))
    assert np.abs(a.mean(q) - b.mean(q)).max() <= 1e-11 and np.abs(a.variance(q) - b.variance(q)).max() <= 1e-11


def test_golden_kg(api, golden):
    cases, _ = golden
    ran = 0
    for c in cases:
        i = c.inp
        gp = _dev_gp(api, i)
        Xp = i["Xp"] if int(i["p"]) > 0 else None
        r = gp.kg(i["inner_gd"], i["bounds"], i["discrete"], i["Xq"], Xp, int(i["M"]), float(i["best_so_far"]),
                  i["kg_normals"], want_best_points=True)
        assert abs(r["kg"] - float(c.out["kg"])) <= TOL["kg"] * abs(float(c.out["kg"]))
        gtol, ptol = kg_tolerances(c)
        assert np.abs(r["grad"] - c.out["grad_kg"]).max() <= gtol
        assert np.abs(r["best_point"] - c.out["kg_best_point"]).max() <= ptol
        rv = gp.kg(i["inner_gd"], i["bounds"], i["discrete"], i["Xq"], Xp, int(i["M"]), float(i["best_so_far"]),
                   i["kg_normals"], want_grad=False)
        assert abs(rv["kg"] - float(c.out["kg_value_only"])) <= TOL["kg"] * abs(float(c.out["kg_value_only"]))
        ran += 1
    assert ran == len(cases) and any(len(c.inp["derivs"]) > 0 for c in cases)  # q-KG and d-KG cases


def test_seeded_vs_oracle(api):
    """Fresh seeded inputs at sizes the C oracle finishes in seconds (incl. C2's full shape), both kernels."""
    from cornell_moe_amd.workloads import make_workload
    from oracle import orc
    for (name, kw, cov) in [("C2", dict(), 1), (None, dict(seed=21, n=300, d=6, q=3, M=200, P=8, derivs=(), p=1), 0),
                            (None, dict(seed=22, n=129, d=2, q=1, M=100, P=3, derivs=(), p=0), 1)]:
        w = make_workload(name, **kw)
        O = orc.OrcGP(cov, w.alpha, w.lengths, w.X, w.y, w.noise, w.derivs)
        G = api.DeviceGP(w.hyperparameters, w.X, w.y, w.noise, w.derivs, cov_type=cov)
        pts = w.query[:16]
        assert rel(G.mean(pts), O.mean(pts)) < TOL["q_mean"]
        assert rel(G.variance(pts), O.var(pts)) < TOL["q_var"]
        best = float(O.additional_mean(w.discrete).min())
        Xp = w.Xp if w.p else None
        ro = O.kg(w.inner_gd, w.bounds, w.discrete, w.Xq, Xp, w.M, best, w.kg_normals)
        rg = G.kg(w.inner_gd, w.bounds, w.discrete, w.Xq, Xp, w.M, best, w.kg_normals, want_best_points=True)
        assert abs(rg["kg"] - ro["kg"]) <= TOL["kg"] * abs(ro["kg"])
        scale = max(np.abs(ro["grad"]).max(), abs(ro["kg"]))
        assert np.abs(rg["grad"] - ro["grad"]).max() <= TOL["grad_kg"] * scale
        mism = np.abs(rg["best_point"] - ro["best_point"]).max(axis=1) > 1e-8
        assert mism.mean() <= 0.002, "fraction of samples whose best point differs by > 1e-8: %g" % mism.mean()
        eb = float(np.min(w.y[:, 0])) + 0.5
        eo, go = O.ei(w.Xq, Xp, w.M, eb, w.ei_normals)
        eg, gg = G.ei(w.Xq, Xp, w.M, eb, w.ei_normals)
        assert abs(eo - eg) <= TOL["ei"] * max(abs(eo), 1e-3) and np.abs(gg - go).max() <= TOL["grad_ei"] * max(np.abs(go).max(), 1e-3)


def test_c1_posterior_plumbing(api):
    """BASELINE config C1: GP posterior mean/var on n=200, d=2 -- device vs oracle at 100 query points."""
    from cornell_moe_amd.workloads import make_workload
    from oracle import orc
    w = make_workload("C1")
    O = orc.OrcGP(1, w.alpha, w.lengths, w.X, w.y, w.noise, ())
    G = api.DeviceGP(w.hyperparameters, w.X, w.y, w.noise, ())
    assert rel(G.mean(w.query), O.mean(w.query)) < 1e-12
    for k in range(0, 100, 20):  # variance blocks of 20 points (q + p <= anything: pure posterior query)
        assert rel(G.variance(w.query[k:k + 20]), O.var(w.query[k:k + 20])) < 1e-11


def test_headline_shape_properties(api):
    """BASELINE config C3 at full size (n=1000, d=8, q=4, M=10000): properties that need no oracle run."""
    from cornell_moe_amd import dist as mdist
    from cornell_moe_amd.workloads import make_workload
    w = make_workload("C3", num_restarts=3)
    G = api.DeviceGP(w.hyperparameters, w.X, w.y, w.noise, ())
    best = float(G.additional_mean(w.discrete).min())
    args = (w.inner_gd, w.bounds, w.discrete, w.Xq, None, w.M, best, w.kg_normals)
    whole = G.kg(*args, want_best_points=True)
    again = G.kg(*args)
    assert whole["kg_sum"] == again["kg_sum"] and np.array_equal(whole["grad_sum"], again["grad_sum"])  # deterministic
    assert np.isfinite(whole["kg"]) and np.all(np.isfinite(whole["grad"]))
    bp = whole["best_point"]
    assert bp.shape == (w.M, 8) and bp.min() >= 0.0 and bp.max() <= 1.0  # LimitUpdate keeps every x* inside the domain
    # shard-sum invariance (what the multi-GPU all-reduce relies on): 1, 2, 4, 8 even-aligned MC shards
    for world in (2, 4, 8):
        ks, gs = 0.0, np.zeros_like(whole["grad_sum"])
        for r in range(world):
            first, count = mdist.shard_samples(w.M, r, world)
            part = G.kg(*args, first_sample=first, num_local=count)
            ks += part["kg_sum"]
            gs += part["grad_sum"]
        assert abs(ks - whole["kg_sum"]) <= 1e-12 * abs(whole["kg_sum"])
        assert np.abs(gs - whole["grad_sum"]).max() <= 1e-12 * max(np.abs(whole["grad_sum"]).max(), abs(whole["kg_sum"]))
    # batch == single evaluations
    b = G.kg_batch(w.inner_gd, w.bounds, w.discrete, w.Xq_restarts, None, w.M, best, w.kg_normals)
    assert b["kg_sum"][0] == whole["kg_sum"] and np.array_equal(b["grad_sum"][0], whole["grad_sum"])
    one = G.kg(w.inner_gd, w.bounds, w.discrete, w.Xq_restarts[2], None, w.M, best, w.kg_normals)
    assert b["kg_sum"][2] == one["kg_sum"]
    # value-only path agrees with the gradient path's value bit for bit (same MC kernel)
    v = G.kg(*args, want_grad=False)
    assert v["kg_sum"] == whole["kg_sum"]
    # counters: >= A + 1 value passes and >= 1 gradient pass per sample, <= max_num_steps gradient passes
    assert whole["grad_evals"] <= 6 * w.M and whole["grad_evals"] >= w.M and whole["mean_evals"] >= w.M


def test_dkg_shape_properties(api, monkeypatch):
    """BASELINE config C5 at full size (d-KG: n=2000, d=12, 3 observed derivatives, q=8, M=20000; N = 8000, m = 32): what can
    be said without an oracle run (the reference needs hours per evaluation here) -- determinism, MC shard-sum invariance,
    the two MC kernels against each other, every optimum inside the domain, pass counters in range."""
    from cornell_moe_amd import dist as mdist
    from cornell_moe_amd.workloads import make_workload
    w = make_workload("C5")
    G = api.DeviceGP(w.hyperparameters, w.X, w.y, w.noise, w.derivs)
    best = float(G.additional_mean(w.discrete).min())
    args = (w.inner_gd, w.bounds, w.discrete, w.Xq, None, w.M, best, w.kg_normals)
    whole = G.kg(*args, want_best_points=True)
    again = G.kg(*args)
    assert whole["kg_sum"] == again["kg_sum"] and np.array_equal(whole["grad_sum"], again["grad_sum"])
    assert np.isfinite(whole["kg"]) and np.all(np.isfinite(whole["grad"])) and whole["kg"] > 0.0
    bp = whole["best_point"]
    assert bp.shape == (w.M, 12) and bp.min() >= 0.0 and bp.max() <= 1.0
    assert w.M <= whole["grad_evals"] <= 6 * w.M and whole["mean_evals"] >= w.M
    ks, gs = 0.0, np.zeros_like(whole["grad_sum"])
    for r in range(4):
        first, count = mdist.shard_samples(w.M, r, 4)
        part = G.kg(*args, first_sample=first, num_local=count)
        ks += part["kg_sum"]
        gs += part["grad_sum"]
    gscale = max(np.abs(whole["grad_sum"]).max(), abs(whole["kg_sum"]))
    assert abs(ks - whole["kg_sum"]) <= 1e-12 * abs(whole["kg_sum"]) and np.abs(gs - whole["grad_sum"]).max() <= 1e-11 * gscale
    monkeypatch.setenv("MOE_KG_VARIANT", "0")  # wave-per-sample kernel, coordinates streamed from L2 at this size
    other = G.kg(*args, want_best_points=True)
    monkeypatch.delenv("MOE_KG_VARIANT")
    assert abs(other["kg"] - whole["kg"]) <= 1e-10 * abs(whole["kg"])
    assert np.abs(other["grad"] - whole["grad"]).max() <= 1e-9 * max(np.abs(whole["grad"]).max(), abs(whole["kg"]))
    assert (np.abs(other["best_point"] - bp).max(axis=1) > 1e-9).mean() <= 0.005
    assert other["grad_evals"] == whole["grad_evals"]


def test_workgroup_per_sample_kernel_matches_wave_per_sample(api, monkeypatch):
    """The two MC kernel variants (csrc/kg_mc.hpp: wave-per-sample with LDS tables, workgroup-per-sample with register
    tiles) implement the same algorithm with different summation trees: q-KG and d-KG results agree to rounding, and both
    match the oracle where it is affordable."""
    from cornell_moe_amd.workloads import make_workload
    from oracle import orc
    for kw, cov in ((dict(seed=61, n=700, d=5, q=3, M=400, P=8, derivs=(), p=1), 1),
                    (dict(seed=62, n=300, d=6, q=2, M=300, P=6, derivs=(1, 4), p=0), 1),
                    (dict(seed=63, n=150, d=3, q=2, M=200, P=5, derivs=(0, 1, 2), p=1), 0)):
        w = make_workload(**kw)
        G = api.DeviceGP(w.hyperparameters, w.X, w.y, w.noise, w.derivs, cov_type=cov)
        best = float(G.additional_mean(w.discrete).min())
        args = (w.inner_gd, w.bounds, w.discrete, w.Xq, w.Xp if w.p else None, w.M, best, w.kg_normals)
        monkeypatch.setenv("MOE_KG_VARIANT", "0")
        a = G.kg(*args, want_best_points=True)
        monkeypatch.setenv("MOE_KG_VARIANT", "1")
        b = G.kg(*args, want_best_points=True)
        # the workgroup-per-sample kernel takes beta / the discretised-set winner from a pre-pass and its weights from a
        # streamed table; computing them inside the kernel (the fall-back for tables that do not fit) gives the same bits
        monkeypatch.setenv("MOE_KG_PREP", "0")
        c = G.kg(*args, want_best_points=True)
        monkeypatch.delenv("MOE_KG_PREP")
        monkeypatch.setenv("MOE_KG_V_MAX_GB", "0")
        c2 = G.kg(*args, want_best_points=True)
        monkeypatch.delenv("MOE_KG_V_MAX_GB")
        for other in (c, c2):
            assert other["kg_sum"] == b["kg_sum"] and np.array_equal(other["grad_sum"], b["grad_sum"])
            assert np.array_equal(other["best_point"], b["best_point"])
        monkeypatch.delenv("MOE_KG_VARIANT")
        scale = max(np.abs(a["grad"]).max(), abs(a["kg"]))
        assert abs(a["kg"] - b["kg"]) <= 1e-10 * abs(a["kg"])
        assert np.abs(a["grad"] - b["grad"]).max() <= 1e-10 * scale
        assert (np.abs(a["best_point"] - b["best_point"]).max(axis=1) > 1e-9).mean() <= 0.005
        assert a["grad_evals"] == b["grad_evals"]
        if w.n <= 300:
            O = orc.OrcGP(cov, w.alpha, w.lengths, w.X, w.y, w.noise, w.derivs)
            ro = O.kg(w.inner_gd, w.bounds, w.discrete, w.Xq, w.Xp if w.p else None, w.M, best, w.kg_normals)
            so = max(np.abs(ro["grad"]).max(), abs(ro["kg"]))
            assert abs(b["kg"] - ro["kg"]) <= TOL["kg"] * abs(ro["kg"])
            assert np.abs(b["grad"] - ro["grad"]).max() <= TOL["grad_kg"] * so


def test_error_mapping(api):
    rng = np.random.default_rng(1)
    X = rng.uniform(size=(12, 2))
    X[5] = X[4]
    with pytest.raises(api.SingularMatrixException) as e:
        api.DeviceGP([1.0, 0.5, 0.5], X, np.zeros((12, 1)), [0.0])  # duplicate point, zero noise
    assert e.value.num_rows == 12 and 1 <= e.value.leading_minor_index <= 12
    X = rng.uniform(size=(12, 2))
    gp = api.DeviceGP([1.0, 0.5, 0.5], X, rng.uniform(size=(12, 1)), [0.0])
    with pytest.raises(api.SingularMatrixException):
        gp.cholesky_variance(np.vstack([X[0], X[0]]))  # duplicated query point duplicating a sampled point, 0 noise
    with pytest.raises(api.BoundsException):
        api.DeviceGP([1.0, -0.5, 0.5], X, np.zeros((12, 1)), [0.1])
    with pytest.raises(api.OptimalLearningException):
        gp.kg((1, 6, 1, 3, 0.0, 1.0, 0.1, 1e-10), [0, 1, 0, 1], X[:3], X[:2], None, 10, 0.0, np.zeros((5, 2)),
              first_sample=1, num_local=4)  # odd-aligned shard


def test_add_points_matches_fresh_build(api):
    rng = np.random.default_rng(2)
    X = rng.uniform(size=(50, 3))
    y = rng.uniform(size=(50, 1))
    a = api.DeviceGP([1.2, 0.5, 0.6, 0.7], X[:40], y[:40], [0.02])
    a.add_points(X[40:], y[40:])
    b = api.DeviceGP([1.2, 0.5, 0.6, 0.7], X, y, [0.02])
    q = rng.uniform(size=(7, 3))
    assert np.abs(a.mean(q) - b.mean(q)).max() <= 1e-12 and np.abs(a.variance(q) - b.variance(q)).max() <= 1e-12
    La, Lb = a.get_factor()[0], b.get_factor()[0]
    assert np.abs(La - Lb).max() <= 1e-12
    # more rows than the head-room behind the factorisation: the full rebuild, bit-identical to a fresh build
    X2, y2 = rng.uniform(size=(200, 3)), rng.uniform(size=(200, 1))
    a.add_points(X2, y2)
    c = api.DeviceGP([1.2, 0.5, 0.6, 0.7], np.vstack([X, X2]), np.vstack([y, y2]), [0.02])
    assert np.array_equal(a.mean(q), c.mean(q)) and np.array_equal(a.variance(q), c.variance(q))


@pytest.mark.parametrize("n0,adds,d,derivs", [(60, [4], 3, []), (300, [1, 3, 8, 2], 5, []), (1000, [4, 4, 4], 8, []),
                                              (90, [2, 5], 4, [0, 2]), (700, [16, 1], 6, [1])])
def test_add_points_rank_k_append(api, n0, adds, d, derivs):
    """A few new observations extend L and L^-1 by a block row (launch_cholesky_append) instead of refactorising
    (AddPointsToGP, gpp_math.cpp:1699-1737, refactorises): posterior, gradients and q-KG must match a GP built on all
    the points at once, and the oracle, at round-off level."""
    from oracle import orc
    rng = np.random.default_rng(n0 + d)
    g = len(derivs)
    n = n0 + sum(adds)
    X = rng.uniform(size=(n, d))
    y = np.sin(3 * X).sum(1, keepdims=True) + 0.1 * rng.uniform(size=(n, 1))
    if g:
        y = np.hstack([y] + [3 * np.cos(3 * X[:, [k]]) for k in derivs])
    hyper = [1.1] + list(0.4 + 0.05 * np.arange(d))
    noise = [0.01] * (1 + g)
    a = api.DeviceGP(hyper, X[:n0], y[:n0], noise, derivatives=derivs)
    at = n0
    for k in adds:
        a.add_points(X[at:at + k], y[at:at + k])
        at += k
    b = api.DeviceGP(hyper, X, y, noise, derivatives=derivs)
    #  the factor's strict upper triangle holds EXACT zeros after every append (and after the rebuild an append beyond the
    # head-room falls back to): the buffer is cleared once per shape and every writer -- covariance build, panel / update / diagonal
    # kernels, the append -- stays on or below the diagonal; trtri_levels and the K_chol download read it as an operand with explicit zeros
    La, Lb = a.get_factor()[0], b.get_factor()[0]   # [row][col]
    assert np.count_nonzero(np.triu(La, 1)) == 0 and np.all(np.isfinite(La))
    assert np.abs(np.tril(La) - np.tril(Lb)).max() <= 1e-10 * np.abs(Lb).max()
    q = rng.uniform(size=(9, d))
    for fa, fb in ((a.mean(q), b.mean(q)), (a.variance(q), b.variance(q)), (a.grad_mean(q), b.grad_mean(q)),
                   (a.cholesky_variance(q[:4]), b.cholesky_variance(q[:4])),
                   (a.grad_cholesky_variance(q[:3], 3), b.grad_cholesky_variance(q[:3], 3))):
        scale = max(1.0, float(np.abs(fb).max()))
        assert np.abs(np.asarray(fa) - np.asarray(fb)).max() <= 2e-10 * scale
    o = orc.OrcGP(1, hyper[0], hyper[1:], X, y, noise, derivs)
    assert np.abs(a.mean(q) - o.mean(q)).max() <= 1e-9
    assert np.abs(a.variance(q) - o.var(q)).max() <= 1e-9
    Z = rng.standard_normal((100, 3 * (1 + g)))
    bounds = [0.0, 1.0] * d
    gd = (1, 6, 1, 3, 0.0, 1.0, 0.1, 1e-10)
    ra = a.kg(gd, bounds, q[:5], q[5:8], None, 200, 0.1, Z)
    rb = b.kg(gd, bounds, q[:5], q[5:8], None, 200, 0.1, Z)
    assert abs(ra["kg"] - rb["kg"]) <= 1e-8 * max(1.0, abs(rb["kg"]))
    assert np.abs(ra["grad"] - rb["grad"]).max() <= 1e-7 * max(1.0, np.abs(rb["grad"]).max())


def test_add_points_append_singular(api):
    """A duplicate of a sampled point with zero noise makes the Schur complement singular: same error as a fresh build."""
    rng = np.random.default_rng(3)
    X = rng.uniform(size=(40, 2))
    gp = api.DeviceGP([1.0, 0.5, 0.5], X, rng.uniform(size=(40, 1)), [0.0])
    q = rng.uniform(size=(5, 2))
    before = (gp.mean(q), gp.variance(q))
    with pytest.raises(api.SingularMatrixException) as e:
        gp.add_points(X[[7]], [[0.3]])
    assert e.value.num_rows == 41
    # the failed append was rolled back: the handle still holds the 40 points and answers as before, on both sides
    assert gp.n == 40 and gp.N == 40
    assert np.abs(gp.mean(q) - before[0]).max() <= 1e-12 and np.abs(gp.variance(q) - before[1]).max() <= 1e-12
    gp.add_points(rng.uniform(size=(2, 2)), rng.uniform(size=(2, 1)))
    assert gp.n == 42


def test_point_per_thread_covariance_assembly_is_bit_identical(api, monkeypatch):
    """K(X, X) and K(X, x*) with derivative observations: the thread-per-point kernel (radial scalars once per pair, rows
    transposed through LDS) writes exactly the bits of the row-per-thread kernel (MOE_COV_FAST=0) -- sizes around the 64 / 256
    point blocking, derivative lists out of order, both covariance types."""
    rng = np.random.default_rng(31)
    for n, d, derivs, cov in ((70, 3, (2,), 1), (257, 5, (4, 0, 2), 0), (64, 12, tuple(range(12)), 1), (300, 4, (1, 3), 1)):
        g = len(derivs)
        X = rng.uniform(size=(n, d))
        y = rng.normal(size=(n, 1 + g))
        hyper = np.r_[1.2, rng.uniform(0.4, 1.1, size=d)]
        noise = np.full(1 + g, 0.05)
        pts = rng.uniform(size=(37, d))
        out = []
        for fast in ("1", "0"):
            monkeypatch.setenv("MOE_COV_FAST", fast)
            G = api.DeviceGP(hyper, X, y, noise, derivs, cov_type=cov)
            K, kiy, _ = G.get_factor()
            out.append((K, kiy, G.mix_covariance(pts, list(derivs)), G.mix_covariance(pts), G.mean(pts), G.variance(pts)))
        for a, b in zip(*out):
            assert np.array_equal(a, b)


def test_one_handle_from_several_threads(api):
    """A handle serialises its callers (per-handle mutex at the C ABI; ctypes drops the GIL for the duration of a call): four
    threads issuing KG evaluations, posterior queries and EI on ONE GP get exactly the results of the same calls made one after
    the other."""
    import threading
    from cornell_moe_amd.workloads import make_workload
    w = make_workload(seed=77, n=120, d=3, q=2, M=64, P=5, derivs=(), num_restarts=6)
    G = api.DeviceGP(w.hyperparameters, w.X, w.y, w.noise, ())
    best = float(G.additional_mean(w.discrete).min())

    def job(i):
        kg = G.kg(w.inner_gd, w.bounds, w.discrete, w.Xq_restarts[i], None, w.M, best, w.kg_normals)
        mu = G.mean(w.Xq_restarts[i])
        ei = G.ei(w.Xq_restarts[i], None, w.M, best, w.ei_normals)
        return kg["kg_sum"], kg["grad_sum"].copy(), mu.copy(), ei[0], ei[1].copy()

    serial = [job(i) for i in range(6)]
    out = [None] * 6
    errors = []

    def run(i):
        try:
            for _ in range(3):
                out[i] = job(i)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=run, args=(i,)) for i in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for a, b in zip(serial, out):
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3] == b[3] and np.array_equal(a[4], b[4])


def test_fastmath(api):
    """Device exp(-x) / sqrt(x) (csrc/fastmath.hpp) vs numpy: <= 2 ulp over the ranges the covariance loops produce."""
    rng = np.random.default_rng(5)
    x = np.concatenate([[0.0, 1e-300, 1e-200, 1e-30, 1e-16, 0.5, 1.0, 2.0, 700.0, 745.0, 800.0],
                        rng.uniform(0, 40, 200000), 10.0 ** rng.uniform(-12, 2.5, 200000)])
    e, r = api.debug_math(x)
    ref_e, ref_r = np.exp(-x), np.sqrt(x)
    big = ref_e > 1e-300
    ulp_e = np.abs(e[big] - ref_e[big]) / np.spacing(ref_e[big])
    assert ulp_e.max() <= 2.0, ulp_e.max()
    assert np.all(e[~big] <= 1e-299)
    pos = x >= 1e-290
    ulp_r = np.abs(r[pos] - ref_r[pos]) / np.spacing(ref_r[pos])
    assert ulp_r.max() <= 1.0, ulp_r.max()
    assert r[0] == 1e-150 and e[0] == 1.0


def test_two_level_cholesky(api, monkeypatch):
    """The large-N factorisation (outer blocks of 512 columns, LDS diagonal kernel, rank-512 trailing update on the matrix pipe):
    (1) forced at small sizes -- partial last blocks, derivative observations -- it gives the one-level path's factor, inverse
    application and posterior to round-off, and reports a singular matrix at the same minor; (2) at N = 2600 (its own
    territory) L L^T reproduces K to 1e-12 and L matches LAPACK's factor."""
    rng = np.random.default_rng(19)
    for n, d, derivs in ((300, 3, ()), (700, 4, ()), (260, 5, (0, 3))):
        X = rng.uniform(size=(n, d))
        g = len(derivs)
        y = rng.uniform(size=(n, 1 + g))
        hyper = [1.3] + list(0.3 + 0.1 * np.arange(d))
        noise = [0.02] * (1 + g)
        monkeypatch.delenv("MOE_CHOL_TWO_LEVEL_MIN", raising=False)
        a = api.DeviceGP(hyper, X, y, noise, derivatives=derivs)
        monkeypatch.setenv("MOE_CHOL_TWO_LEVEL_MIN", "64")
        b = api.DeviceGP(hyper, X, y, noise, derivatives=derivs)
        La, kiya, _ = a.get_factor()
        Lb, kiyb, _ = b.get_factor()
        assert np.abs(La - Lb).max() <= 1e-13 * np.abs(La).max()
        assert np.abs(kiya - kiyb).max() <= 1e-10 * np.abs(kiya).max()
        q = rng.uniform(size=(6, d))
        assert np.abs(a.mean(q) - b.mean(q)).max() <= 1e-11 and np.abs(a.variance(q) - b.variance(q)).max() <= 1e-11
    Xs = rng.uniform(size=(200, 2))
    Xs[150] = Xs[20]
    with pytest.raises(api.SingularMatrixException) as e:
        api.DeviceGP([1.0, 0.5, 0.5], Xs, np.zeros((200, 1)), [0.0])
    assert e.value.leading_minor_index == 151
    # r3: the fused step schedule (one launch per 64-column step: panel solve, update, the next diagonal block factored ahead
    # inside the kernel; in-kernel release / acquire flags between workgroups) against the round-2 schedule of three launches
    # per step: same factor and K^-1 y to round-off (the 64^3 products run on the matrix pipe in one and on FMA tiles in the
    # other), and bit for bit the same from run to run; sizes with a partial last block and a single outer block included
    monkeypatch.setenv("MOE_CHOL_TWO_LEVEL_MIN", "64")
    for n_ in (900, 1100, 130):
        X = rng.uniform(size=(n_, 3))
        y = rng.uniform(size=(n_, 1))
        facs = []
        for fs in ("0", "1", "1", "1"):
            monkeypatch.setenv("MOE_CHOL_FUSED_STEP", fs)
            L_, kiy_, _ = api.DeviceGP([1.1, 0.4, 0.5, 0.6], X, y, [0.03]).get_factor()
            facs.append((L_, kiy_))
        assert np.abs(facs[1][0] - facs[0][0]).max() <= 1e-13 * np.abs(facs[0][0]).max()
        assert np.abs(facs[1][1] - facs[0][1]).max() <= 1e-10 * np.abs(facs[0][1]).max()
        for L_, kiy_ in facs[2:]:
            assert np.array_equal(L_, facs[1][0]) and np.array_equal(kiy_, facs[1][1])
    monkeypatch.delenv("MOE_CHOL_FUSED_STEP")
    # a singular matrix is reported at the same minor by the in-kernel look-ahead (the duplicate sits in the second 64-block)
    Xd = rng.uniform(size=(300, 2))
    Xd[100] = Xd[7]
    with pytest.raises(api.SingularMatrixException) as e2:
        api.DeviceGP([1.0, 0.5, 0.5], Xd, np.zeros((300, 1)), [0.0])
    assert e2.value.leading_minor_index == 101
    X = rng.uniform(size=(900, 3))
    y = rng.uniform(size=(900, 1))
    th = np.array([[1.1, 0.4, 0.5, 0.6, 0.03], [0.9, 0.5, 0.5, 0.7, 0.05]])
    ll_two = api.LogLikelihood(X, y).evaluate(th)
    monkeypatch.setenv("MOE_CHOL_TWO_LEVEL_MIN", "1000000")
    ll_one = api.LogLikelihood(X, y).evaluate(th)
    assert np.abs(ll_two - ll_one).max() <= 1e-11 * np.abs(ll_one).max()
    monkeypatch.delenv("MOE_CHOL_TWO_LEVEL_MIN")
    n, d = 2600, 6
    X = rng.uniform(size=(n, d))
    y = np.sin(3 * X).sum(1, keepdims=True)
    G = api.DeviceGP([1.0] + [0.4] * d, X, y, [0.01])
    L, kiy, mean = G.get_factor()
    K = G.mix_covariance(X) + 0.01 * np.eye(n)
    assert np.abs(L @ L.T - K).max() <= 1e-12 * np.abs(K).max()
    Lref = np.linalg.cholesky(K)
    assert np.abs(np.tril(L) - Lref).max() <= 1e-11 * np.abs(Lref).max()
    assert np.abs(K @ kiy - (y[:, 0] - mean)).max() <= 1e-8 * np.abs(y).max()


def test_early_inverse_schedule(api, monkeypatch):
    """r5: the build's early-inverse schedule (kernels_linalg.hip: launch_cholesky_and_inverse -- the leading half's inverse and the top
    level's first product on a second stream while the trailing half is still being factored) against the level-wise schedule on one
    stream (MOE_CHOL_EARLY_INVERSE=0): same factor bit for bit (the factorisation's launches do not change), the same inverse
    application and posterior to round-off (a level's kernel is picked from its batch, which the schedule splits), bit-identical from
    run to run (no race between the two streams), K (K^-1 y) = y - mean.  Sizes: the split on the first eligible outer-block boundary
    with a short trailing part, a trailing part almost as long as the leading one, derivative observations."""
    rng = np.random.default_rng(29)
    for n, d, derivs in ((2400, 3, ()), (3900, 4, ()), (1100, 5, (0, 2))):
        X = rng.uniform(size=(n, d))
        g = len(derivs)
        y = rng.uniform(size=(n, 1 + g))
        hyper = [1.2] + list(0.25 + 0.1 * np.arange(d))
        noise = [0.02] * (1 + g)
        monkeypatch.setenv("MOE_CHOL_EARLY_INVERSE", "0")
        a = api.DeviceGP(hyper, X, y, noise, derivatives=derivs)
        monkeypatch.setenv("MOE_CHOL_EARLY_INVERSE", "1")
        b = api.DeviceGP(hyper, X, y, noise, derivatives=derivs)
        b2 = api.DeviceGP(hyper, X, y, noise, derivatives=derivs)
        La, kiya, _ = a.get_factor()
        Lb, kiyb, mean = b.get_factor()
        Lb2, kiyb2, _ = b2.get_factor()
        assert np.array_equal(La, Lb)
        assert np.array_equal(Lb, Lb2) and np.array_equal(kiyb, kiyb2)
        assert np.abs(kiya - kiyb).max() <= 1e-10 * np.abs(kiya).max()
        q = rng.uniform(size=(5, d))
        assert np.abs(a.mean(q) - b.mean(q)).max() <= 1e-11 and np.abs(a.variance(q) - b.variance(q)).max() <= 1e-11
        assert np.array_equal(b.variance(q), b2.variance(q))
        if g == 0:
            K = b.mix_covariance(X) + noise[0] * np.eye(n)
            assert np.abs(K @ kiyb - (y[:, 0] - mean)).max() <= 1e-8 * np.abs(y).max()
        monkeypatch.delenv("MOE_CHOL_EARLY_INVERSE")


def test_gemm128_products_of_the_build(api, monkeypatch):
    """r4: the 128-tile matrix-pipe kernel (csrc/gemm128.hpp) behind the inverse factor's two products per level and the rank-512
    update, forced at sizes where the 64-tile kernel is the default -- row / column counts that are no multiple of 128, K ranges that
    are no multiple of 16, a level whose last node is cut off by the matrix edge: same factor, K^-1 y, posterior and log-likelihood
    gradient as the 64-tile kernels to round-off, L^-1 L = I, and bit-identical from run to run."""
    rng = np.random.default_rng(23)
    for n, d, derivs in ((1300, 3, ()), (650, 4, (1,)), (2100, 2, ())):
        X = rng.uniform(size=(n, d))
        g = len(derivs)
        y = rng.uniform(size=(n, 1 + g))
        hyper = [1.2] + list(0.25 + 0.1 * np.arange(d))
        noise = [0.02] * (1 + g)
        monkeypatch.setenv("MOE_GEMM128", "0")
        a = api.DeviceGP(hyper, X, y, noise, derivatives=derivs)
        monkeypatch.setenv("MOE_GEMM128", "2")
        b = api.DeviceGP(hyper, X, y, noise, derivatives=derivs)
        b2 = api.DeviceGP(hyper, X, y, noise, derivatives=derivs)
        La, kiya, _ = a.get_factor()
        Lb, kiyb, _ = b.get_factor()
        Lb2, kiyb2, _ = b2.get_factor()
        assert np.array_equal(Lb, Lb2) and np.array_equal(kiyb, kiyb2)
        assert np.abs(La - Lb).max() <= 1e-13 * np.abs(La).max()
        assert np.abs(kiya - kiyb).max() <= 1e-10 * np.abs(kiya).max()
        q = rng.uniform(size=(5, d))
        assert np.abs(a.mean(q) - b.mean(q)).max() <= 1e-11 and np.abs(a.variance(q) - b.variance(q)).max() <= 1e-11
        if g == 0:  # K^-1 y = L^-T (L^-1 yc) must solve the system (both products of the inverse factor enter)
            K = b.mix_covariance(X) + noise[0] * np.eye(n)
            assert np.abs(K @ kiyb - (y[:, 0] - b.get_factor()[2])).max() <= 1e-8 * np.abs(y).max()
        monkeypatch.delenv("MOE_GEMM128")


def test_ei_device_algebra_matches_host_algebra(api, golden, monkeypatch):
    """r3: the u x u algebra of an EI evaluation on the device (ei.hip: ei_state_kernel -- one sync per call) against the host
    algebra it replaces (MOE_EI_DEVICE_ALGEBRA=0: two syncs): same formulas, the device's exp / sqrt instead of libm's, so the
    results agree to rounding, not bit for bit; shapes with points being sampled, derivative observations on the GP, q up to 3,
    value-only calls; a duplicated point is reported singular with the same leading minor by both."""
    cases, _ = golden
    worst = 0.0
    for c in cases:
        i = c.inp
        gp = _dev_gp(api, i)
        Xp = i["Xp"] if int(i["p"]) > 0 else None
        res = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("MOE_EI_DEVICE_ALGEBRA", mode)
            res[mode] = gp.ei(i["Xq"], Xp, int(i["M"]), float(i["ei_best"]), i["ei_normals"])
            res[mode + "v"] = gp.ei(i["Xq"], Xp, int(i["M"]), float(i["ei_best"]), i["ei_normals"], want_grad=False)[0]
        e1, g1 = res["1"]
        e0, g0 = res["0"]
        sc = max(abs(e0), float(np.abs(g0).max()), 1e-3)
        worst = max(worst, abs(e1 - e0) / sc, float(np.abs(g1 - g0).max()) / sc)
        assert abs(e1 - e0) <= 1e-12 * sc and np.abs(g1 - g0).max() <= 1e-11 * sc
        assert res["1v"] == e1 and res["0v"] == e0
        # against the reference itself, on the device path
        assert abs(e1 - float(c.out["ei"])) <= TOL["ei"] * max(abs(float(c.out["ei"])), 1e-3)
        assert rel(g1, c.out["grad_ei"]) < TOL["grad_ei"]
    print("device vs host EI algebra: worst relative difference %.2e" % worst)
    i = cases[0].inp
    gp = _dev_gp(api, i)
    Xq = np.vstack([i["Xq"][:1], i["Xq"][:1]])  # the same point twice: singular variance matrix... with the 1e-6 jitter it is
    for mode in ("1", "0"):                      # NOT singular (gpp_math.cpp:2000-2002) -- both paths must agree on that too
        monkeypatch.setenv("MOE_EI_DEVICE_ALGEBRA", mode)
        e, g = gp.ei(Xq, None, 16, float(i["ei_best"]), np.random.default_rng(5).standard_normal((16, 2)))
        assert np.isfinite(e) and np.all(np.isfinite(g))
    monkeypatch.delenv("MOE_EI_DEVICE_ALGEBRA")


def test_variance_of_hundreds_of_points_on_the_device():
    """r5: compute_variance_of_points / compute_cholesky_variance_of_points for query sets of hundreds of points take
    the device algebra (gp.hip: variance_on_device -- Kss by the covariance kernel, the Gram subtraction, the GP's blocked Cholesky)
    instead of the host's O(m^2 d) covariance calls and O(m^3) scalar factorisation: k = 512 points (and 150 points with two observed
    derivatives: m = 450) against the unmodified reference / the restatement; the small-set host path and the device path agree
    where they meet; the log likelihood still sees its own workspace afterwards; a duplicate point is reported singular."""
    import time
    from cornell_moe_amd import api
    from cornell_moe_amd.workloads import make_workload
    from oracle import orc
    from helpers import reference_checker
    for n, d, derivs, k in ((400, 4, (), 512), (300, 3, (0, 2), 150)):
        w = make_workload(seed=500 + k, n=n, d=d, q=2, M=8, P=4, derivs=derivs)
        noise = np.maximum(w.noise, 1e-3)
        G = api.DeviceGP(w.hyperparameters, w.X, w.y, noise, derivs)
        R = reference_checker(1, w.alpha, w.lengths, w.X, w.y, noise, derivs) or orc.OrcGP(1, w.alpha, w.lengths, w.X, w.y, noise, derivs)
        pts = np.random.default_rng(k).uniform(0.02, 0.98, size=(k, d))
        m = k * (1 + len(derivs))
        ll0 = G.log_likelihood() if hasattr(G, "log_likelihood") else None
        t0 = time.perf_counter()
        var = G.variance(pts).reshape(m, m)
        t_var = time.perf_counter() - t0
        t0 = time.perf_counter()
        chol = G.cholesky_variance(pts).reshape(m, m)
        t_chol = time.perf_counter() - t0
        vr = np.asarray(R.var(pts)).reshape(m, m)
        scale = np.abs(vr).max()
        assert np.abs(var - vr).max() <= 1e-10 * scale, np.abs(var - vr).max() / scale
        L = np.tril(chol.T)                       # (flat col-major: row-major view is the transpose)
        assert np.abs(L @ L.T - vr).max() <= 1e-9 * scale
        assert np.array_equal(np.triu(chol.T, 1), np.triu(var.T, 1))   # the variance's entries stay above the diagonal (host path's layout)
        cr = np.asarray(R.chol_var(pts)).reshape(m, m)
        assert np.abs(L - np.tril(cr.T)).max() <= 1e-8 * np.sqrt(scale)
        if ll0 is not None:
            assert G.log_likelihood() == ll0
        # the device path at its thresholds (r6: the variance assembled on the device from 16 rows, its factor by the host's column sweep up
        # to 223 rows and by the blocked device kernels from 224 -- gp.hpp: device_variance_min_m) against the path just below them: the
        # same points, one or two fewer
        for thr in (16, 224):
            small = pts[: max(1, (thr - 1) // (1 + len(derivs)))]
            big = pts[: thr // (1 + len(derivs)) + 1]
            ms, mb = len(small) * (1 + len(derivs)), len(big) * (1 + len(derivs))
            assert ms < thr <= mb
            vs, vb = G.variance(small), G.variance(big)
            assert np.abs(vb.reshape(mb, mb)[:ms, :ms] - vs.reshape(ms, ms)).max() <= 1e-12 * scale
            cs, cb = G.cholesky_variance(small), G.cholesky_variance(big)   # (the leading block of a factor is the factor of the leading block)
            assert np.abs(np.tril(cb.reshape(mb, mb).T)[:ms, :ms] - np.tril(cs.reshape(ms, ms).T)).max() <= 1e-8 * np.sqrt(scale)   # (the bound this test holds the factor to against the reference)
        print("variance of %d points (m = %d): %.1f ms, Cholesky variance %.1f ms" % (k, m, 1e3 * t_var, 1e3 * t_chol))
        dup = np.vstack([pts[:40], pts[:1]])
        Gz = api.DeviceGP(w.hyperparameters, w.X, w.y, np.zeros_like(noise), derivs)
        with pytest.raises(api.SingularMatrixException):
            Gz.cholesky_variance(np.vstack([dup, w.X[:1]]))


def test_hyperparameter_optimisers_against_reference():
    """r5 (SURVEY 8b 'next'): moe_ll_multistart -- MultistartGradientDescentHyperparameterOptimization
    (gpp_model_selection.hpp:1063-1103) from explicit linear-space guesses, every start's restarted gradient ascent stepped together on the
    device -- against the unmodified reference (tests/golden/ref_ll_multistart.npz, tools/make_golden.py --ll-multistart: the reference's
    own function body minus its Latin-hypercube draw): the maximum-likelihood hyper-parameters to 1e-6 relative, its log likelihood to
    1e-9, the found flag (contractive step sizes; under the reference's own test settings only the likelihood can be pinned, see below); and the boundary functions GPP.multistart_hyperparameter_optimization / restarted_hyperparameter_optimization."""
    import os
    from cornell_moe_amd import GPP, api
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_ll_multistart.npz"))
    for k in range(int(z["num"])):
        X, y, derivs = z["s%d_X" % k], z["s%d_y" % k], [int(v) for v in z["s%d_derivs" % k]]
        gd, dom, guesses = tuple(z["s%d_gd" % k]), z["s%d_domain_log10" % k], z["s%d_guesses" % k]
        LL = api.LogLikelihood(X, y, derivs)
        v0 = LL.evaluate(guesses)
        assert np.abs(v0 - z["s%d_initial_values" % k]).max() <= 1e-9 * np.abs(z["s%d_initial_values" % k]).max()
        best, val, found = LL.multistart(gd, dom, guesses)
        rb, rv = z["s%d_best" % k], float(z["s%d_best_value" % k])
        assert found == bool(z["s%d_found" % k])
        if int(z["s%d_contractive" % k]):
            assert abs(val - rv) <= 1e-9 * max(abs(rv), 1.0), (k, val, rv)
            assert np.abs(best / rb - 1.0).max() <= 1e-6, (k, best, rb)
        else:
            # the reference's own test settings (pre_mult 0.5, max_relative_change 0.02): every step is cut to 2 % of the distance to the
            # wall, only the SIGN of each gradient component enters, and near an optimum it flips on rounding -- a 1e-12 relative
            # perturbation of the reference's own gradient moves ITS end point by 1.6 % (tools/make_golden.py): the likelihood reached is
            # pinned, not the point
            assert abs(val - rv) <= 1e-3 * abs(rv), (k, val, rv)
        lo, hi = 10.0 ** dom[:, 0], 10.0 ** dom[:, 1]
        assert np.all(best >= lo) and np.all(best <= hi)
        # restarted optimiser from the first guess: ends where the reference's single-start run ends (case 1 IS a single-start run)
        end = LL.ascend(gd, dom, guesses[0])
        if guesses.shape[0] == 1 and int(z["s%d_contractive" % k]):
            assert np.abs(end / rb - 1.0).max() <= 1e-6
        assert LL.evaluate(end[None, :])[0] >= v0[0]
    # the boundary: names / argument order of gpp_python_model_selection.cpp:428-474
    k = 0
    X, y = z["s0_X"], z["s0_y"]
    n, d = X.shape

    class Opt(object):
        objective_type = GPP.LogLikelihoodTypes.log_marginal_likelihood
        optimizer_type = GPP.OptimizerTypes.gradient_descent
        num_random_samples = 64
        optimizer_parameters = GPP.GradientDescentParameters(4, 40, 2, 0, 0.5, 0.5, 0.02, 1.0e-7)

    rnd = GPP.RandomnessSourceContainer(1)
    rnd.SetExplicitUniformGeneratorSeed(7)
    status = {}
    dom = z["s0_domain_log10"]
    got = GPP.multistart_hyperparameter_optimization(Opt(), list(dom.ravel()), list(X.ravel()), list(y.ravel()), d, n, [1.0, [0.5] * d],
                                                     [0.1], [], 0, 4, rnd, status)
    assert len(got) == 1 + d + 1 and "log_marginal_likelihood_gradient_descent_found_update" in status
    ll_got = GPP.compute_log_likelihood(list(X.ravel()), list(y.ravel()), d, n, GPP.LogLikelihoodTypes.log_marginal_likelihood,
                                        [got[0], list(got[1:1 + d])], [], 0, [got[1 + d]])
    start = GPP.compute_log_likelihood(list(X.ravel()), list(y.ravel()), d, n, GPP.LogLikelihoodTypes.log_marginal_likelihood,
                                       [1.0, [0.5] * d], [], 0, [0.1])
    assert ll_got > start
    Opt.optimizer_type = GPP.OptimizerTypes.null
    st2 = {}
    lhc = GPP.multistart_hyperparameter_optimization(Opt(), list(dom.ravel()), list(X.ravel()), list(y.ravel()), d, n, [1.0, [0.5] * d],
                                                     [0.1], [], 0, 4, rnd, st2)
    assert st2["log_marginal_likelihood_lhc_found_update"] is True and len(lhc) == 1 + d + 1
    Opt.optimizer_type = GPP.OptimizerTypes.gradient_descent
    end = GPP.restarted_hyperparameter_optimization(Opt(), list(dom.ravel()), list(X.ravel()), list(y.ravel()), d, n, [1.0, [0.5] * d],
                                                    [0.1], [], 0, {})
    assert len(end) == 1 + d + 1 and np.all(np.isfinite(end))


def test_gradient_query_endpoints_on_the_device():
    """r6 (VERDICT r5 missing 3): compute_grad_variance_of_points / compute_grad_cholesky_variance_of_points finish on the device
    (csrc/query_grad.hip: one workgroup per differentiated point -- the variance gradient's assembly, Var and its factor in the
    reference's order, Smith's forward-mode derivative) instead of on the host over a downloaded Gram matrix.  Against the unmodified
    reference (the restatement where it is not built): a q-EI-sized state, a state with derivative observations differentiated in all
    of its points (m = 120), fewer differentiated points than points, both covariances; a duplicated point is reported singular."""
    from cornell_moe_amd import api
    from cornell_moe_amd.workloads import make_workload
    from oracle import orc
    from helpers import reference_checker
    for cov, n, d, derivs, k, nd in ((1, 120, 4, (), 6, 6), (1, 90, 3, (0, 2), 40, 40), (0, 70, 5, (1,), 9, 4), (1, 60, 8, (), 1, 1)):
        w = make_workload(seed=900 + k, n=n, d=d, q=2, M=8, P=4, derivs=derivs)
        noise = np.maximum(w.noise, 1e-3)
        G = api.DeviceGP(w.hyperparameters, w.X, w.y, noise, derivs, cov_type=cov)
        R = reference_checker(cov, w.alpha, w.lengths, w.X, w.y, noise, derivs) or orc.OrcGP(cov, w.alpha, w.lengths, w.X, w.y, noise, derivs)
        pts = np.random.default_rng(77 + k).uniform(0.02, 0.98, size=(k, d))
        m = k * (1 + len(derivs))
        gv = G.grad_variance(pts, nd)
        gr = np.asarray(R.grad_var(pts, nd))
        assert gv.size == nd * d * m * m == gr.size
        assert rel(gv.ravel(), gr.ravel()) < TOL["q_grad_var"], (cov, k, rel(gv.ravel(), gr.ravel()))
        gc = G.grad_cholesky_variance(pts, nd)
        cr = np.asarray(R.grad_chol_var(pts, nd))
        assert rel(gc.ravel(), cr.ravel()) < TOL["q_grad_chol_var"], (cov, k, rel(gc.ravel(), cr.ravel()))
        # entries below the block are zeros, exactly (gpp_math.cpp:1403-1411)
        blk = gc.reshape(nd, m, m, d)   # [p][col block i][row within the column][dd]: rows > i are zero
        for i in range(m - 1):
            assert not blk[:, i, i + 1:, :].any()
    # a duplicated query point with zero noise: Var is singular -- the reference's SingularMatrixException, not garbage
    w = make_workload(seed=950, n=40, d=3, q=2, M=8, P=4, derivs=())
    G = api.DeviceGP(w.hyperparameters, w.X, w.y, np.zeros(1), ())
    pts = np.vstack([w.query[:2], w.query[:1]])
    with pytest.raises(api.SingularMatrixException):
        G.grad_cholesky_variance(pts, 2)
    assert np.all(np.isfinite(G.grad_variance(pts, 2)))   # (the variance's gradient itself needs no factor)
