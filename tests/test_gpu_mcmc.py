"""GPU parity of the MCMC-averaged evaluators through the C ABI: moe_kg_mcmc_batch, moe_ei_mcmc_batch,
moe_kg_mcmc_multistart, moe_ei_mcmc_multistart against the golden fixtures generated from the reference's
GaussianProcessMCMC / KnowledgeGradientMCMCEvaluator / ExpectedImprovementMCMCEvaluator, the GP-index shard identity, and the
wrapper flow (cpp_wrappers mirror -> GPP stand-in -> C ABI)."""
import numpy as np
import pytest

from helpers import TOL, load_golden_mcmc, rel

pytestmark = pytest.mark.gpu


def _dev(c, members=None):
    from cornell_moe_amd import api
    i = c.inp
    return api.DeviceGPMCMC(i["hypers"], i["noises"], i["X"], i["y"], list(i["derivs"]), members=members)


def test_golden_kg_and_ei_mcmc():
    for c in load_golden_mcmc():
        i = c.inp
        d, f, M = int(i["d"]), int(i["num_fidelity"]), int(i["M"])
        G = _dev(c)
        bounds = i["bounds"][:2 * (d - f)]
        kg, grad = G.kg_batch(tuple(i["inner_gd"]), bounds, i["discrete"], i["Xq"][None], i["Xp"], M, i["kg_best"], i["kg_normals"],
                              num_fidelity=f)
        ref_kg, ref_grad = float(c.out["kg"]), c.out["grad_kg"]
        assert abs(kg[0] - ref_kg) <= TOL["kg"] * abs(ref_kg)
        assert np.abs(grad[0] - ref_grad).max() <= TOL["grad_kg"] * max(np.abs(ref_grad).max(), abs(ref_kg))
        kv, none = G.kg_batch(tuple(i["inner_gd"]), bounds, i["discrete"], i["Xq"][None], i["Xp"], M, i["kg_best"], i["kg_normals"],
                              want_grad=False, num_fidelity=f)
        assert none is None and abs(kv[0] - float(c.out["kg_value_only"])) <= TOL["kg"] * abs(kv[0])
        ei, gei = G.ei_batch(i["Xq"][None], i["Xp"], M, i["ei_best"], i["ei_normals"])
        assert abs(ei[0] - float(c.out["ei"])) <= TOL["ei"] * abs(ei[0])
        assert rel(gei[0], c.out["grad_ei"]) < TOL["grad_ei"]
        # GP-index shards (the multi-GPU axis): plain sums over disjoint member sets add up to the whole, then one finalize
        parts = [_dev(c, members=m) for m in ([0, 2], [1])]
        sums = [P.kg_batch(tuple(i["inner_gd"]), bounds, i["discrete"], i["Xq"][None], i["Xp"], M, i["kg_best"], i["kg_normals"],
                           num_fidelity=f, finalize=False) for P in parts]
        k2, g2 = G.kg_finalize(sums[0][0] + sums[1][0], sums[0][1] + sums[1][1], i["Xq"][None], num_fidelity=f)
        assert abs(k2[0] - kg[0]) <= 1e-13 * abs(kg[0]) and np.abs(g2 - grad).max() <= 1e-13 * max(np.abs(grad).max(), 1.0)


def test_golden_ei_mcmc_multistart_analytic():
    for c in load_golden_mcmc():
        i = c.inp
        d = int(i["d"])
        G = _dev(c)
        starts = i["ms_starts"].reshape(-1, 1, d)
        pt, val, found = G.ei_multistart(tuple(i["ms_gd"]), i["bounds"], starts, None, 1, i["ei_best"], None)
        assert found == bool(c.out["ms_found"])
        assert np.abs(pt.ravel() - c.out["ms_best_point"]).max() <= 1e-8
        assert abs(val - float(c.out["ms_best_ei"])) <= 1e-10 * abs(val)
        all_ei = G.ei_batch(starts, None, 1, i["ei_best"], None, want_grad=False, analytic=True)[0]
        pt0, val0, found0 = G.ei_multistart(tuple(i["ms_gd"]), i["bounds"], starts, None, 1, i["ei_best"], None, gradient_ascent=False)
        assert found0 and val0 == all_ei.max() and np.array_equal(pt0.ravel(), i["ms_starts"][int(np.argmax(all_ei))])


def test_mcmc_wrapper_flow():
    """The reference's Python call sequence for the MCMC objects (knowledge_gradient_mcmc.py / expected_improvement_mcmc.py)
    on the mirror classes, checked against the oracle restatement fed the same normal tables."""
    from cornell_moe_amd import GPP
    import wrappers_mirror as cw
    from cornell_moe_amd.api import normal_draws
    from cornell_moe_amd.workloads import make_workload
    from oracle import orc
    w = make_workload(seed=51, n=50, d=3, q=2, M=200, P=5, derivs=(), p=1)
    rng = np.random.default_rng(3)
    nm = 4
    hypers = np.c_[rng.uniform(0.8, 1.4, nm), rng.uniform(0.4, 0.9, size=(nm, 3))]
    noises = rng.uniform(0.01, 0.05, size=(nm, 1))
    hd = cw.HistoricalData(dim=3, num_derivatives=0)
    hd.append_sample_points([cw.SamplePoint(w.X[k], w.y[k], 0.01) for k in range(w.n)])
    gpm = cw.GaussianProcessMCMC(hypers, noises, hd, [])
    models = gpm.member_models()
    assert len(models) == nm and gpm.dim == 3 and gpm.num_sampled == w.n
    O = orc.OrcGPMCMC(hypers, noises, w.X, w.y, ())
    dom = cw.TensorProductDomain([[0.0, 1.0]] * 3)
    ps = cw.PosteriorMeanMCMC(models, 0)
    inner = cw.GradientDescentOptimizer(dom, ps, cw.GradientDescentParameters(1, 6, 1, 3, 0.0, 1.0, 0.1, 1e-10))
    rnd = GPP.RandomnessSourceContainer(1)
    rnd.SetExplicitNormalRNGSeed(77)
    rnd.SetExplicitUniformGeneratorSeed(5)
    discrete_list = [rng.uniform(size=(5, 3)) for _ in range(nm)]
    kg = cw.KnowledgeGradientMCMC(gpm, models, 0, inner, discrete_list, 2, points_to_sample=w.Xq, points_being_sampled=w.Xp,
                                  num_mc_iterations=w.M, randomness=rnd)
    want_best = [float(np.min(O.gps[k].additional_mean(discrete_list[k]))) for k in range(nm)]
    assert np.allclose(kg._best_so_far_list, want_best, rtol=1e-11, atol=1e-12)
    table = normal_draws(77, ((w.M + 1) // 2) * 3).reshape(-1, 3)
    ko, go = O.kg(w.inner_gd, w.bounds, np.array(discrete_list), w.Xq, w.Xp, w.M, kg._best_so_far_list, table)
    v = kg.compute_knowledge_gradient_mcmc()
    g = kg.compute_grad_knowledge_gradient_mcmc()
    assert abs(v - ko) <= TOL["kg"] * abs(ko) and np.abs(g - go).max() <= TOL["grad_kg"] * max(np.abs(go).max(), abs(ko))
    pts = rng.uniform(0.1, 0.9, size=(3, 2, 3))
    vals = kg.evaluate_at_point_list(pts, randomness=rnd, max_num_threads=1)
    for k in range(3):
        ok, _ = O.kg(w.inner_gd, w.bounds, np.array(discrete_list), pts[k], w.Xp, w.M, kg._best_so_far_list, table, want_grad=False)
        assert abs(vals[k] - ok) <= TOL["kg"] * abs(ok)
    # posterior mean averaged over the ensemble
    ps.set_current_point(w.query[2])
    assert abs(ps.compute_posterior_mean_mcmc() + np.mean([gp.mean(w.query[2:3])[0] for gp in O.gps])) < 1e-11
    # outer optimisation: runs, stays in the domain, fills the status like the reference, never ends below its best start
    outer = cw.GradientDescentOptimizer(dom, kg, cw.GradientDescentParameters(10, 4, 1, 4, 0.7, 0.05, 0.2, 1e-7), 10)
    status = {}
    best = cw.multistart_knowledge_gradient_mcmc_optimization(outer, inner, 10, discrete_list, 2, 5, randomness=rnd,
                                                              max_num_threads=1, status=status)
    assert best.shape == (2, 3) and best.min() >= 0.0 and best.max() <= 1.0
    assert status == {"gradient_descent_tensor_product_domain_found_update": True}
    # EI twin
    ei = cw.ExpectedImprovementMCMC(gpm, 2, points_to_sample=w.Xq, points_being_sampled=w.Xp, num_mc_iterations=w.M, randomness=rnd)
    ei._best_so_far_list = nm * [float(np.median(w.y[:, 0]))]
    etable = normal_draws(77, w.M * 3).reshape(w.M, 3)
    eo, geo = O.ei(w.Xq, w.Xp, w.M, ei._best_so_far_list, etable)
    assert abs(ei.compute_expected_improvement() - eo) <= TOL["ei"] * abs(eo)
    assert rel(ei.compute_grad_expected_improvement(), geo) < TOL["grad_ei"]
    ei1 = cw.ExpectedImprovementMCMC(gpm, 1, num_mc_iterations=w.M, randomness=rnd)
    ei1._best_so_far_list = ei._best_so_far_list
    pts1 = rng.uniform(size=(12, 1, 3))
    got = ei1.evaluate_at_point_list(pts1, max_num_threads=1)
    want = np.array([O.ei_analytic(p.ravel(), ei._best_so_far_list, want_grad=False)[0] for p in pts1])
    assert np.abs(got - want).max() <= 1e-11 * max(want.max(), 1e-6)
    opt1 = cw.GradientDescentOptimizer(dom, ei1, cw.GradientDescentParameters(24, 15, 2, 4, 0.7, 0.05, 0.2, 1e-8), 10)
    b1 = cw.multistart_expected_improvement_mcmc_optimization(opt1, 24, 1, randomness=rnd, max_num_threads=1)
    assert b1.shape == (1, 3) and b1.min() >= 0.0 and b1.max() <= 1.0
    assert O.ei_analytic(b1.ravel(), ei._best_so_far_list, want_grad=False)[0] >= want.max() * 0.5


def test_bo_loop_example_runs():
    """examples/bo_loop.py: two iterations of the full loop (hyper-parameter sampling on moe_ll_evaluate, ensemble build,
    KG-MCMC multistart, point addition) end to end on the device path; the suggestions stay in the domain and the best
    observed Branin value does not get worse."""
    import importlib.util
    import os
    import sys
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "bo_loop.py")
    spec = importlib.util.spec_from_file_location("bo_loop", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    argv = sys.argv
    sys.argv = ["bo_loop.py", "2", "2", "4"]
    try:
        y = mod.main()
    finally:
        sys.argv = argv
    assert y.shape == (8 + 2 * 2,) and np.all(np.isfinite(y)) and y.min() <= y[:8].min()
