"""The drop-in boundary: cornell_moe_amd.GPP must export what the reference's cpp_wrappers bind from
``moe.build.GPP`` for the hot path, with the same positional signatures; cornell_moe_amd.cpp_wrappers mirrors the reference's
wrapper classes.  CPU-only checks here (no compute); the GPU flow is tests/test_gpu_boundary.py."""
import inspect
import os
import re

import numpy as np
import pytest

REF_WRAPPERS = "/root/reference/moe/optimal_learning/python/cpp_wrappers"

# (GPP name, number of positional arguments) of the boost::python exports on the hot path, with the reference line that
# defines each wrapper's parameter list.
HOT_PATH_EXPORTS = [
    ("compute_posterior_mean", 3),                      # gpp_python_knowledge_gradient.cpp:44-46
    ("compute_grad_posterior_mean", 3),                 # :60-62
    ("compute_knowledge_gradient", 13),                 # :76-84
    ("compute_grad_knowledge_gradient", 13),            # :115-123
    ("multistart_knowledge_gradient_optimization", 15),  # :243-252
    ("evaluate_KG_at_point_list", 15),                  # :344-354
    ("posterior_mean_optimization", 6),                 # :315-320
    ("compute_expected_improvement", 9),                # gpp_python_expected_improvement.cpp:44-50
    ("compute_grad_expected_improvement", 9),           # :77-83
    ("evaluate_EI_at_point_list", 11),                  # :401-408 (EvaluateEIAtPointListWrapper)
    ("multistart_expected_improvement_optimization", 13),  # :221-229 (MultistartExpectedImprovementOptimizationWrapper)
    # MCMC-averaged evaluators: gpp_python_knowledge_gradient_mcmc.cpp, gpp_python_expected_improvement_mcmc.cpp
    ("compute_knowledge_gradient_mcmc", 13), ("compute_grad_knowledge_gradient_mcmc", 13),
    ("multistart_knowledge_gradient_mcmc_optimization", 15), ("evaluate_KG_mcmc_at_point_list", 15),
    ("compute_expected_improvement_mcmc", 8), ("compute_grad_expected_improvement_mcmc", 8),
    ("multistart_expected_improvement_mcmc_optimization", 11), ("evaluate_EI_mcmc_at_point_list", 11),
    # log marginal likelihood: gpp_python_model_selection.cpp:43-51, 281-291
    ("compute_log_likelihood", 9), ("evaluate_log_likelihood_at_hyperparameter_list", 13),
    ("compute_hyperparameter_grad_log_likelihood", 9),  # gpp_python_model_selection.cpp:88-96
]
GP_METHODS = [  # gpp_python_gaussian_process.cpp:294-465 (self + listed arguments)
    ("compute_mean_of_points", 2), ("compute_mean_of_additional_points", 2), ("compute_grad_mean_of_points", 2),
    ("compute_variance_of_points", 2), ("compute_cholesky_variance_of_points", 2), ("compute_grad_variance_of_points", 3),
    ("compute_grad_cholesky_variance_of_points", 3), ("add_sampled_points", 3), ("sample_point_from_gp", 1),
    ("sample_global_optima", 3), ("set_explicit_seed", 1), ("set_randomized_seed", 1), ("reset_to_most_recent_seed", 0),
    ("print_historical_data", 0),
]


def _positional(fn):
    return [p for p in inspect.signature(fn).parameters.values() if p.default is inspect.Parameter.empty]


def test_gpp_exports_and_arity():
    from cornell_moe_amd import GPP
    for name, nargs in HOT_PATH_EXPORTS:
        assert len(_positional(getattr(GPP, name))) == nargs, name
    for name, nargs in GP_METHODS:
        assert len(_positional(getattr(GPP.GaussianProcess, name))) == nargs + 1, name
    assert len(_positional(GPP.GaussianProcess.__init__)) == 1 + 8  # make_gaussian_process, :42-47
    assert len(_positional(GPP.GaussianProcessMCMC.__init__)) == 1 + 9  # make_gaussian_process_mcmc (..._mcmc.cpp:45-50)
    for cls in ("OptimalLearningException", "BoundsException", "InvalidValueException", "SingularMatrixException"):
        assert issubclass(getattr(GPP, cls), Exception)
    assert issubclass(GPP.SingularMatrixException, GPP.OptimalLearningException)
    for enum, members in (("OptimizerTypes", ("null", "gradient_descent", "newton")), ("DomainTypes", ("tensor_product", "simplex")),
                          ("LogLikelihoodTypes", ("log_marginal_likelihood", "leave_one_out_log_likelihood"))):
        for m in members:
            assert hasattr(getattr(GPP, enum), m)
    gd = GPP.GradientDescentParameters(1, 6, 1, 3, 0.0, 1.0, 0.1, 1e-10)
    gd.max_num_steps = 7  # fields are read/write like the boost struct (gpp_python_common.cpp:243-279)
    assert gd._as_tuple() == (1, 7, 1, 3, 0.0, 1.0, 0.1, 1e-10)


@pytest.mark.skipif(not os.path.isdir(REF_WRAPPERS), reason="reference tree not present (GPU box)")
def test_every_hot_path_binding_of_the_reference_wrappers_exists():
    """Static scan of the reference's own wrapper files: each C_GP.<name> they use on the hot path resolves in our module."""
    from cornell_moe_amd import GPP
    in_scope = ["knowledge_gradient.py", "expected_improvement.py", "gaussian_process.py", "domain.py", "optimization.py",
                "covariance.py", "knowledge_gradient_mcmc.py", "expected_improvement_mcmc.py"]
    out_of_scope = set()
    missing = []
    for fn in in_scope:
        src = open(os.path.join(REF_WRAPPERS, fn)).read()
        for name in sorted(set(re.findall(r"C_GP\.([A-Za-z_]+)", src))):
            if name not in out_of_scope and not hasattr(GPP, name):
                missing.append((fn, name))
    assert not missing, missing


@pytest.mark.skipif(not os.path.isdir(REF_WRAPPERS), reason="reference tree not present (GPU box)")
def test_call_sites_of_the_reference_wrappers_fit_our_signatures():
    """Every ``C_GP.<name>(...)`` call in the reference's hot-path wrapper files passes an argument count our function of that
    name accepts (ast scan of the reference sources; nothing is imported from them)."""
    import ast
    from cornell_moe_amd import GPP
    checked = 0
    for fn in ("knowledge_gradient.py", "expected_improvement.py", "gaussian_process.py", "knowledge_gradient_mcmc.py",
               "expected_improvement_mcmc.py"):
        tree = ast.parse(open(os.path.join(REF_WRAPPERS, fn)).read())
        for node in ast.walk(tree):
            if not (isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute)
                    and isinstance(node.func.value, ast.Name) and node.func.value.id == "C_GP"):
                continue
            obj = getattr(GPP, node.func.attr)
            params = list(inspect.signature(obj.__init__ if inspect.isclass(obj) else obj).parameters.values())
            if inspect.isclass(obj):
                params = params[1:]
            required = [p for p in params if p.default is inspect.Parameter.empty]
            n = len(node.args) + len(node.keywords)
            assert len(required) <= n <= len(params), (fn, node.func.attr, n, len(required), len(params))
            checked += 1
    assert checked >= 30


def test_randomness_source_semantics():
    from cornell_moe_amd import GPP
    r = GPP.RandomnessSourceContainer(3)
    assert r.num_normal_rng == 3
    assert [s.last_seed for s in r.normal_rng_vec] == [314, 315, 316]  # kNormalDefaultSeed + i
    r.SetExplicitNormalRNGSeed(1000)
    assert [s.last_seed for s in r.normal_rng_vec] == [1000, 1001, 1002]  # gpp_python_common.cpp:152-156
    assert r.SetNormalRNGSeedPythonList([5, 6, 7], [1, 0, 1]) is True
    assert [s.last_seed for s in r.normal_rng_vec] == [5, 1001, 7]
    assert r.SetNormalRNGSeedPythonList([5], [1]) is False
    t1 = r.normal_rng_vec[0].table(10).copy()
    t2 = r.normal_rng_vec[0].table(6)
    assert np.array_equal(t1[:6], t2)  # every evaluation replays the stream from the last seed
    a = GPP.RandomnessSourceContainer(1)
    a.SetRandomizedNormalRNGSeed(0)
    b = GPP.RandomnessSourceContainer(1)
    b.SetRandomizedNormalRNGSeed(0)
    assert a.normal_rng_vec[0].last_seed != b.normal_rng_vec[0].last_seed


def test_normal_draws_are_standard_normal():
    from cornell_moe_amd.api import normal_draws
    z = normal_draws(314, 200001)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1.0) < 0.01 and abs((z ** 3).mean()) < 0.03
    assert np.array_equal(z[:1000], normal_draws(314, 1000))  # prefix-stable stream
    assert not np.array_equal(z[:1000], normal_draws(315, 1000))


def test_wrapper_mirror_containers():
    import wrappers_mirror as cw
    hd = cw.HistoricalData(2, 1)
    hd.append_sample_points([cw.SamplePoint([0.1, 0.2], [1.0, 0.5], 0.01), ([0.3, 0.4], [2.0, -0.5], 0.01)])
    assert hd.num_sampled == 2 and hd.points_sampled.shape == (2, 2) and hd.points_sampled_value.shape == (2, 2)
    with pytest.raises(ValueError):
        cw.SamplePoint([0.0], [0.0], -1.0)
    dom = cw.TensorProductDomain([[0.0, 1.0], [-1.0, 2.0]])
    assert cw.cppify(dom.domain_bounds) == [0.0, 1.0, -1.0, 2.0] and dom.check_point_inside([0.5, 0.0])
    opt = cw.GradientDescentOptimizer(dom, None, cw.GradientDescentParameters(1, 6, 1, 3, 0.0, 1.0, 0.1, 1e-10))
    assert opt.optimizer_parameters.optimizer_parameters.max_num_steps == 6
    assert int(opt.optimizer_parameters.domain_type) == 0 and int(opt.optimizer_parameters.optimizer_type) == 1
    cov = cw.SquareExponential([1.0, 0.5, 0.6])
    assert cw.cppify_hyperparameters(cov.hyperparameters) == [1.0, [0.5, 0.6]]


def test_multistart_host_pieces():
    """LimitUpdate (gpp_domain.cpp:64-105) vectorised == a scalar restatement; Latin hypercube has one point per slice."""
    import ms_restatement as ms
    rng = np.random.default_rng(3)
    bounds = np.array([[0.0, 1.0], [-2.0, 3.0], [5.0, 5.5]])

    def scalar(lo, hi, mrc, x, s):
        dist = min(x - lo, hi - x)
        if abs(s) > mrc * dist:
            s = np.copysign(mrc * dist, s)
        nxt = x + s
        if nxt < lo:
            s = 0.5 * (lo - x) if x + 0.5 * s < lo else 0.5 * s
        elif nxt > hi:
            s = 0.5 * (hi - x) if x + 0.5 * s > hi else 0.5 * s
        return s

    x = bounds[:, 0] + rng.uniform(size=(50, 2, 3)) * (bounds[:, 1] - bounds[:, 0])
    x[0, 0] = bounds[:, 0]  # on the walls
    x[1, 1] = bounds[:, 1]
    step = rng.normal(scale=2.0, size=x.shape)
    got = ms.limit_update(bounds, 0.7, x, step)
    for idx in np.ndindex(x.shape):
        lo, hi = bounds[idx[-1]]
        assert got[idx] == scalar(lo, hi, 0.7, x[idx], step[idx])
    assert np.all(x + got >= bounds[:, 0] - 1e-15) and np.all(x + got <= bounds[:, 1] + 1e-15)
    u = np.random.RandomState(1)
    pts = ms.repeated_domain_starts(bounds.ravel(), 16, 3, lambda n: u.uniform(size=n))
    assert pts.shape == (16, 3, 3)
    for r in range(3):
        for k in range(3):
            cells = np.floor((pts[:, r, k] - bounds[k, 0]) / ((bounds[k, 1] - bounds[k, 0]) / 16)).astype(int)
            assert sorted(cells) == list(range(16))
