"""INTEGRATION.md route A, executed: the reference's OWN wrapper modules (`moe.optimal_learning.python.cpp_wrappers.*`, imported
from /root/reference unchanged) running on `cornell_moe_amd.GPP` aliased as `moe.build.GPP`.

No GPU in this container, so the flow is checked in two layers:
  * the real `GaussianProcess` wrapper is constructed and its first `C_GP.*` call is followed down to the C ABI -- a recorder on
    `moe_gp_create` sees the arrays and sizes the wrapper marshalled, the library (no device) answers MOE_ERR_RUNTIME and the
    wrapper's caller gets the reference's `OptimalLearningException` class;
  * with the device object behind `GPP.GaussianProcess` replaced by a recorder, the real `PosteriorMean`, `KnowledgeGradient`
    and `ExpectedImprovement` wrappers are driven through every `C_GP.*` call of SURVEY
    Appendix C and the shapes / values that reach `api.DeviceGP` are asserted.
Skipped where /root/reference does not exist (the GPU box: tests/test_gpu_boundary.py runs the same sequence there on a mirror)."""
import os
import sys
import types

import numpy as np
import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "moe", "optimal_learning", "python", "cpp_wrappers")),
                                reason="the reference tree is not present")


@pytest.fixture(scope="module")
def ref():
    """Install route A's alias and import the reference's wrapper modules."""
    here = os.path.dirname(os.path.abspath(__file__))
    added = [p for p in (os.path.join(here, "shims"), REF) if p not in sys.path]
    try:
        import future.utils  # noqa: F401  (the real package wins where it is installed)
        added = [p for p in added if not p.endswith("shims")]
    except ImportError:
        pass
    sys.path[:0] = added
    saved = {k: sys.modules.get(k) for k in ("moe", "moe.build", "moe.build.GPP")}
    import cornell_moe_amd.GPP as GPP
    import moe
    build = types.ModuleType("moe.build")
    build.GPP = GPP
    sys.modules["moe.build"] = build
    sys.modules["moe.build.GPP"] = GPP
    moe.build = build
    ns = types.SimpleNamespace(GPP=GPP)
    # (log_likelihood.py imports emcee at module level -- not in this image; its three C_GP entry points are covered by
    #  tests/test_boundary.py's call-site scan and tests/test_gpu_boundary.py)
    from moe.optimal_learning.python.cpp_wrappers import (covariance, domain, expected_improvement, gaussian_process,
                                                          knowledge_gradient, optimization)
    from moe.optimal_learning.python import data_containers
    ns.covariance, ns.domain, ns.expected_improvement, ns.gaussian_process = covariance, domain, expected_improvement, gaussian_process
    ns.knowledge_gradient, ns.optimization, ns.data_containers = knowledge_gradient, optimization, data_containers
    assert knowledge_gradient.C_GP is GPP and gaussian_process.C_GP is GPP
    yield ns
    for p in added:
        sys.path.remove(p)
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v
    for k in [k for k in sys.modules if k.startswith("moe.optimal_learning") or k in ("future", "future.utils", "past", "past.utils")]:
        if "shims" in (getattr(sys.modules[k], "__file__", "") or "") or k.startswith("moe.optimal_learning"):
            sys.modules.pop(k, None)


def _data(ref, n=7, d=3, derivs=(0, 2), seed=5):
    rng = np.random.default_rng(seed)
    g = len(derivs)
    X = rng.uniform(-1.0, 1.0, size=(n, d))
    Y = rng.normal(size=(n, 1 + g))
    hd = ref.data_containers.HistoricalData(dim=d, num_derivatives=g)
    hd.append_historical_data(X, Y, np.full((n, 1 + g), 0.1))
    hyper = np.concatenate([[2.5], rng.uniform(0.5, 2.0, size=d)])
    return X, Y, hd, hyper, list(derivs), np.full(1 + g, 0.1)


def test_real_gaussian_process_wrapper_reaches_the_c_abi(ref):
    from cornell_moe_amd import _lib
    X, Y, hd, hyper, derivs, noise = _data(ref)
    L = _lib.load()
    seen = {}
    orig = L.moe_gp_create

    def recorder(*args):
        seen["args"] = args
        return orig(*args)

    L.moe_gp_create = recorder
    try:
        cov = ref.covariance.SquareExponential(hyper)
        if _lib.device_count() > 0:
            gp = ref.gaussian_process.GaussianProcess(cov, noise, hd, derivs)
            assert gp.dim == 3 and gp.num_sampled == 7
        else:
            with pytest.raises(ref.GPP.OptimalLearningException):
                ref.gaussian_process.GaussianProcess(cov, noise, hd, derivs)
    finally:
        L.moe_gp_create = orig
    # what the reference's wrapper marshalled (gaussian_process.py:77-86) arrived at the C ABI (moe_hip.h: moe_gp_create(hyper,
    # cov_type, X, y, noise, derivatives, g, dim, n, device, &handle, &err)): sizes and the arrays' contents
    import ctypes as C
    hp, cov_type, Xp, yp, noisep, dvp, g, dim, n = seen["args"][:9]
    assert (g, dim, n) == (2, 3, 7)
    dbl = lambda ptr, count: np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(count,)).copy()  # noqa: E731
    assert np.array_equal(dbl(hp, 1 + dim), hyper)
    assert np.array_equal(dbl(Xp, n * dim).reshape(n, dim), X) and np.array_equal(dbl(yp, n * (1 + g)).reshape(n, 1 + g), Y)
    assert np.array_equal(dbl(noisep, 1 + g), noise)
    assert list(np.ctypeslib.as_array(C.cast(dvp, C.POINTER(C.c_int)), shape=(g,))) == derivs


class _FakeDev(object):
    """Stands where api.DeviceGP stands behind GPP.GaussianProcess: records what GPP.py hands to the device layer."""

    def __init__(self, hyperparameters, X, y, noise, derivatives, cov_type=None, device=0):
        self.n, self.d = X.shape
        self.derivatives = tuple(int(v) for v in derivatives)
        self.g = len(self.derivatives)
        self.hyper, self.X, self.y, self.noise = np.array(hyperparameters), np.array(X), np.array(y), np.array(noise)
        self.calls = []

    def additional_mean(self, pts):
        self.calls.append(("additional_mean", np.array(pts)))
        return np.arange(len(pts), dtype=float) - 2.0

    def mean(self, pts):
        self.calls.append(("mean", np.array(pts)))
        return np.zeros(len(pts) * (1 + self.g))

    def variance(self, pts):
        m = len(pts) * (1 + self.g)
        self.calls.append(("variance", np.array(pts)))
        return np.eye(m).ravel()

    def posterior_mean(self, point, num_fidelity, want_grad=True):
        self.calls.append(("posterior_mean", np.array(point), num_fidelity, want_grad))
        return 0.25, (np.zeros(self.d - num_fidelity) if want_grad else None)   # (value, gradient): api.DeviceGP.posterior_mean

    def kg(self, inner, bounds, discrete, Xq, Xp, M, best, normals, want_grad=True, num_fidelity=0, **kw):
        self.calls.append(("kg", inner, np.array(bounds), np.array(discrete), np.array(Xq), None if Xp is None else np.array(Xp), M, best,
                           np.array(normals), want_grad, num_fidelity))
        q = np.array(Xq).reshape(-1, self.d).shape[0]
        return {"kg": 0.5, "grad": np.full((q, self.d), 0.125)}

    def ei(self, Xq, Xp, M, best, normals, want_grad=True, want_value=True):
        self.calls.append(("ei", np.array(Xq), None if Xp is None else np.array(Xp), M, best, np.array(normals), want_grad))
        q = np.array(Xq).reshape(-1, self.d).shape[0]
        return 0.75, np.full((q, self.d), 0.5)


def test_real_wrappers_marshal_every_call_of_appendix_c(ref, monkeypatch):
    from cornell_moe_amd import api
    monkeypatch.setattr(api, "DeviceGP", _FakeDev)
    X, Y, hd, hyper, derivs, noise = _data(ref)
    d, g = 3, 2
    gp = ref.gaussian_process.GaussianProcess(ref.covariance.SquareExponential(hyper), noise, hd, derivs)
    dev = gp._gaussian_process._dev
    assert isinstance(dev, _FakeDev) and np.allclose(dev.X, X) and np.allclose(dev.y, Y) and dev.derivatives == (0, 2)
    assert np.allclose(dev.hyper, hyper) and np.allclose(dev.noise, noise)
    assert gp.dim == d and gp.num_sampled == 7

    bounds = [ref.domain.ClosedInterval(-1.0, 1.0)] * d if hasattr(ref.domain, "ClosedInterval") else None
    if bounds is None:
        from moe.optimal_learning.python.geometry_utils import ClosedInterval
        bounds = [ClosedInterval(-1.0, 1.0)] * d
    dom = ref.domain.TensorProductDomain(bounds)
    gdp = ref.optimization.GradientDescentParameters(1, 6, 1, 3, 0.0, 1.0, 0.1, 1.0e-10)
    ps = ref.knowledge_gradient.PosteriorMean(gp, 0)
    inner = ref.optimization.GradientDescentOptimizer(dom, ps, gdp)
    rng = np.random.default_rng(11)
    discrete = rng.uniform(-1.0, 1.0, size=(5, d))
    Xq = rng.uniform(-1.0, 1.0, size=(2, d))
    Xp = rng.uniform(-1.0, 1.0, size=(1, d))
    randomness = ref.GPP.RandomnessSourceContainer(1)
    randomness.SetExplicitNormalRNGSeed(314)
    M = 12
    kg = ref.knowledge_gradient.KnowledgeGradient(gp, 0, inner, discrete, points_to_sample=Xq, points_being_sampled=Xp,
                                                  num_mc_iterations=M, randomness=randomness)
    # best_so_far = min mu(discrete) (knowledge_gradient.py:366-368) from the recorder's additional_mean
    assert kg._best_so_far == -2.0
    name, pts = dev.calls[-1][0], dev.calls[-1][1]
    assert name == "additional_mean" and np.allclose(pts.reshape(5, d), discrete)
    val = kg.compute_knowledge_gradient()
    grad = kg.compute_grad_knowledge_gradient()
    assert val == 0.5 and grad.shape == (2, d) and np.all(grad == 0.125)
    for call, want_grad in zip(dev.calls[-2:], (False, True)):
        (name, inner_t, b, disc, xq, xp, m_iter, best, normals, wg, nf) = call
        assert name == "kg" and wg == want_grad and nf == 0 and m_iter == M and best == -2.0
        assert tuple(inner_t)[:3] == (1, 6, 1) and np.allclose(b.ravel(), np.tile([-1.0, 1.0], d))
        assert np.allclose(disc.reshape(5, d), discrete) and np.allclose(xq.reshape(2, d), Xq) and np.allclose(xp.reshape(1, d), Xp)
        assert normals.size == ((M + 1) // 2) * 3 * (1 + g)   # antithetic table: ceil(M/2) rows of m = (q + p)(1 + g) draws
    # posterior mean and its gradient (knowledge_gradient.py:120-160)
    ps.set_current_point(Xq[:1])
    assert ps.compute_posterior_mean() == 0.25 and ps.compute_grad_posterior_mean().shape == (1, d)
    assert dev.calls[-1][0] == "posterior_mean" and np.allclose(dev.calls[-1][1].ravel(), Xq[0])
    # q,p-EI (expected_improvement.py:309-362): best_so_far = min of the function values (:124-176)
    ei = ref.expected_improvement.ExpectedImprovement(gp, points_to_sample=Xq, points_being_sampled=Xp, num_mc_iterations=M,
                                                      randomness=randomness)
    assert ei.compute_expected_improvement(force_monte_carlo=True) == 0.75
    ge = ei.compute_grad_expected_improvement(force_monte_carlo=True)
    assert ge.shape == (2, d)
    name, xq, xp, m_iter, best, normals, wg = dev.calls[-1]
    assert name == "ei" and wg and m_iter == M and abs(best - float(Y[:, 0].min())) < 1e-15
    assert np.allclose(xq.reshape(2, d), Xq) and np.allclose(xp.reshape(1, d), Xp) and normals.size == M * 3
