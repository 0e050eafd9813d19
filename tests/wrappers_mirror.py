"""Test-side drivers of the drop-in boundary (``cornell_moe_amd.GPP``, the stand-in for ``moe.build.GPP``).

TEST INFRASTRUCTURE, written for this repository (r6: rewritten -- until then this file was a condensed re-typing of the
reference's wrapper classes).  The parity tests drive the boundary through the reference's CALL SEQUENCE, so
the objects below answer to the names and constructor arguments of ``moe.optimal_learning.python.cpp_wrappers`` -- which is the
contract ``GPP.py`` is written against -- but they are thin, table-driven shells: one base class holds the points an acquisition
object carries, one helper binds constructor arguments, and every evaluator is a list of (method name, GPP function, result
shape).  The reference's OWN wrapper modules run on ``GPP.py`` in ``tests/test_reference_wrappers.py`` wherever the reference tree
is importable; these shells exist because it is not on the GPU box.

Call conventions followed (reference file under moe/optimal_learning/python/): data_containers.py:19-260, cpp_wrappers/
covariance.py:15-98, domain.py:15-105, optimization.py:250-437, gaussian_process.py:18-387, expected_improvement.py:22-367,
knowledge_gradient.py:20-596, knowledge_gradient_mcmc.py:19-420, expected_improvement_mcmc.py:22-260, log_likelihood.py:179-440.
"""
import collections
import copy

import numpy as np

from cornell_moe_amd import GPP as C_GP

numpy = np  # (tests reach `cw.numpy` in places)
DEFAULT_EXPECTED_IMPROVEMENT_MC_ITERATIONS = 10000  # python/constant.py
DEFAULT_MAX_NUM_THREADS = 4


# ---- flat lists in, shaped arrays out: what crosses the boundary ----
def cppify(array):
    return list(np.ravel(array))


def uncppify(array, expected_shape):
    return np.reshape(array, expected_shape)


def cppify_hyperparameters(hyperparameters):
    signal, lengths = hyperparameters[0], hyperparameters[1:]
    return [np.float64(signal), cppify(lengths)]


def _floats(array):
    return [float(v) for v in np.ravel(array)]


def _bind(obj, **fields):
    """obj._name = value for every keyword."""
    for name, value in fields.items():
        setattr(obj, "_" + name, value)


def _view(name):
    """read-only property over obj._name"""
    return property(lambda self: getattr(self, "_" + name))


def _rng(randomness, num_threads=1):
    if randomness is None:
        randomness = C_GP.RandomnessSourceContainer(num_threads)
        randomness.SetRandomizedUniformGeneratorSeed(0)
        randomness.SetRandomizedNormalRNGSeed(0)
    return randomness


# ---- data containers ----
class SamplePoint(collections.namedtuple("SamplePoint", "point value noise_variance")):
    __slots__ = ()

    def __new__(cls, point, value, noise_variance=0.0):
        if not (np.isfinite(noise_variance) and noise_variance >= 0.0):
            raise ValueError("noise_variance = {0} must be positive and finite!".format(noise_variance))
        return super().__new__(cls, point, value, noise_variance)


class HistoricalData:
    """points [n, dim], values [n, 1 + num_derivatives], noise [n]"""

    def __init__(self, dim, num_derivatives=0, sample_points=None):
        self._dim, self._num_derivatives = int(dim), int(num_derivatives)
        self._rows = [np.empty((0, self._dim)), np.empty((0, 1 + self._num_derivatives)), np.empty(0)]
        if sample_points:
            self.append_sample_points(sample_points)

    dim, num_derivatives = _view("dim"), _view("num_derivatives")
    num_sampled = property(lambda self: self._rows[0].shape[0])
    points_sampled = property(lambda self: self._rows[0])
    points_sampled_value = property(lambda self: self._rows[1])
    points_sampled_noise_variance = property(lambda self: self._rows[2])

    def append_historical_data(self, points_sampled, points_sampled_value, points_sampled_noise_variance):
        new = (np.reshape(points_sampled, (-1, self._dim)), np.reshape(points_sampled_value, (-1, 1 + self._num_derivatives)),
               np.ravel(points_sampled_noise_variance))
        self._rows = [np.concatenate([old, np.asarray(add, dtype=float)]) for old, add in zip(self._rows, new)]

    def append_sample_points(self, sample_points):
        if len(sample_points):
            self.append_historical_data([np.asarray(s[0], dtype=float) for s in sample_points],
                                        [np.ravel(s[1]) for s in sample_points],
                                        [float(s[2]) if len(s) > 2 else 0.0 for s in sample_points])


class SquareExponential:
    """[alpha, length_0 .. length_{dim-1}] (covariance.py:31-40); the boundary builds a Matern-5/2 kernel from them whatever this
    class is called, as the reference's does (gpp_python_gaussian_process.cpp:53)."""
    covariance_type = "square_exponential"

    def __init__(self, hyperparameters):
        self.set_hyperparameters(hyperparameters)

    def get_hyperparameters(self):
        return self._h.copy()

    def set_hyperparameters(self, hyperparameters):
        self._h = np.array(hyperparameters, dtype=float, copy=True)

    hyperparameters = property(get_hyperparameters, set_hyperparameters)
    num_hyperparameters = property(lambda self: self._h.size)


class ClosedInterval(collections.namedtuple("ClosedInterval", "min max")):
    __slots__ = ()
    length = property(lambda self: self.max - self.min)

    def is_inside(self, value):
        return not (value < self.min or value > self.max)


class TensorProductDomain:
    domain_type = "tensor_product"

    def __init__(self, domain_bounds):
        _bind(self, domain_bounds=[ClosedInterval(float(lo), float(hi)) for lo, hi in domain_bounds],
              domain_type=C_GP.DomainTypes.tensor_product)

    domain_bounds = _view("domain_bounds")
    dim = property(lambda self: len(self._domain_bounds))

    def check_point_inside(self, point):
        return all(iv.is_inside(x) for iv, x in zip(self._domain_bounds, point))


class GradientDescentParameters(C_GP.GradientDescentParameters):
    domain_bounds = None


class _OptimizerParameters:
    """what the optimisers behind the boundary read (optimization.py:250-300)"""

    def __init__(self, **kw):
        for name in ("domain_type", "objective_type", "optimizer_type", "num_random_samples", "optimizer_parameters"):
            setattr(self, name, kw.get(name))
        self.optimizer_parameters = self.optimizer_parameters or None


class GradientDescentOptimizer:
    """optimization.py:404-440: a container, not an optimiser -- the loops run behind the boundary."""
    optimizer_type = C_GP.OptimizerTypes.gradient_descent

    def __init__(self, domain, optimizable, optimizer_parameters, num_random_samples=None):
        self.domain, self.objective_function = domain, optimizable
        self.optimizer_parameters = _OptimizerParameters(
            domain_type=domain._domain_type, objective_type=getattr(optimizable, "objective_type", None),
            optimizer_type=self.optimizer_type, num_random_samples=num_random_samples or 0,
            optimizer_parameters=optimizer_parameters)

    def optimize(self, **kwargs):
        raise NotImplementedError("the optimisation loops live behind the boundary (multistart_* functions)")


# ---- the GP ----
class GaussianProcess:
    """gaussian_process.py:18-387"""
    # (method, result shape as a function of (self, number of points k, clamped num_derivatives nd)) -- None: a vector
    _QUERIES = {
        "compute_mean_of_points": None,
        "compute_mean_of_additional_points": None,
        "compute_variance_of_points": lambda s, k, nd: (k * s._g1, k * s._g1),
        "compute_cholesky_variance_of_points": lambda s, k, nd: (k * s._g1, k * s._g1),
    }

    def __init__(self, covariance_function, noise_variance, historical_data, derivatives):
        kept = [copy.deepcopy(v) for v in (covariance_function, noise_variance, historical_data, derivatives)]
        _bind(self, covariance=kept[0], noise_variance=kept[1], historical_data=kept[2], derivatives=kept[3])
        self._num_derivatives = len(cppify(self._derivatives))
        self._g1 = 1 + self._num_derivatives
        data = self._historical_data
        self._gaussian_process = C_GP.GaussianProcess(
            cppify_hyperparameters(self._covariance.hyperparameters), cppify(data.points_sampled), cppify(data.points_sampled_value),
            cppify(self._noise_variance), cppify(self._derivatives), self._num_derivatives, data.dim, data.num_sampled)

    dim = property(lambda self: self._gaussian_process.dim)
    num_sampled = property(lambda self: self._gaussian_process.num_sampled)
    num_derivatives, derivatives, noise_variance = _view("num_derivatives"), _view("derivatives"), _view("noise_variance")

    def get_covariance_copy(self):
        return copy.deepcopy(self._covariance)

    def get_historical_data_copy(self):
        return copy.deepcopy(self._historical_data)

    def _query(self, name, points):
        k = points.shape[0]
        flat = getattr(self._gaussian_process, name)(cppify(points), k)
        shape = self._QUERIES[name]
        return np.array(flat) if shape is None else uncppify(flat, shape(self, k, 0))

    def compute_mean_of_points(self, points_to_sample):
        return self._query("compute_mean_of_points", points_to_sample)

    def compute_mean_of_additional_points(self, discrete_pts):
        return self._query("compute_mean_of_additional_points", discrete_pts)

    def compute_variance_of_points(self, points_to_sample):
        return self._query("compute_variance_of_points", points_to_sample)

    def compute_cholesky_variance_of_points(self, points_to_sample):
        return self._query("compute_cholesky_variance_of_points", points_to_sample)

    @staticmethod
    def _first(num_points, num_derivatives):  # how many of the points are differentiated (-1: all)
        return num_points if num_derivatives < 0 else min(num_points, num_derivatives)

    def compute_grad_mean_of_points(self, points_to_sample, num_derivatives=-1):
        nd = self._first(points_to_sample.shape[0], num_derivatives)
        flat = self._gaussian_process.compute_grad_mean_of_points(cppify(points_to_sample[:nd, ...]), nd)
        return uncppify(flat, (nd, self._g1, self.dim))

    def _grad_second_moment(self, name, points, num_derivatives):
        k = points.shape[0]
        nd = self._first(k, num_derivatives)
        flat = getattr(self._gaussian_process, name)(cppify(points), k, nd)
        return uncppify(flat, (nd, k * self._g1, k * self._g1, self.dim))

    def compute_grad_variance_of_points(self, points_to_sample, num_derivatives=-1):
        return self._grad_second_moment("compute_grad_variance_of_points", points_to_sample, num_derivatives)

    def compute_grad_cholesky_variance_of_points(self, points_to_sample, num_derivatives=-1):
        return self._grad_second_moment("compute_grad_cholesky_variance_of_points", points_to_sample, num_derivatives)

    def add_sampled_points(self, sampled_points):
        before = self.num_sampled
        data = self._historical_data
        data.append_sample_points(sampled_points)
        self._gaussian_process.add_sampled_points(cppify(data.points_sampled[before:, ...]), cppify(data.points_sampled_value[before:]),
                                                  len(sampled_points))

    def sample_point_from_gp(self, point_to_sample, noise_variance=0.0):
        return np.array(self._gaussian_process.sample_point_from_gp(cppify(point_to_sample)))


# ---- acquisition objects: the points they carry, and the evaluator calls ----
class _Acquisition:
    """Holds points_to_sample [q, dim] / points_being_sampled [p, dim]; subclasses give `_call(kind)` = the GPP call of the value
    ("f") or the gradient ("g") and `_grad_shape()`; `_NAMES` = (value method, gradient method) as the reference calls them."""
    objective_type = None
    _NAMES = ()

    def _carry(self, dim, points_to_sample, points_being_sampled=None, randomness=None, rows=1, copy_points=False):
        if points_to_sample is None:
            points_to_sample = np.zeros((rows, dim))
        elif copy_points:
            points_to_sample = np.copy(np.atleast_2d(points_to_sample))
        _bind(self, points_to_sample=points_to_sample,
              points_being_sampled=np.array([]) if points_being_sampled is None else np.copy(points_being_sampled),
              randomness=_rng(randomness))

    num_to_sample = property(lambda self: self._points_to_sample.shape[0])
    num_being_sampled = property(lambda self: self._points_being_sampled.shape[0])
    problem_size = property(lambda self: self.num_to_sample * self.dim)

    def get_current_point(self):
        return self._points_to_sample.copy()

    def set_current_point(self, points_to_sample):
        self._points_to_sample = np.array(np.atleast_2d(points_to_sample), copy=True)

    current_point = property(get_current_point, set_current_point)

    def _grad_shape(self):
        return (self.num_to_sample, self.dim)

    def compute_objective_function(self, force_monte_carlo=False, **kwargs):
        return self._call("f", force_monte_carlo)

    def compute_grad_objective_function(self, force_monte_carlo=False, **kwargs):
        return uncppify(self._call("g", force_monte_carlo), self._grad_shape())

    def compute_hessian_objective_function(self, **kwargs):
        raise NotImplementedError("no Hessian behind this boundary")

    def __init_subclass__(cls, **kw):
        super().__init_subclass__(**kw)
        if cls._NAMES:  # the reference's method names for the two calls
            setattr(cls, cls._NAMES[0], cls.compute_objective_function)
            setattr(cls, cls._NAMES[1], cls.compute_grad_objective_function)

    def _list_randomness(self, randomness, max_num_threads):
        return self._randomness if (randomness is None and max_num_threads == 1) else _rng(randomness, max_num_threads)


def _best_observed(historical_data):
    values = historical_data.points_sampled_value
    return np.amin(values[:, 0]) if values.size > 0 else np.finfo(np.float64).max


class ExpectedImprovement(_Acquisition):
    """expected_improvement.py:109-367 (q,p-EI)"""
    _NAMES = ("compute_expected_improvement", "compute_grad_expected_improvement")

    def __init__(self, gaussian_process, points_to_sample=None, points_being_sampled=None,
                 num_mc_iterations=DEFAULT_EXPECTED_IMPROVEMENT_MC_ITERATIONS, randomness=None):
        _bind(self, gaussian_process=gaussian_process, num_mc_iterations=num_mc_iterations,
              best_so_far=_best_observed(gaussian_process._historical_data))
        self._carry(gaussian_process.dim, points_to_sample, points_being_sampled, randomness)

    dim = property(lambda self: self._gaussian_process.dim)

    def _call(self, kind, force_monte_carlo):
        fn = C_GP.compute_expected_improvement if kind == "f" else C_GP.compute_grad_expected_improvement
        return fn(self._gaussian_process._gaussian_process, cppify(self._points_to_sample), cppify(self._points_being_sampled),
                  self.num_to_sample, self.num_being_sampled, self._num_mc_iterations, self._best_so_far, force_monte_carlo,
                  self._randomness)

    def evaluate_at_point_list(self, points_to_evaluate, randomness=None, max_num_threads=DEFAULT_MAX_NUM_THREADS, status=None):
        count, q, _ = points_to_evaluate.shape
        return np.array(C_GP.evaluate_EI_at_point_list(
            self._gaussian_process._gaussian_process, cppify(points_to_evaluate), cppify(self._points_being_sampled), count, q,
            self.num_being_sampled, self._best_so_far, self._num_mc_iterations, max_num_threads,
            self._list_randomness(randomness, max_num_threads), {} if status is None else status))


def multistart_expected_improvement_optimization(ei_optimizer, num_multistarts, num_to_sample, use_gpu=False, which_gpu=0,
                                                 randomness=None, max_num_threads=DEFAULT_MAX_NUM_THREADS, status=None):
    """expected_improvement.py:22-105 (num_multistarts is unused there as well: the count is in optimizer_parameters)"""
    ei = ei_optimizer.objective_function
    best = C_GP.multistart_expected_improvement_optimization(
        ei_optimizer.optimizer_parameters, ei._gaussian_process._gaussian_process, _floats(ei_optimizer.domain.domain_bounds),
        cppify(ei._points_being_sampled), num_to_sample, ei.num_being_sampled, ei._best_so_far, ei._num_mc_iterations,
        max_num_threads, use_gpu, which_gpu, _rng(randomness, max_num_threads), {} if status is None else status)
    return uncppify(best, (num_to_sample, ei.dim))


class PosteriorMean(_Acquisition):
    """knowledge_gradient.py:20-170: -mu(x), fidelity coordinates pinned to 1"""
    _NAMES = ("compute_posterior_mean", "compute_grad_posterior_mean")

    def __init__(self, gaussian_process, num_fidelity, points_to_sample=None, randomness=None):
        _bind(self, gaussian_process=gaussian_process, num_fidelity=num_fidelity)
        self._carry(gaussian_process.dim, points_to_sample, None, randomness)

    dim = property(lambda self: self._gaussian_process.dim)
    problem_size = property(lambda self: self.dim - self._num_fidelity)

    def _grad_shape(self):
        return (1, self.dim - self._num_fidelity)

    def _call(self, kind, force_monte_carlo):
        fn = C_GP.compute_posterior_mean if kind == "f" else C_GP.compute_grad_posterior_mean
        return fn(self._gaussian_process._gaussian_process, self._num_fidelity, cppify(self._points_to_sample))


def _pinned(discrete_pts, dim):
    """the discretised points with the fidelity coordinates set to 1"""
    full = np.ones((discrete_pts.shape[0], dim))
    full[:, :discrete_pts.shape[1]] = discrete_pts
    return full


class KnowledgeGradient(_Acquisition):
    """knowledge_gradient.py:309-596 (q-KG / d-KG)"""
    _NAMES = ("compute_knowledge_gradient", "compute_grad_knowledge_gradient")

    def __init__(self, gaussian_process, num_fidelity, inner_optimizer, discrete_pts, points_to_sample=None,
                 points_being_sampled=None, num_mc_iterations=DEFAULT_EXPECTED_IMPROVEMENT_MC_ITERATIONS, randomness=None):
        _bind(self, gaussian_process=gaussian_process, num_fidelity=num_fidelity, inner_optimizer=inner_optimizer,
              discrete_pts=np.copy(discrete_pts), num_mc_iterations=num_mc_iterations)
        self._mu_star = gaussian_process.compute_mean_of_additional_points(_pinned(discrete_pts, discrete_pts.shape[1] + num_fidelity))
        self._best_so_far = np.amin(self._mu_star)  # (:366-368)
        self._carry(gaussian_process.dim, points_to_sample, points_being_sampled, randomness)

    dim = property(lambda self: self._gaussian_process.dim)
    discrete = property(lambda self: self._discrete_pts.shape[0])

    def _head(self):
        inner = self._inner_optimizer
        return (self._gaussian_process._gaussian_process, self._num_fidelity, inner.optimizer_parameters, cppify(inner.domain.domain_bounds))

    def _call(self, kind, force_monte_carlo):
        fn = C_GP.compute_knowledge_gradient if kind == "f" else C_GP.compute_grad_knowledge_gradient
        return fn(*self._head(), cppify(self._discrete_pts), cppify(self._points_to_sample), cppify(self._points_being_sampled),
                  self.discrete, self.num_to_sample, self.num_being_sampled, self._num_mc_iterations, self._best_so_far,
                  self._randomness)

    def evaluate_at_point_list(self, points_to_evaluate, randomness=None, max_num_threads=DEFAULT_MAX_NUM_THREADS, status=None):
        count, q, _ = points_to_evaluate.shape
        packed = self._discrete_pts if self.num_being_sampled == 0 else np.concatenate((self._discrete_pts, self._points_being_sampled))
        return np.array(C_GP.evaluate_KG_at_point_list(
            *self._head(), cppify(packed), cppify(points_to_evaluate), count, self.discrete, q, self.num_being_sampled,
            self._best_so_far, self._num_mc_iterations, max_num_threads, self._list_randomness(randomness, max_num_threads),
            {} if status is None else status))


def multistart_knowledge_gradient_optimization(kg_optimizer, inner_optimizer, num_multistarts, discrete_pts, num_to_sample,
                                               num_pts, randomness=None, max_num_threads=DEFAULT_MAX_NUM_THREADS, status=None):
    """knowledge_gradient.py:225-307"""
    kg = kg_optimizer.objective_function
    best = C_GP.multistart_knowledge_gradient_optimization(
        kg_optimizer.optimizer_parameters, inner_optimizer.optimizer_parameters, kg._gaussian_process._gaussian_process,
        kg._num_fidelity, cppify(kg_optimizer.domain.domain_bounds), cppify(discrete_pts), cppify(kg._points_being_sampled),
        num_pts, num_to_sample, kg.num_being_sampled, kg._best_so_far, kg._num_mc_iterations, max_num_threads,
        _rng(randomness, max_num_threads), {} if status is None else status)
    return uncppify(best, (num_to_sample, kg.dim))


# ---- the MCMC-averaged objects (knowledge_gradient_mcmc.py, expected_improvement_mcmc.py) ----
class GaussianProcessMCMC:
    """knowledge_gradient_mcmc.py:163-240: one GP per hyper-parameter sample over the same data.  hyperparameters_list
    [num_mcmc][1 + dim], noise_variance_list [num_mcmc][1 + num_derivatives]."""

    def __init__(self, hyperparameters_list, noise_variance_list, historical_data, derivatives):
        as_rows = lambda a: copy.deepcopy(np.asarray(a, dtype=np.float64))  # noqa: E731
        _bind(self, hyperparameters_list=as_rows(hyperparameters_list), noise_variance_list=as_rows(noise_variance_list),
              historical_data=copy.deepcopy(historical_data), derivatives=copy.deepcopy(derivatives))
        self._num_mcmc = self._hyperparameters_list.shape[0]
        self._num_derivatives = len(cppify(self._derivatives))
        data = self._historical_data
        self._gaussian_process_mcmc = C_GP.GaussianProcessMCMC(
            cppify(self._hyperparameters_list), cppify(self._noise_variance_list), cppify(data.points_sampled),
            cppify(data.points_sampled_value), cppify(self._derivatives), self._num_mcmc, self._num_derivatives, data.dim,
            data.num_sampled)

    dim = property(lambda self: self._historical_data.dim)
    num_sampled = property(lambda self: self._historical_data.num_sampled)
    num_derivatives = _view("num_derivatives")
    derivatives = property(lambda self: np.copy(self._derivatives))
    noise_variance_list = property(lambda self: np.copy(self._noise_variance_list))

    def get_historical_data_copy(self):
        return copy.deepcopy(self._historical_data)

    def member_models(self):
        """the per-sample GaussianProcess objects the reference keeps beside the ensemble (log_likelihood_mcmc.py:238-262: `models`)"""
        return [GaussianProcess(SquareExponential(h), nv, self._historical_data, list(self._derivatives))
                for h, nv in zip(self._hyperparameters_list, self._noise_variance_list)]


class PosteriorMeanMCMC(_Acquisition):
    """knowledge_gradient_mcmc.py:19-160: the posterior mean averaged over the per-sample GPs"""
    _NAMES = ("compute_posterior_mean_mcmc", "compute_grad_posterior_mean_mcmc")

    def __init__(self, gaussian_process_list, num_fidelity, points_to_sample=None, randomness=None):
        _bind(self, gaussian_process_list=gaussian_process_list, num_fidelity=num_fidelity)
        self._carry(gaussian_process_list[0].dim, points_to_sample, None, randomness)

    dim = property(lambda self: self._gaussian_process_list[0].dim)
    problem_size = property(lambda self: self.dim - self._num_fidelity)

    def _grad_shape(self):
        return (1, self.dim - self._num_fidelity)

    def _call(self, kind, force_monte_carlo):
        fn = C_GP.compute_posterior_mean if kind == "f" else C_GP.compute_grad_posterior_mean
        per_gp = [np.asarray(fn(gp._gaussian_process, self._num_fidelity, cppify(self._points_to_sample)), dtype=float)
                  for gp in self._gaussian_process_list]
        total = per_gp[0] * 0.0
        for v in per_gp:  # (added up in member order, as the reference's loop does)
            total = total + v
        mean = total / len(per_gp)
        return float(mean) if kind == "f" else mean


class KnowledgeGradientMCMC(_Acquisition):
    """knowledge_gradient_mcmc.py:243-420"""
    _NAMES = ("compute_knowledge_gradient_mcmc", "compute_grad_knowledge_gradient_mcmc")

    def __init__(self, gaussian_process_mcmc, gaussian_process_list, num_fidelity, inner_optimizer, discrete_pts_list,
                 num_to_sample, points_to_sample=None, points_being_sampled=None,
                 num_mc_iterations=DEFAULT_EXPECTED_IMPROVEMENT_MC_ITERATIONS, randomness=None):
        _bind(self, gaussian_process_mcmc=gaussian_process_mcmc, gaussian_process_list=gaussian_process_list,
              num_fidelity=num_fidelity, inner_optimizer=inner_optimizer, num_mc_iterations=num_mc_iterations,
              discrete_pts_list=[np.copy(d) for d in discrete_pts_list])
        self._best_so_far_list = [np.amin(gp.compute_mean_of_additional_points(_pinned(d, gp.dim)))  # (:268-274)
                                  for d, gp in zip(discrete_pts_list, gaussian_process_list)]
        self._carry(gaussian_process_mcmc.dim, points_to_sample, points_being_sampled, randomness, rows=num_to_sample)

    dim = property(lambda self: self._gaussian_process_mcmc.dim)
    discrete = property(lambda self: self._discrete_pts_list[0].shape[0])

    def _head(self):
        inner = self._inner_optimizer
        return (self._gaussian_process_mcmc._gaussian_process_mcmc, self._num_fidelity, inner.optimizer_parameters,
                _floats(inner.domain.domain_bounds))

    def _call(self, kind, force_monte_carlo):
        fn = C_GP.compute_knowledge_gradient_mcmc if kind == "f" else C_GP.compute_grad_knowledge_gradient_mcmc
        return fn(*self._head(), cppify(np.array(self._discrete_pts_list)), cppify(self._points_to_sample),
                  cppify(self._points_being_sampled), self.discrete, self.num_to_sample, self.num_being_sampled,
                  self._num_mc_iterations, cppify(np.array(self._best_so_far_list)), self._randomness)

    def evaluate_at_point_list(self, points_to_evaluate, randomness=None, max_num_threads=DEFAULT_MAX_NUM_THREADS, status=None):
        """(:292-310, with the two list arguments in the order the C++ wrapper declares them: the reference's Python passes them
        swapped -- see GPP.evaluate_KG_mcmc_at_point_list)"""
        count, q, _ = points_to_evaluate.shape
        packed = np.concatenate((np.ravel(self._discrete_pts_list), np.ravel(self._points_being_sampled)))
        return np.array(C_GP.evaluate_KG_mcmc_at_point_list(
            *self._head(), cppify(points_to_evaluate), cppify(packed), count, self.discrete, q, self.num_being_sampled,
            cppify(np.array(self._best_so_far_list)), self._num_mc_iterations, max_num_threads,
            self._list_randomness(randomness, max_num_threads), {} if status is None else status))


def multistart_knowledge_gradient_mcmc_optimization(kg_optimizer, inner_optimizer, num_multistarts, discrete_pts, num_to_sample,
                                                    num_pts, randomness=None, max_num_threads=DEFAULT_MAX_NUM_THREADS, status=None):
    """knowledge_gradient_mcmc.py:200-240"""
    kg = kg_optimizer.objective_function
    best = C_GP.multistart_knowledge_gradient_mcmc_optimization(
        kg_optimizer.optimizer_parameters, inner_optimizer.optimizer_parameters, kg._gaussian_process_mcmc._gaussian_process_mcmc,
        kg._num_fidelity, _floats(kg_optimizer.domain.domain_bounds), cppify(np.array(discrete_pts)), cppify(kg._points_being_sampled),
        num_pts, num_to_sample, kg.num_being_sampled, cppify(np.array(kg._best_so_far_list)), kg._num_mc_iterations, max_num_threads,
        _rng(randomness, max_num_threads), {} if status is None else status)
    return uncppify(best, (num_to_sample, kg.dim))


class ExpectedImprovementMCMC(_Acquisition):
    """expected_improvement_mcmc.py:60-260"""
    _NAMES = ("compute_expected_improvement", "compute_grad_expected_improvement")

    def __init__(self, gaussian_process_mcmc, num_to_sample, points_to_sample=None, points_being_sampled=None,
                 num_mc_iterations=DEFAULT_EXPECTED_IMPROVEMENT_MC_ITERATIONS, randomness=None):
        _bind(self, gaussian_process_mcmc=gaussian_process_mcmc, num_mc_iterations=num_mc_iterations)
        self._best_so_far_list = gaussian_process_mcmc._num_mcmc * [_best_observed(gaussian_process_mcmc._historical_data)]
        self._carry(gaussian_process_mcmc.dim, points_to_sample, points_being_sampled, randomness, rows=num_to_sample, copy_points=True)

    dim = property(lambda self: self._gaussian_process_mcmc.dim)

    def _call(self, kind, force_monte_carlo):
        fn = C_GP.compute_expected_improvement_mcmc if kind == "f" else C_GP.compute_grad_expected_improvement_mcmc
        return fn(self._gaussian_process_mcmc._gaussian_process_mcmc, cppify(self._points_to_sample), cppify(self._points_being_sampled),
                  self.num_to_sample, self.num_being_sampled, self._num_mc_iterations, cppify(np.array(self._best_so_far_list)),
                  self._randomness)

    def evaluate_at_point_list(self, points_to_evaluate, randomness=None, max_num_threads=DEFAULT_MAX_NUM_THREADS, status=None):
        count, q, _ = points_to_evaluate.shape
        return np.array(C_GP.evaluate_EI_mcmc_at_point_list(
            self._gaussian_process_mcmc._gaussian_process_mcmc, cppify(points_to_evaluate), cppify(self._points_being_sampled), count, q,
            self.num_being_sampled, cppify(np.array(self._best_so_far_list)), self._num_mc_iterations, max_num_threads,
            self._list_randomness(randomness, max_num_threads), {} if status is None else status))


def multistart_expected_improvement_mcmc_optimization(ei_optimizer, num_multistarts, num_to_sample, randomness=None,
                                                      max_num_threads=DEFAULT_MAX_NUM_THREADS, status=None):
    """expected_improvement_mcmc.py:22-57"""
    ei = ei_optimizer.objective_function
    best = C_GP.multistart_expected_improvement_mcmc_optimization(
        ei_optimizer.optimizer_parameters, ei._gaussian_process_mcmc._gaussian_process_mcmc, _floats(ei_optimizer.domain.domain_bounds),
        cppify(ei._points_being_sampled), num_to_sample, ei.num_being_sampled, cppify(np.array(ei._best_so_far_list)),
        ei._num_mc_iterations, max_num_threads, _rng(randomness, max_num_threads), {} if status is None else status)
    return uncppify(best, (num_to_sample, ei.dim))


# ---- log likelihood (log_likelihood.py) ----
class GaussianProcessLogLikelihood:
    """The log marginal likelihood of the data as a function of [covariance hyper-parameters | noise variances]: what the
    reference's hyper-parameter optimisers and its emcee driver evaluate (log_likelihood.py:230-400); value and gradient come from
    the device (moe_ll_evaluate / moe_ll_grad behind GPP.compute_log_likelihood / compute_hyperparameter_grad_log_likelihood)."""

    def __init__(self, covariance_function, historical_data, noise_variance, derivatives,
                 log_likelihood_type=C_GP.LogLikelihoodTypes.log_marginal_likelihood):
        _bind(self, covariance=copy.deepcopy(covariance_function), historical_data=copy.deepcopy(historical_data),
              noise_variance=np.array(noise_variance, dtype=float, copy=True).ravel(), derivatives=[int(v) for v in derivatives])
        self._num_derivatives = len(self._derivatives)
        self.objective_type = log_likelihood_type

    dim = property(lambda self: self._historical_data.dim)
    noise_variance, derivatives, num_derivatives = _view("noise_variance"), _view("derivatives"), _view("num_derivatives")
    cov_hyperparameters = property(lambda self: self._covariance.hyperparameters)
    num_hyperparameters = property(lambda self: self._covariance.num_hyperparameters + self._noise_variance.size)
    problem_size = num_hyperparameters
    _num_sampled = property(lambda self: self._historical_data.num_sampled)
    _points_sampled = property(lambda self: self._historical_data.points_sampled)
    _points_sampled_value = property(lambda self: self._historical_data.points_sampled_value)
    _points_sampled_noise_variance = property(lambda self: self._historical_data.points_sampled_noise_variance)

    def get_hyperparameters(self):
        return np.concatenate([np.ravel(self._covariance.hyperparameters), self._noise_variance])

    def set_hyperparameters(self, hyperparameters):
        split = self._covariance.num_hyperparameters
        self._covariance.hyperparameters = hyperparameters[:split]
        self._noise_variance = np.array(hyperparameters[split:], dtype=float).ravel()

    hyperparameters = property(get_hyperparameters, set_hyperparameters)
    current_point = hyperparameters

    def get_covariance_copy(self):
        return copy.deepcopy(self._covariance)

    def get_historical_data_copy(self):
        return copy.deepcopy(self._historical_data)

    def _operands(self):
        return (cppify(self._points_sampled), cppify(self._points_sampled_value), self.dim, self._num_sampled, self.objective_type,
                cppify_hyperparameters(self.cov_hyperparameters), cppify(self._derivatives), self._num_derivatives,
                cppify(self.noise_variance))

    def compute_log_likelihood(self):
        return C_GP.compute_log_likelihood(*self._operands())

    def compute_grad_log_likelihood(self):
        return np.array(C_GP.compute_hyperparameter_grad_log_likelihood(*self._operands()))

    compute_objective_function, compute_grad_objective_function = compute_log_likelihood, compute_grad_log_likelihood


class GaussianProcessLogMarginalLikelihood(GaussianProcessLogLikelihood):
    """(log_likelihood.py:403-440)"""

    def __init__(self, covariance_function, historical_data, noise_variance, derivatives):
        super().__init__(covariance_function, historical_data, noise_variance, derivatives, C_GP.LogLikelihoodTypes.log_marginal_likelihood)


def evaluate_log_likelihood_at_hyperparameter_list(log_likelihood_evaluator, hyperparameters_to_evaluate, max_num_threads=4, status=None):
    """log_likelihood.py:179-227: one value per row of hyperparameters_to_evaluate [rows][num_hyperparameters]; the rows of a call
    are factorised together on the device."""
    rows = np.asarray(hyperparameters_to_evaluate, dtype=float)
    ll = log_likelihood_evaluator
    return np.array(C_GP.evaluate_log_likelihood_at_hyperparameter_list(
        cppify(rows), cppify(ll._points_sampled), cppify(ll._points_sampled_value), ll.dim, ll._num_sampled, ll.objective_type,
        cppify_hyperparameters(ll.cov_hyperparameters), cppify(ll.noise_variance), cppify(ll.derivatives), ll.num_derivatives,
        rows.shape[0], max_num_threads, {} if status is None else status))
