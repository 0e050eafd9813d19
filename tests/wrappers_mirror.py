"""Host-side mirror of ``moe.optimal_learning.python.cpp_wrappers`` for the hot path: same class names, constructor
arguments, method names, array shapes and error behaviour as the reference's wrappers, written against ``cornell_moe_amd.GPP``
(this package's stand-in for ``moe.build.GPP``).  TEST INFRASTRUCTURE, not a deliverable (r3: moved out of the package): what
ships is the C ABI + ``GPP.py``, and the reference's own wrapper files run unchanged on top of that module (INTEGRATION.md
route A).  This mirror exists because the reference tree (and its ``future`` dependency) is not importable on the GPU box,
so that the parity tests can drive the boundary through the reference's call sequence.

Mirrored (reference file under moe/optimal_learning/python/): data_containers.py:19-260 (SamplePoint, HistoricalData),
cpp_wrappers/covariance.py:15-98, domain.py:15-105, optimization.py:250-437, gaussian_process.py:18-387,
expected_improvement.py:109-367, knowledge_gradient.py:20-596.
"""
import collections
import copy

import numpy

from cornell_moe_amd import GPP as C_GP

DEFAULT_EXPECTED_IMPROVEMENT_MC_ITERATIONS = 10000  # moe/optimal_learning/python/constant.py
DEFAULT_MAX_NUM_THREADS = 4


def cppify(array):
    return list(numpy.ravel(array))


def uncppify(array, expected_shape):
    return numpy.reshape(array, expected_shape)


def cppify_hyperparameters(hyperparameters):
    return [numpy.float64(hyperparameters[0]), cppify(hyperparameters[1:])]


_BaseSamplePoint = collections.namedtuple("_BaseSamplePoint", ["point", "value", "noise_variance"])


class SamplePoint(_BaseSamplePoint):
    __slots__ = ()

    def __new__(cls, point, value, noise_variance=0.0):
        if noise_variance >= 0.0 and numpy.isfinite(noise_variance):
            return super(SamplePoint, cls).__new__(cls, point, value, noise_variance)
        raise ValueError("noise_variance = {0} must be positive and finite!".format(noise_variance))


class HistoricalData(object):
    """(points_sampled [n, dim], points_sampled_value [n, 1 + num_derivatives], noise) container."""

    def __init__(self, dim, num_derivatives=0, sample_points=None):
        self._dim, self._num_derivatives = int(dim), int(num_derivatives)
        self._points_sampled = numpy.empty((0, self._dim))
        self._points_sampled_value = numpy.empty((0, 1 + self._num_derivatives))
        self._points_sampled_noise_variance = numpy.empty(0)
        if sample_points:
            self.append_sample_points(sample_points)

    dim = property(lambda self: self._dim)
    num_derivatives = property(lambda self: self._num_derivatives)
    num_sampled = property(lambda self: self._points_sampled.shape[0])
    points_sampled = property(lambda self: self._points_sampled)
    points_sampled_value = property(lambda self: self._points_sampled_value)
    points_sampled_noise_variance = property(lambda self: self._points_sampled_noise_variance)

    def append_sample_points(self, sample_points):
        if len(sample_points) == 0:
            return
        pts = numpy.array([numpy.asarray(p[0], dtype=float) for p in sample_points]).reshape(-1, self._dim)
        vals = numpy.array([numpy.ravel(p[1]) for p in sample_points], dtype=float).reshape(-1, 1 + self._num_derivatives)
        noise = numpy.array([float(p[2]) if len(p) > 2 else 0.0 for p in sample_points])
        self._points_sampled = numpy.vstack([self._points_sampled, pts])
        self._points_sampled_value = numpy.vstack([self._points_sampled_value, vals])
        self._points_sampled_noise_variance = numpy.concatenate([self._points_sampled_noise_variance, noise])

    def append_historical_data(self, points_sampled, points_sampled_value, points_sampled_noise_variance):
        self._points_sampled = numpy.vstack([self._points_sampled, numpy.reshape(points_sampled, (-1, self._dim))])
        self._points_sampled_value = numpy.vstack([self._points_sampled_value,
                                                   numpy.reshape(points_sampled_value, (-1, 1 + self._num_derivatives))])
        self._points_sampled_noise_variance = numpy.concatenate([self._points_sampled_noise_variance,
                                                                 numpy.ravel(points_sampled_noise_variance)])


class SquareExponential(object):
    """hyperparameters = [alpha, length_0 .. length_{dim-1}]  (cpp_wrappers/covariance.py:31-40).  As in the reference the
    C++ side builds a Matern-5/2 kernel from these regardless of this class' name (gpp_python_gaussian_process.cpp:53)."""
    covariance_type = "square_exponential"

    def __init__(self, hyperparameters):
        self.hyperparameters = numpy.copy(hyperparameters)

    @property
    def num_hyperparameters(self):
        return self.hyperparameters.size

    def get_hyperparameters(self):
        return numpy.copy(self._hyperparameters)

    def set_hyperparameters(self, hyperparameters):
        self._hyperparameters = numpy.copy(hyperparameters)

    hyperparameters = property(get_hyperparameters, set_hyperparameters)


class ClosedInterval(collections.namedtuple("ClosedInterval", ["min", "max"])):
    __slots__ = ()

    @property
    def length(self):
        return self.max - self.min

    def is_inside(self, value):
        return self.min <= value <= self.max


class TensorProductDomain(object):
    domain_type = "tensor_product"

    def __init__(self, domain_bounds):
        self._domain_bounds = [ClosedInterval(float(b[0]), float(b[1])) for b in domain_bounds]
        self._domain_type = C_GP.DomainTypes.tensor_product

    dim = property(lambda self: len(self._domain_bounds))
    domain_bounds = property(lambda self: self._domain_bounds)

    def check_point_inside(self, point):
        return all(b.is_inside(x) for b, x in zip(self._domain_bounds, point))


class GradientDescentParameters(C_GP.GradientDescentParameters):
    domain_bounds = None


class _CppOptimizerParameters(object):
    __slots__ = ("domain_type", "objective_type", "optimizer_type", "num_random_samples", "optimizer_parameters")

    def __init__(self, domain_type=None, objective_type=None, optimizer_type=None, num_random_samples=None,
                 optimizer_parameters=None):
        self.domain_type = domain_type
        self.objective_type = objective_type
        self.optimizer_type = optimizer_type
        self.num_random_samples = num_random_samples
        self.optimizer_parameters = optimizer_parameters if optimizer_parameters else None


class GradientDescentOptimizer(object):
    """cpp_wrappers/optimization.py:404-440: a container the C++-side (here: device-side) optimisers read."""

    def __init__(self, domain, optimizable, optimizer_parameters, num_random_samples=None):
        self.domain = domain
        self.objective_function = optimizable
        self.optimizer_type = C_GP.OptimizerTypes.gradient_descent
        self.optimizer_parameters = _CppOptimizerParameters(
            domain_type=domain._domain_type,
            objective_type=getattr(optimizable, "objective_type", None),
            optimizer_type=self.optimizer_type,
            num_random_samples=0 if num_random_samples is None else num_random_samples,
            optimizer_parameters=optimizer_parameters,
        )

    def optimize(self, **kwargs):
        raise NotImplementedError("C++ wrapper currently does not support optimization member functions.")


class GaussianProcess(object):
    """cpp_wrappers/gaussian_process.py:18-387."""

    def __init__(self, covariance_function, noise_variance, historical_data, derivatives):
        self._covariance = copy.deepcopy(covariance_function)
        self._historical_data = copy.deepcopy(historical_data)
        self._noise_variance = copy.deepcopy(noise_variance)
        self._derivatives = copy.deepcopy(derivatives)
        self._num_derivatives = len(cppify(self._derivatives))
        self._gaussian_process = C_GP.GaussianProcess(
            cppify_hyperparameters(self._covariance.hyperparameters),
            cppify(self._historical_data.points_sampled),
            cppify(self._historical_data.points_sampled_value),
            cppify(self._noise_variance),
            cppify(self._derivatives),
            self._num_derivatives,
            self._historical_data.dim,
            self._historical_data.num_sampled,
        )

    dim = property(lambda self: self._gaussian_process.dim)
    num_sampled = property(lambda self: self._gaussian_process.num_sampled)
    num_derivatives = property(lambda self: self._num_derivatives)
    derivatives = property(lambda self: self._derivatives)
    noise_variance = property(lambda self: self._noise_variance)

    def get_covariance_copy(self):
        return copy.deepcopy(self._covariance)

    def get_historical_data_copy(self):
        return copy.deepcopy(self._historical_data)

    def _clamp_num_derivatives(self, num_points, num_derivatives):
        return num_points if num_derivatives < 0 else min(num_points, num_derivatives)

    def compute_mean_of_points(self, points_to_sample):
        return numpy.array(self._gaussian_process.compute_mean_of_points(cppify(points_to_sample), points_to_sample.shape[0]))

    def compute_mean_of_additional_points(self, discrete_pts):
        return numpy.array(self._gaussian_process.compute_mean_of_additional_points(cppify(discrete_pts),
                                                                                    discrete_pts.shape[0]))

    def compute_grad_mean_of_points(self, points_to_sample, num_derivatives=-1):
        num_derivatives = self._clamp_num_derivatives(points_to_sample.shape[0], num_derivatives)
        grad_mu = self._gaussian_process.compute_grad_mean_of_points(cppify(points_to_sample[:num_derivatives, ...]),
                                                                     num_derivatives)
        return uncppify(grad_mu, (num_derivatives, 1 + self._num_derivatives, self.dim))

    def compute_variance_of_points(self, points_to_sample):
        k = points_to_sample.shape[0]
        var = self._gaussian_process.compute_variance_of_points(cppify(points_to_sample), k)
        return uncppify(var, (k * (1 + self._num_derivatives), k * (1 + self._num_derivatives)))

    def compute_cholesky_variance_of_points(self, points_to_sample):
        k = points_to_sample.shape[0]
        chol = self._gaussian_process.compute_cholesky_variance_of_points(cppify(points_to_sample), k)
        return uncppify(chol, (k * (1 + self._num_derivatives), k * (1 + self._num_derivatives)))

    def compute_grad_variance_of_points(self, points_to_sample, num_derivatives=-1):
        k = points_to_sample.shape[0]
        num_derivatives = self._clamp_num_derivatives(k, num_derivatives)
        gv = self._gaussian_process.compute_grad_variance_of_points(cppify(points_to_sample), k, num_derivatives)
        m = k * (1 + self._num_derivatives)
        return uncppify(gv, (num_derivatives, m, m, self.dim))

    def compute_grad_cholesky_variance_of_points(self, points_to_sample, num_derivatives=-1):
        k = points_to_sample.shape[0]
        num_derivatives = self._clamp_num_derivatives(k, num_derivatives)
        gc = self._gaussian_process.compute_grad_cholesky_variance_of_points(cppify(points_to_sample), k, num_derivatives)
        m = k * (1 + self._num_derivatives)
        return uncppify(gc, (num_derivatives, m, m, self.dim))

    def add_sampled_points(self, sampled_points):
        prev = self.num_sampled
        self._historical_data.append_sample_points(sampled_points)
        self._gaussian_process.add_sampled_points(
            cppify(self._historical_data.points_sampled[prev:, ...]),
            cppify(self._historical_data.points_sampled_value[prev:]),
            len(sampled_points),
        )

    def sample_point_from_gp(self, point_to_sample, noise_variance=0.0):
        return numpy.array(self._gaussian_process.sample_point_from_gp(cppify(point_to_sample)))


def _default_randomness(randomness, num_threads=1):
    if randomness is not None:
        return randomness
    r = C_GP.RandomnessSourceContainer(num_threads)
    r.SetRandomizedUniformGeneratorSeed(0)
    r.SetRandomizedNormalRNGSeed(0)
    return r


def multistart_expected_improvement_optimization(ei_optimizer, num_multistarts, num_to_sample, use_gpu=False, which_gpu=0,
                                                 randomness=None, max_num_threads=DEFAULT_MAX_NUM_THREADS, status=None):
    """cpp_wrappers/expected_improvement.py:22-105 (num_multistarts is unused there too: the count lives in
    ei_optimizer.optimizer_parameters)."""
    randomness = _default_randomness(randomness, max_num_threads)
    status = {} if status is None else status
    ei = ei_optimizer.objective_function
    best = C_GP.multistart_expected_improvement_optimization(
        ei_optimizer.optimizer_parameters, ei._gaussian_process._gaussian_process,
        [float(x) for x in cppify(ei_optimizer.domain.domain_bounds)], cppify(ei._points_being_sampled), num_to_sample,
        ei.num_being_sampled, ei._best_so_far, ei._num_mc_iterations, max_num_threads, use_gpu, which_gpu, randomness, status)
    return uncppify(best, (num_to_sample, ei.dim))


class ExpectedImprovement(object):
    """cpp_wrappers/expected_improvement.py:109-367 (q,p-EI by Monte Carlo)."""

    def __init__(self, gaussian_process, points_to_sample=None, points_being_sampled=None,
                 num_mc_iterations=DEFAULT_EXPECTED_IMPROVEMENT_MC_ITERATIONS, randomness=None):
        self._num_mc_iterations = num_mc_iterations
        self._gaussian_process = gaussian_process
        if gaussian_process._historical_data.points_sampled_value.size > 0:
            self._best_so_far = numpy.amin(gaussian_process._historical_data.points_sampled_value[:, 0])
        else:
            self._best_so_far = numpy.finfo(numpy.float64).max
        self._points_being_sampled = numpy.array([]) if points_being_sampled is None else numpy.copy(points_being_sampled)
        self._points_to_sample = numpy.zeros((1, gaussian_process.dim)) if points_to_sample is None else points_to_sample
        self._randomness = _default_randomness(randomness)
        self.objective_type = None

    dim = property(lambda self: self._gaussian_process.dim)
    num_to_sample = property(lambda self: self._points_to_sample.shape[0])
    num_being_sampled = property(lambda self: self._points_being_sampled.shape[0])
    problem_size = property(lambda self: self.num_to_sample * self.dim)

    def get_current_point(self):
        return numpy.copy(self._points_to_sample)

    def set_current_point(self, points_to_sample):
        self._points_to_sample = numpy.copy(numpy.atleast_2d(points_to_sample))

    current_point = property(get_current_point, set_current_point)

    def compute_expected_improvement(self, force_monte_carlo=False):
        return C_GP.compute_expected_improvement(
            self._gaussian_process._gaussian_process, cppify(self._points_to_sample), cppify(self._points_being_sampled),
            self.num_to_sample, self.num_being_sampled, self._num_mc_iterations, self._best_so_far, force_monte_carlo,
            self._randomness)

    compute_objective_function = compute_expected_improvement

    def compute_grad_expected_improvement(self, force_monte_carlo=False):
        grad_ei = C_GP.compute_grad_expected_improvement(
            self._gaussian_process._gaussian_process, cppify(self._points_to_sample), cppify(self._points_being_sampled),
            self.num_to_sample, self.num_being_sampled, self._num_mc_iterations, self._best_so_far, force_monte_carlo,
            self._randomness)
        return uncppify(grad_ei, (self.num_to_sample, self.dim))

    compute_grad_objective_function = compute_grad_expected_improvement

    def evaluate_at_point_list(self, points_to_evaluate, randomness=None, max_num_threads=DEFAULT_MAX_NUM_THREADS, status=None):
        randomness = self._randomness if (randomness is None and max_num_threads == 1) else _default_randomness(
            randomness, max_num_threads)
        status = {} if status is None else status
        num_to_evaluate, num_to_sample, _ = points_to_evaluate.shape
        return numpy.array(C_GP.evaluate_EI_at_point_list(
            self._gaussian_process._gaussian_process, cppify(points_to_evaluate), cppify(self._points_being_sampled), num_to_evaluate, num_to_sample, self.num_being_sampled, self._best_so_far,
            self._num_mc_iterations, max_num_threads, randomness, status))


class PosteriorMean(object):
    """cpp_wrappers/knowledge_gradient.py:20-170: -mu(x) with the fidelity coordinates pinned to 1."""

    def __init__(self, gaussian_process, num_fidelity, points_to_sample=None, randomness=None):
        self._gaussian_process = gaussian_process
        self._num_fidelity = num_fidelity
        self._points_to_sample = numpy.zeros((1, gaussian_process.dim)) if points_to_sample is None else points_to_sample
        self._randomness = _default_randomness(randomness)
        self.objective_type = None

    dim = property(lambda self: self._gaussian_process.dim)
    problem_size = property(lambda self: self.dim - self._num_fidelity)

    def get_current_point(self):
        return numpy.copy(self._points_to_sample)

    def set_current_point(self, points_to_sample):
        self._points_to_sample = numpy.copy(numpy.atleast_2d(points_to_sample))

    current_point = property(get_current_point, set_current_point)

    def compute_posterior_mean(self, force_monte_carlo=False):
        return C_GP.compute_posterior_mean(self._gaussian_process._gaussian_process, self._num_fidelity,
                                           cppify(self._points_to_sample))

    compute_objective_function = compute_posterior_mean

    def compute_grad_posterior_mean(self, force_monte_carlo=False):
        grad = C_GP.compute_grad_posterior_mean(self._gaussian_process._gaussian_process, self._num_fidelity,
                                                cppify(self._points_to_sample))
        return uncppify(grad, (1, self.dim - self._num_fidelity))

    compute_grad_objective_function = compute_grad_posterior_mean


class KnowledgeGradient(object):
    """cpp_wrappers/knowledge_gradient.py:309-596 (q-KG / d-KG value and gradient by Monte Carlo)."""

    def __init__(self, gaussian_process, num_fidelity, inner_optimizer, discrete_pts, points_to_sample=None,
                 points_being_sampled=None, num_mc_iterations=DEFAULT_EXPECTED_IMPROVEMENT_MC_ITERATIONS, randomness=None):
        self._num_mc_iterations = num_mc_iterations
        self._gaussian_process = gaussian_process
        self._num_fidelity = num_fidelity
        self._inner_optimizer = inner_optimizer
        self._discrete_pts = numpy.copy(discrete_pts)
        full_points = numpy.zeros((discrete_pts.shape[0], discrete_pts.shape[1] + num_fidelity))
        full_points[:, :discrete_pts.shape[1]] = discrete_pts
        full_points[:, discrete_pts.shape[1]:] = 1.0
        self._mu_star = self._gaussian_process.compute_mean_of_additional_points(full_points)
        self._best_so_far = numpy.amin(self._mu_star)  # knowledge_gradient.py:366-368
        self._points_being_sampled = numpy.array([]) if points_being_sampled is None else numpy.copy(points_being_sampled)
        self._points_to_sample = numpy.zeros((1, gaussian_process.dim)) if points_to_sample is None else points_to_sample
        self._randomness = _default_randomness(randomness)
        self.objective_type = None

    dim = property(lambda self: self._gaussian_process.dim)
    num_to_sample = property(lambda self: self._points_to_sample.shape[0])
    num_being_sampled = property(lambda self: self._points_being_sampled.shape[0])
    discrete = property(lambda self: self._discrete_pts.shape[0])
    problem_size = property(lambda self: self.num_to_sample * self.dim)

    def get_current_point(self):
        return numpy.copy(self._points_to_sample)

    def set_current_point(self, points_to_sample):
        self._points_to_sample = numpy.copy(numpy.atleast_2d(points_to_sample))

    current_point = property(get_current_point, set_current_point)

    def _args(self):
        return (self._gaussian_process._gaussian_process, self._num_fidelity, self._inner_optimizer.optimizer_parameters,
                cppify(self._inner_optimizer.domain.domain_bounds), cppify(self._discrete_pts),
                cppify(self._points_to_sample), cppify(self._points_being_sampled), self.discrete, self.num_to_sample,
                self.num_being_sampled, self._num_mc_iterations, self._best_so_far, self._randomness)

    def compute_knowledge_gradient(self, force_monte_carlo=False):
        return C_GP.compute_knowledge_gradient(*self._args())

    compute_objective_function = compute_knowledge_gradient

    def compute_grad_knowledge_gradient(self, force_monte_carlo=False):
        return uncppify(C_GP.compute_grad_knowledge_gradient(*self._args()), (self.num_to_sample, self.dim))

    compute_grad_objective_function = compute_grad_knowledge_gradient

    def compute_hessian_objective_function(self, **kwargs):
        raise NotImplementedError("Currently we cannot compute the hessian of knowledge gradient.")

    def evaluate_at_point_list(self, points_to_evaluate, randomness=None, max_num_threads=DEFAULT_MAX_NUM_THREADS, status=None):
        randomness = self._randomness if (randomness is None and max_num_threads == 1) else _default_randomness(
            randomness, max_num_threads)
        status = {} if status is None else status
        num_to_evaluate, num_to_sample, _ = points_to_evaluate.shape
        if self.num_being_sampled > 0:
            discrete_being_sampled = numpy.concatenate((self._discrete_pts, self._points_being_sampled))
        else:
            discrete_being_sampled = self._discrete_pts
        return numpy.array(C_GP.evaluate_KG_at_point_list(
            self._gaussian_process._gaussian_process, self._num_fidelity, self._inner_optimizer.optimizer_parameters,
            cppify(self._inner_optimizer.domain.domain_bounds), cppify(discrete_being_sampled), cppify(points_to_evaluate),
            num_to_evaluate, self.discrete, num_to_sample, self.num_being_sampled, self._best_so_far,
            self._num_mc_iterations, max_num_threads, randomness, status))


def multistart_knowledge_gradient_optimization(kg_optimizer, inner_optimizer, num_multistarts, discrete_pts, num_to_sample,
                                               num_pts, randomness=None, max_num_threads=DEFAULT_MAX_NUM_THREADS,
                                               status=None):
    """cpp_wrappers/knowledge_gradient.py:225-307."""
    randomness = _default_randomness(randomness, max_num_threads)
    status = {} if status is None else status
    kg = kg_optimizer.objective_function
    best = C_GP.multistart_knowledge_gradient_optimization(
        kg_optimizer.optimizer_parameters, inner_optimizer.optimizer_parameters, kg._gaussian_process._gaussian_process,
        kg._num_fidelity, cppify(kg_optimizer.domain.domain_bounds), cppify(discrete_pts),
        cppify(kg._points_being_sampled), num_pts, num_to_sample, kg.num_being_sampled, kg._best_so_far,
        kg._num_mc_iterations, max_num_threads, randomness, status)
    return uncppify(best, (num_to_sample, kg.dim))


# ---- MCMC-averaged wrappers (cpp_wrappers/knowledge_gradient_mcmc.py, expected_improvement_mcmc.py) ----
class GaussianProcessMCMC(object):
    """cpp_wrappers/knowledge_gradient_mcmc.py:163-240: one GP per hyper-parameter sample over the same historical data.
    hyperparameters_list [num_mcmc][1 + dim], noise_variance_list [num_mcmc][1 + num_derivatives]."""

    def __init__(self, hyperparameters_list, noise_variance_list, historical_data, derivatives):
        self._hyperparameters_list = copy.deepcopy(numpy.asarray(hyperparameters_list, dtype=numpy.float64))
        self._num_mcmc = self._hyperparameters_list.shape[0]
        self._historical_data = copy.deepcopy(historical_data)
        self._noise_variance_list = copy.deepcopy(numpy.asarray(noise_variance_list, dtype=numpy.float64))
        self._derivatives = copy.deepcopy(derivatives)
        self._num_derivatives = len(cppify(self._derivatives))
        self._gaussian_process_mcmc = C_GP.GaussianProcessMCMC(
            cppify(self._hyperparameters_list), cppify(self._noise_variance_list), cppify(self._historical_data.points_sampled),
            cppify(self._historical_data.points_sampled_value), cppify(self._derivatives), self._num_mcmc, self._num_derivatives,
            self._historical_data.dim, self._historical_data.num_sampled)

    dim = property(lambda self: self._historical_data.dim)
    num_sampled = property(lambda self: self._historical_data.num_sampled)
    num_derivatives = property(lambda self: self._num_derivatives)
    derivatives = property(lambda self: numpy.copy(self._derivatives))
    noise_variance_list = property(lambda self: numpy.copy(self._noise_variance_list))

    def get_historical_data_copy(self):
        return copy.deepcopy(self._historical_data)

    def member_models(self):
        """The per-sample GaussianProcess wrappers the reference keeps beside the MCMC object
        (log_likelihood_mcmc.py:238-262: `models`), built the same way: Matern-5/2 with each sample's hyper-parameters."""
        return [GaussianProcess(SquareExponential(self._hyperparameters_list[i]), self._noise_variance_list[i],
                                self._historical_data, list(self._derivatives)) for i in range(self._num_mcmc)]


class PosteriorMeanMCMC(object):
    """cpp_wrappers/knowledge_gradient_mcmc.py:19-160: the posterior mean averaged over the per-sample GPs."""

    def __init__(self, gaussian_process_list, num_fidelity, points_to_sample=None, randomness=None):
        self._gaussian_process_list = gaussian_process_list
        self._num_fidelity = num_fidelity
        self._points_to_sample = numpy.zeros((1, gaussian_process_list[0].dim)) if points_to_sample is None else points_to_sample
        self._randomness = _default_randomness(randomness)
        self.objective_type = None

    dim = property(lambda self: self._gaussian_process_list[0].dim)
    problem_size = property(lambda self: self.dim - self._num_fidelity)

    def get_current_point(self):
        return numpy.copy(self._points_to_sample)

    def set_current_point(self, points_to_sample):
        self._points_to_sample = numpy.copy(numpy.atleast_2d(points_to_sample))

    current_point = property(get_current_point, set_current_point)

    def compute_posterior_mean_mcmc(self, force_monte_carlo=False):
        total = 0.0
        for gp in self._gaussian_process_list:
            total += C_GP.compute_posterior_mean(gp._gaussian_process, self._num_fidelity, cppify(self._points_to_sample))
        return total / len(self._gaussian_process_list)

    compute_objective_function = compute_posterior_mean_mcmc

    def compute_grad_posterior_mean_mcmc(self, force_monte_carlo=False):
        grad = numpy.zeros((1, self.dim - self._num_fidelity))
        for gp in self._gaussian_process_list:
            grad += uncppify(C_GP.compute_grad_posterior_mean(gp._gaussian_process, self._num_fidelity,
                                                              cppify(self._points_to_sample)), (1, self.dim - self._num_fidelity))
        return grad / len(self._gaussian_process_list)

    compute_grad_objective_function = compute_grad_posterior_mean_mcmc


class KnowledgeGradientMCMC(object):
    """cpp_wrappers/knowledge_gradient_mcmc.py:243-420."""

    def __init__(self, gaussian_process_mcmc, gaussian_process_list, num_fidelity, inner_optimizer, discrete_pts_list,
                 num_to_sample, points_to_sample=None, points_being_sampled=None,
                 num_mc_iterations=DEFAULT_EXPECTED_IMPROVEMENT_MC_ITERATIONS, randomness=None):
        self._num_mc_iterations = num_mc_iterations
        self._gaussian_process_mcmc = gaussian_process_mcmc
        self._gaussian_process_list = gaussian_process_list
        self._num_fidelity = num_fidelity
        self._inner_optimizer = inner_optimizer
        self._discrete_pts_list, self._best_so_far_list = [], []
        for discrete_pts, gp in zip(discrete_pts_list, gaussian_process_list):
            self._discrete_pts_list.append(numpy.copy(discrete_pts))
            full = numpy.ones((discrete_pts.shape[0], gp.dim))
            full[:, :discrete_pts.shape[1]] = discrete_pts
            self._best_so_far_list.append(numpy.amin(gp.compute_mean_of_additional_points(full)))  # :268-274
        self._points_being_sampled = numpy.array([]) if points_being_sampled is None else numpy.copy(points_being_sampled)
        self._points_to_sample = (numpy.zeros((num_to_sample, gaussian_process_mcmc.dim)) if points_to_sample is None
                                  else points_to_sample)
        self._randomness = _default_randomness(randomness)
        self.objective_type = None

    dim = property(lambda self: self._gaussian_process_mcmc.dim)
    num_to_sample = property(lambda self: self._points_to_sample.shape[0])
    num_being_sampled = property(lambda self: self._points_being_sampled.shape[0])
    discrete = property(lambda self: self._discrete_pts_list[0].shape[0])
    problem_size = property(lambda self: self.num_to_sample * self.dim)

    def get_current_point(self):
        return numpy.copy(self._points_to_sample)

    def set_current_point(self, points_to_sample):
        self._points_to_sample = numpy.copy(numpy.atleast_2d(points_to_sample))

    current_point = property(get_current_point, set_current_point)

    def _args(self):
        return (self._gaussian_process_mcmc._gaussian_process_mcmc, self._num_fidelity, self._inner_optimizer.optimizer_parameters,
                [float(x) for x in cppify(self._inner_optimizer.domain.domain_bounds)], cppify(numpy.array(self._discrete_pts_list)),
                cppify(self._points_to_sample), cppify(self._points_being_sampled), self.discrete, self.num_to_sample,
                self.num_being_sampled, self._num_mc_iterations, cppify(numpy.array(self._best_so_far_list)), self._randomness)

    def compute_knowledge_gradient_mcmc(self, force_monte_carlo=False):
        return C_GP.compute_knowledge_gradient_mcmc(*self._args())

    compute_objective_function = compute_knowledge_gradient_mcmc

    def compute_grad_knowledge_gradient_mcmc(self, force_monte_carlo=False):
        return uncppify(C_GP.compute_grad_knowledge_gradient_mcmc(*self._args()), (self.num_to_sample, self.dim))

    compute_grad_objective_function = compute_grad_knowledge_gradient_mcmc

    def evaluate_at_point_list(self, points_to_evaluate, randomness=None, max_num_threads=DEFAULT_MAX_NUM_THREADS, status=None):
        """:292-310, with the two list arguments in the order the C++ wrapper declares them (the reference's Python passes
        them swapped; see GPP.evaluate_KG_mcmc_at_point_list)."""
        randomness = self._randomness if (randomness is None and max_num_threads == 1) else _default_randomness(
            randomness, max_num_threads)
        status = {} if status is None else status
        num_to_evaluate, num_to_sample, _ = points_to_evaluate.shape
        packed = numpy.concatenate((numpy.ravel(self._discrete_pts_list), numpy.ravel(self._points_being_sampled)))
        bounds = [float(x) for x in cppify(self._inner_optimizer.domain.domain_bounds)]
        return numpy.array(C_GP.evaluate_KG_mcmc_at_point_list(
            self._gaussian_process_mcmc._gaussian_process_mcmc, self._num_fidelity, self._inner_optimizer.optimizer_parameters,
            bounds, cppify(points_to_evaluate), cppify(packed), num_to_evaluate, self.discrete, num_to_sample,
            self.num_being_sampled, cppify(numpy.array(self._best_so_far_list)), self._num_mc_iterations, max_num_threads,
            randomness, status))


def multistart_knowledge_gradient_mcmc_optimization(kg_optimizer, inner_optimizer, num_multistarts, discrete_pts, num_to_sample,
                                                    num_pts, randomness=None, max_num_threads=DEFAULT_MAX_NUM_THREADS,
                                                    status=None):
    """cpp_wrappers/knowledge_gradient_mcmc.py:200-240."""
    randomness = _default_randomness(randomness, max_num_threads)
    status = {} if status is None else status
    kg = kg_optimizer.objective_function
    best = C_GP.multistart_knowledge_gradient_mcmc_optimization(
        kg_optimizer.optimizer_parameters, inner_optimizer.optimizer_parameters, kg._gaussian_process_mcmc._gaussian_process_mcmc,
        kg._num_fidelity, [float(x) for x in cppify(kg_optimizer.domain.domain_bounds)], cppify(numpy.array(discrete_pts)),
        cppify(kg._points_being_sampled), num_pts, num_to_sample, kg.num_being_sampled,
        cppify(numpy.array(kg._best_so_far_list)), kg._num_mc_iterations, max_num_threads, randomness, status)
    return uncppify(best, (num_to_sample, kg.dim))


class ExpectedImprovementMCMC(object):
    """cpp_wrappers/expected_improvement_mcmc.py:60-260."""

    def __init__(self, gaussian_process_mcmc, num_to_sample, points_to_sample=None, points_being_sampled=None,
                 num_mc_iterations=DEFAULT_EXPECTED_IMPROVEMENT_MC_ITERATIONS, randomness=None):
        self._num_mc_iterations = num_mc_iterations
        self._gaussian_process_mcmc = gaussian_process_mcmc
        values = gaussian_process_mcmc._historical_data.points_sampled_value
        best = numpy.amin(values[:, 0]) if values.size > 0 else numpy.finfo(numpy.float64).max
        self._best_so_far_list = gaussian_process_mcmc._num_mcmc * [best]
        self._points_being_sampled = numpy.array([]) if points_being_sampled is None else numpy.copy(points_being_sampled)
        self._points_to_sample = (numpy.zeros((num_to_sample, gaussian_process_mcmc.dim)) if points_to_sample is None
                                  else numpy.copy(numpy.atleast_2d(points_to_sample)))
        self._randomness = _default_randomness(randomness)
        self.objective_type = None

    dim = property(lambda self: self._gaussian_process_mcmc.dim)
    num_to_sample = property(lambda self: self._points_to_sample.shape[0])
    num_being_sampled = property(lambda self: self._points_being_sampled.shape[0])
    problem_size = property(lambda self: self.num_to_sample * self.dim)

    def get_current_point(self):
        return numpy.copy(self._points_to_sample)

    def set_current_point(self, points_to_sample):
        self._points_to_sample = numpy.copy(numpy.atleast_2d(points_to_sample))

    current_point = property(get_current_point, set_current_point)

    def _args(self):
        return (self._gaussian_process_mcmc._gaussian_process_mcmc, cppify(self._points_to_sample),
                cppify(self._points_being_sampled), self.num_to_sample, self.num_being_sampled, self._num_mc_iterations,
                cppify(numpy.array(self._best_so_far_list)), self._randomness)

    def compute_expected_improvement(self, force_monte_carlo=False):
        return C_GP.compute_expected_improvement_mcmc(*self._args())

    compute_objective_function = compute_expected_improvement

    def compute_grad_expected_improvement(self, force_monte_carlo=False):
        return uncppify(C_GP.compute_grad_expected_improvement_mcmc(*self._args()), (self.num_to_sample, self.dim))

    compute_grad_objective_function = compute_grad_expected_improvement

    def evaluate_at_point_list(self, points_to_evaluate, randomness=None, max_num_threads=DEFAULT_MAX_NUM_THREADS, status=None):
        randomness = self._randomness if (randomness is None and max_num_threads == 1) else _default_randomness(
            randomness, max_num_threads)
        status = {} if status is None else status
        num_to_evaluate, num_to_sample, _ = points_to_evaluate.shape
        return numpy.array(C_GP.evaluate_EI_mcmc_at_point_list(
            self._gaussian_process_mcmc._gaussian_process_mcmc, cppify(points_to_evaluate), cppify(self._points_being_sampled),
            num_to_evaluate, num_to_sample, self.num_being_sampled, cppify(numpy.array(self._best_so_far_list)),
            self._num_mc_iterations, max_num_threads, randomness, status))


def multistart_expected_improvement_mcmc_optimization(ei_optimizer, num_multistarts, num_to_sample, randomness=None,
                                                      max_num_threads=DEFAULT_MAX_NUM_THREADS, status=None):
    """cpp_wrappers/expected_improvement_mcmc.py:22-57."""
    randomness = _default_randomness(randomness, max_num_threads)
    status = {} if status is None else status
    ei = ei_optimizer.objective_function
    best = C_GP.multistart_expected_improvement_mcmc_optimization(
        ei_optimizer.optimizer_parameters, ei._gaussian_process_mcmc._gaussian_process_mcmc,
        [float(x) for x in cppify(ei_optimizer.domain.domain_bounds)], cppify(ei._points_being_sampled), num_to_sample,
        ei.num_being_sampled, cppify(numpy.array(ei._best_so_far_list)), ei._num_mc_iterations, max_num_threads, randomness, status)
    return uncppify(best, (num_to_sample, ei.dim))


# ---- log likelihood (py/cpp_wrappers/log_likelihood.py) ----
class GaussianProcessLogLikelihood(object):
    """Log marginal likelihood of the historical data as a function of the hyper-parameters (covariance hyper-parameters
    followed by the noise variances), the object the reference's hyper-parameter optimisers and its emcee driver evaluate
    (log_likelihood.py:230-400).  compute_log_likelihood / compute_grad_log_likelihood go to the device
    (GPP.compute_log_likelihood / compute_hyperparameter_grad_log_likelihood -> moe_ll_evaluate / moe_ll_grad)."""

    def __init__(self, covariance_function, historical_data, noise_variance, derivatives,
                 log_likelihood_type=C_GP.LogLikelihoodTypes.log_marginal_likelihood):
        self._covariance = copy.deepcopy(covariance_function)
        self._historical_data = copy.deepcopy(historical_data)
        self._noise_variance = numpy.array(noise_variance, dtype=float, copy=True).ravel()
        self._derivatives = [int(v) for v in derivatives]
        self._num_derivatives = len(self._derivatives)
        self.objective_type = log_likelihood_type

    dim = property(lambda self: self._historical_data.dim)
    num_hyperparameters = property(lambda self: self._covariance.num_hyperparameters + self._noise_variance.size)
    problem_size = num_hyperparameters
    cov_hyperparameters = property(lambda self: self._covariance.hyperparameters)
    noise_variance = property(lambda self: self._noise_variance)
    derivatives = property(lambda self: self._derivatives)
    num_derivatives = property(lambda self: self._num_derivatives)
    _num_sampled = property(lambda self: self._historical_data.num_sampled)
    _points_sampled = property(lambda self: self._historical_data.points_sampled)
    _points_sampled_value = property(lambda self: self._historical_data.points_sampled_value)
    _points_sampled_noise_variance = property(lambda self: self._historical_data.points_sampled_noise_variance)

    def get_hyperparameters(self):
        return numpy.append(self._covariance.hyperparameters, self._noise_variance)

    def set_hyperparameters(self, hyperparameters):
        k = self._covariance.num_hyperparameters
        self._covariance.hyperparameters = hyperparameters[:k]
        self._noise_variance = numpy.array(hyperparameters[k:], dtype=float).ravel()

    hyperparameters = property(get_hyperparameters, set_hyperparameters)
    current_point = hyperparameters

    def get_covariance_copy(self):
        return copy.deepcopy(self._covariance)

    def get_historical_data_copy(self):
        return copy.deepcopy(self._historical_data)

    def _args(self):
        return (cppify(self._points_sampled), cppify(self._points_sampled_value), self.dim, self._num_sampled, self.objective_type,
                cppify_hyperparameters(self.cov_hyperparameters), cppify(self._derivatives), self._num_derivatives,
                cppify(self.noise_variance))

    def compute_log_likelihood(self):
        return C_GP.compute_log_likelihood(*self._args())

    compute_objective_function = compute_log_likelihood

    def compute_grad_log_likelihood(self):
        return numpy.array(C_GP.compute_hyperparameter_grad_log_likelihood(*self._args()))

    compute_grad_objective_function = compute_grad_log_likelihood


class GaussianProcessLogMarginalLikelihood(GaussianProcessLogLikelihood):
    """(log_likelihood.py:403-440)"""

    def __init__(self, covariance_function, historical_data, noise_variance, derivatives):
        super(GaussianProcessLogMarginalLikelihood, self).__init__(covariance_function, historical_data, noise_variance, derivatives,
                                                                   C_GP.LogLikelihoodTypes.log_marginal_likelihood)


def evaluate_log_likelihood_at_hyperparameter_list(log_likelihood_evaluator, hyperparameters_to_evaluate, max_num_threads=4,
                                                   status=None):
    """log_likelihood.py:179-227: the log likelihood at each row of hyperparameters_to_evaluate
    [num_to_eval][num_hyperparameters]; the rows of one call are factorised together on the device."""
    if status is None:
        status = {}
    h = numpy.asarray(hyperparameters_to_evaluate, dtype=float)
    ev = log_likelihood_evaluator
    return numpy.array(C_GP.evaluate_log_likelihood_at_hyperparameter_list(
        cppify(h), cppify(ev._points_sampled), cppify(ev._points_sampled_value), ev.dim, ev._num_sampled, ev.objective_type,
        cppify_hyperparameters(ev.cov_hyperparameters), cppify(ev.noise_variance), cppify(ev.derivatives), ev.num_derivatives,
        h.shape[0], max_num_threads, status))

