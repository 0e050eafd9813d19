"""Drop-in for the reference's ``moe.build.GPP`` extension module (BOOST_PYTHON_MODULE(GPP), gpp_python.cpp:453-600),
restricted to the GP-posterior + Monte-Carlo acquisition hot path and backed by libmoe_hip.so (include/moe_hip.h).

Every name below carries the Python-visible signature of the boost::python export it replaces, so
``moe/optimal_learning/python/cpp_wrappers/*.py`` can run unchanged on top of it after

    import cornell_moe_amd.GPP, sys; sys.modules["moe.build.GPP"] = cornell_moe_amd.GPP

(INTEGRATION.md).  Data conventions are the reference's (gpp_python_common.cpp:52-129): inputs are flat Python lists of
float (row-major ``[point][dim]``), sizes are explicit ints, outputs are new Python lists.

Randomness: the reference seeds boost::mt19937 + boost::normal_distribution, whose draw algorithm is Boost-version
dependent and pinned by no reference test.  Here a RandomnessSourceContainer seeds ``moe_normal_draws``
(mt19937 + Box-Muller); the *semantics* are kept -- thread i is seeded ``seed + i`` (gpp_python_common.cpp:152-156), and
every evaluation first rewinds to the most recent seed (gpp_knowledge_gradient_optimization.cpp:164, gpp_math.cpp:2076),
so repeated calls use common random numbers -- but draw-for-draw equality with a given Boost is not claimed.
There is no CPU fallback: every compute entry point needs a visible gfx950 device.
"""
import os
import time

import numpy as np

from . import _lib
from . import api as _api
from .api import (BoundsException, InvalidValueException, OptimalLearningException,  # noqa: F401  (module-level exports)
                  SingularMatrixException)


# ---- enums (gpp_python_common.cpp:200-241) ----
class _Enum(object):
    def __init__(self, name, value):
        self.name, self.value = name, value

    def __repr__(self):
        return "GPP.%s" % self.name

    def __int__(self):
        return self.value


class OptimizerTypes(object):
    null = _Enum("OptimizerTypes.null", 0)
    gradient_descent = _Enum("OptimizerTypes.gradient_descent", 1)
    newton = _Enum("OptimizerTypes.newton", 2)


class DomainTypes(object):
    tensor_product = _Enum("DomainTypes.tensor_product", 0)
    simplex = _Enum("DomainTypes.simplex", 1)


def _check_domain_type(optimizer_parameters, kg=False):
    """The dispatch of gpp_python_knowledge_gradient.cpp:279-296, 327-341 / gpp_python_expected_improvement.cpp:262-271: tensor
    product or its intersection with the unit simplex; anything else is the reference's "invalid domain choice".  For KG the
    reference builds the INNER domain of every sample's posterior-mean optimisation of the same type; the MC kernels' line search
    takes it (r4: csrc/kg_mc.hpp simplex_limit)."""
    if int(optimizer_parameters.domain_type) not in (int(DomainTypes.tensor_product), int(DomainTypes.simplex)):
        raise OptimalLearningException("ERROR: invalid domain choice. Setting all coordinates to 0.0.")


def _domain_name(optimizer_parameters):
    """DomainType::kName (gpp_domain.hpp:76, 229): part of the status keys the wrappers fill."""
    return "simplex_tensor_product" if int(optimizer_parameters.domain_type) == int(DomainTypes.simplex) else "tensor_product"


class LogLikelihoodTypes(object):
    log_marginal_likelihood = _Enum("LogLikelihoodTypes.log_marginal_likelihood", 0)
    leave_one_out_log_likelihood = _Enum("LogLikelihoodTypes.leave_one_out_log_likelihood", 1)


# ---- optimizer parameter structs (gpp_python_common.cpp:243-301, gpp_optimizer_parameters.hpp:81-133) ----
class GradientDescentParameters(object):
    def __init__(self, num_multistarts, max_num_steps, max_num_restarts, num_steps_averaged, gamma, pre_mult,
                 max_relative_change, tolerance):
        self.num_multistarts = int(num_multistarts)
        self.max_num_steps = int(max_num_steps)
        self.max_num_restarts = int(max_num_restarts)
        self.num_steps_averaged = int(num_steps_averaged)
        self.gamma = float(gamma)
        self.pre_mult = float(pre_mult)
        self.max_relative_change = float(max_relative_change)
        self.tolerance = float(tolerance)

    def _as_tuple(self):
        return (self.num_multistarts, self.max_num_steps, self.max_num_restarts, self.num_steps_averaged, self.gamma,
                self.pre_mult, self.max_relative_change, self.tolerance)


class NewtonParameters(object):
    """Field container only (gpp_optimizer_parameters.hpp:186-243); Newton drives hyperparameter optimisation, which is
    outside the hot path."""

    def __init__(self, num_multistarts, max_num_steps, gamma, time_factor, max_relative_change, tolerance):
        self.num_multistarts = int(num_multistarts)
        self.max_num_steps = int(max_num_steps)
        self.gamma = float(gamma)
        self.time_factor = float(time_factor)
        self.max_relative_change = float(max_relative_change)
        self.tolerance = float(tolerance)


# ---- randomness (gpp_python_common.cpp:131-198, gpp_random.hpp:54-303) ----
def _randomized_seed(base_seed, thread_id):
    # NormalRNG::SetRandomizedSeed mixes the seed with time and the thread id (gpp_random.cpp:86-107)
    mix = int(time.time() * 1e6) ^ (os.getpid() << 16) ^ int.from_bytes(os.urandom(4), "little")
    return (int(base_seed) + 0x9E3779B9 * (int(thread_id) + 1) + mix) & 0xFFFFFFFF


class _NormalStream(object):
    """NormalRNG stand-in: remembers its last seed; ``table(count)`` is the stream replayed from that seed."""

    def __init__(self, seed):
        self.last_seed = int(seed) & 0xFFFFFFFF
        self._cache = None

    def set_seed(self, seed):
        self.last_seed = int(seed) & 0xFFFFFFFF
        self._cache = None

    def table(self, count):
        if self._cache is None or self._cache.size < count:
            self._cache = _api.normal_draws(self.last_seed, max(int(count), 1))
        return self._cache[:count]


class RandomnessSourceContainer(object):
    kNormalDefaultSeed = 314  # gpp_python_common.hpp:147-148
    kUniformDefaultSeed = 314

    def __init__(self, num_threads=1):
        self.num_normal_rng = int(num_threads)
        self.uniform_seed = self.kUniformDefaultSeed
        self.normal_rng_vec = [_NormalStream(self.kNormalDefaultSeed + i) for i in range(self.num_normal_rng)]
        self._uniform = np.random.RandomState(self.uniform_seed)

    def SetExplicitUniformGeneratorSeed(self, seed):
        self.uniform_seed = int(seed) & 0xFFFFFFFF
        self._uniform = np.random.RandomState(self.uniform_seed)

    def SetRandomizedUniformGeneratorSeed(self, seed):
        self.SetExplicitUniformGeneratorSeed(_randomized_seed(seed, 0))

    def ResetUniformRNGSeed(self):
        self._uniform = np.random.RandomState(self.uniform_seed)

    def SetExplicitNormalRNGSeed(self, seed):
        for i, rng in enumerate(self.normal_rng_vec):
            rng.set_seed(int(seed) + i)

    def SetRandomizedNormalRNGSeed(self, seed):
        for i, rng in enumerate(self.normal_rng_vec):
            rng.set_seed(_randomized_seed(seed, i))

    def SetNormalRNGSeedPythonList(self, seed_list, seed_flag_list):
        if len(seed_list) != len(seed_flag_list) or len(seed_list) != self.num_normal_rng:
            return False
        for i, (seed, flag) in enumerate(zip(seed_list, seed_flag_list)):
            if int(flag):
                self.normal_rng_vec[i].set_seed(int(seed))
        return True

    def ResetNormalRNGSeed(self):
        pass  # streams are replayed from their last seed on every evaluation already

    def PrintState(self):
        print("Uniform:\n  seed %d" % self.uniform_seed)
        for i, rng in enumerate(self.normal_rng_vec):
            print("NormalRNG %d:\n  last seed %d" % (i, rng.last_seed))

    # used by the multistart drivers (Latin-hypercube start generation, gpp_random.cpp:173-194)
    def _uniform_random(self, size):
        return self._uniform.uniform(0.0, 1.0, size=size)

    def _next_uniform_seed(self):
        """A fresh mt19937 seed per Latin-hypercube draw, itself a deterministic function of the uniform generator's
        seed and the number of draws so far (the reference advances ONE engine across draws)."""
        return int(self._uniform.randint(0, 2 ** 31 - 1))


def _flat(values, count=None):
    a = np.ascontiguousarray(np.asarray(values, dtype=np.float64).ravel())
    if count is not None:
        if a.size < count:
            raise InvalidValueException("input list shorter than the sizes passed with it", a.size, count, 0)
        a = a[:count]
    return a


# ---- GaussianProcess (gpp_python_gaussian_process.cpp:42-62, 294-465) ----
class GaussianProcess(object):
    """``GaussianProcess(hyperparameters=[alpha, [lengths]], points_sampled, points_sampled_value, noise_variance,
    derivatives, num_derivatives, dim, num_sampled)``.  Like the reference (``MaternNu2p5 sqexp(...)``,
    gpp_python_gaussian_process.cpp:53) the kernel is Matern-5/2 whatever the Python covariance class is called;
    ``cov_type`` (keyword only, not in the reference) lets tests select the square exponential."""

    def __init__(self, hyperparameters, points_sampled, points_sampled_value, noise_variance, derivatives, num_derivatives,
                 dim, num_sampled, cov_type=_lib.COV_MATERN_NU_2P5, device=None):
        alpha = float(hyperparameters[0])
        lengths = _flat(hyperparameters[1], dim)
        self.dim = int(dim)
        self._g = int(num_derivatives)
        X = _flat(points_sampled, dim * num_sampled).reshape(num_sampled, dim)
        y = _flat(points_sampled_value, num_sampled * (1 + self._g)).reshape(num_sampled, 1 + self._g)
        noise = _flat(noise_variance, 1 + self._g)
        derivs = [int(v) for v in list(derivatives)[:self._g]]
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0")) % max(_lib.device_count(), 1)
        self._dev = _api.DeviceGP(np.concatenate([[alpha], lengths]), X, y, noise, derivs, cov_type=cov_type, device=device)
        self._X, self._y = X.copy(), y.copy()
        self._seed = 0
        self.set_randomized_seed(0)  # make_gaussian_process calls SetRandomizedSeed(0) (:60)

    @property
    def num_sampled(self):
        return self._dev.n

    def _pts(self, points, num):
        return _flat(points, self.dim * num).reshape(num, self.dim)

    def compute_mean_of_points(self, points_to_sample, num_to_sample):
        return list(self._dev.mean(self._pts(points_to_sample, num_to_sample)))

    def compute_mean_of_additional_points(self, discrete_pts, num_pts):
        return list(self._dev.additional_mean(self._pts(discrete_pts, num_pts)))

    def compute_grad_mean_of_points(self, points_to_sample, num_to_sample):
        return list(self._dev.grad_mean(self._pts(points_to_sample, num_to_sample)))

    def compute_variance_of_points(self, points_to_sample, num_to_sample):
        m = num_to_sample * (1 + self._g)
        var = self._dev.variance(self._pts(points_to_sample, num_to_sample)).reshape(m, m)  # [col][row]; symmetric
        low = np.tril(var.T)  # [row][col], lower triangle
        return list((low + np.tril(low, -1).T).ravel())  # lower copied to upper, emitted row-major (:136-151)

    def compute_cholesky_variance_of_points(self, points_to_sample, num_to_sample):
        m = num_to_sample * (1 + self._g)
        chol = self._dev.cholesky_variance(self._pts(points_to_sample, num_to_sample))  # flat col-major, upper = leftovers
        # ZeroUpperTriangle(num_to_sample, ...) -- the reference passes num_to_sample, not m (:177); reproduced literally
        k = num_to_sample
        for j in range(1, k):
            chol[j * k:j * k + j] = 0.0
        return list(chol.reshape(m, m).T.ravel())  # emitted row-major (:178-185)

    def compute_grad_variance_of_points(self, points_to_sample, num_to_sample, num_derivatives):
        return list(self._dev.grad_variance(self._pts(points_to_sample, num_to_sample), num_derivatives))

    def compute_grad_cholesky_variance_of_points(self, points_to_sample, num_to_sample, num_derivatives):
        return list(self._dev.grad_cholesky_variance(self._pts(points_to_sample, num_to_sample), num_derivatives))

    def add_sampled_points(self, new_points, new_points_value, num_new_points):
        pts = self._pts(new_points, num_new_points)
        vals = _flat(new_points_value, num_new_points * (1 + self._g)).reshape(num_new_points, 1 + self._g)
        self._dev.add_points(pts, vals)
        self._X, self._y = np.vstack([self._X, pts]), np.vstack([self._y, vals])

    def sample_point_from_gp(self, point_to_sample):
        """SamplePointFromGP (gpp_math.cpp:1760-1797): mean + chol(Var) w, one draw of 1 + g normals from the GP's own
        stream (which, unlike the acquisition streams, is NOT rewound between calls)."""
        pt = self._pts(point_to_sample, 1)
        g1 = 1 + self._g
        w = _api.normal_draws(self._seed, self._drawn + g1)[self._drawn:]
        self._drawn += g1
        mean = np.zeros(g1)
        mean[0] = self._dev.mean(pt)[0]
        if self._g:
            # derivative rows of the posterior mean come from the grad-mean of the value row at the observed dims
            gm = self._dev.grad_mean(pt).reshape(g1, self.dim)
            for a, dd in enumerate(self._dev.derivatives):
                mean[1 + a] = gm[0, dd]
        L = np.tril(self._dev.cholesky_variance(pt).reshape(g1, g1).T)
        return list(mean + L @ w)

    def sample_global_optima(self, num_optima, inner_number, domain_bounds):
        raise OptimalLearningException("sample_global_optima (PES support code) is outside the hot path (SURVEY 8, out of scope)")

    def set_explicit_seed(self, seed):
        self._seed, self._drawn = int(seed) & 0xFFFFFFFF, 0

    def set_randomized_seed(self, seed):
        self.set_explicit_seed(_randomized_seed(seed, 0))

    def reset_to_most_recent_seed(self):
        self._drawn = 0

    def print_historical_data(self):
        print(self._X)
        print(self._y[:, 0])


def _gd_params(optimizer_parameters):
    gd = optimizer_parameters.optimizer_parameters  # duck-typed _CppOptimizerParameters (gpp_python_knowledge_gradient.cpp:100)
    if hasattr(gd, "_as_tuple"):
        return gd._as_tuple()
    return (gd.num_multistarts, gd.max_num_steps, gd.max_num_restarts, gd.num_steps_averaged, gd.gamma, gd.pre_mult,
            gd.max_relative_change, gd.tolerance)


def _being_sampled(gp, points_being_sampled, num_being_sampled):
    if num_being_sampled <= 0:
        return None
    return _flat(points_being_sampled, gp.dim * num_being_sampled).reshape(num_being_sampled, gp.dim)


# ---- posterior mean (gpp_python_knowledge_gradient.cpp:44-72) ----
def compute_posterior_mean(gaussian_process, num_fidelity, points_to_sample):
    return gaussian_process._dev.posterior_mean(_flat(points_to_sample, gaussian_process.dim - num_fidelity), num_fidelity,
                                                want_grad=False)[0]


def compute_grad_posterior_mean(gaussian_process, num_fidelity, points_to_sample):
    return list(gaussian_process._dev.posterior_mean(_flat(points_to_sample, gaussian_process.dim - num_fidelity),
                                                     num_fidelity, want_grad=True)[1])


# ---- q,p-EI (gpp_python_expected_improvement.cpp:44-109) ----
def _ei(gaussian_process, points_to_sample, points_being_sampled, num_to_sample, num_being_sampled, max_int_steps,
        best_so_far, randomness_source, want_grad):
    gp = gaussian_process
    Xq = _flat(points_to_sample, gp.dim * num_to_sample).reshape(num_to_sample, gp.dim)
    Xp = _being_sampled(gp, points_being_sampled, num_being_sampled)
    u = num_to_sample + max(num_being_sampled, 0)
    normals = randomness_source.normal_rng_vec[0].table(int(max_int_steps) * u)
    return gp._dev.ei(Xq, Xp, int(max_int_steps), float(best_so_far), normals, want_grad=want_grad, want_value=not want_grad)


def compute_expected_improvement(gaussian_process, points_to_sample, points_being_sampled, num_to_sample, num_being_sampled,
                                 max_int_steps, best_so_far, force_monte_carlo, randomness_source):
    return _ei(gaussian_process, points_to_sample, points_being_sampled, num_to_sample, num_being_sampled, max_int_steps,
               best_so_far, randomness_source, False)[0]


def compute_grad_expected_improvement(gaussian_process, points_to_sample, points_being_sampled, num_to_sample,
                                      num_being_sampled, max_int_steps, best_so_far, force_monte_carlo, randomness_source):
    return list(_ei(gaussian_process, points_to_sample, points_being_sampled, num_to_sample, num_being_sampled,
                    max_int_steps, best_so_far, randomness_source, True)[1].ravel())


def evaluate_EI_at_point_list(gaussian_process, initial_guesses, points_being_sampled, num_multistarts, num_to_sample,
                              num_being_sampled, best_so_far, max_int_steps, max_num_threads, randomness_source, status):
    """EvaluateEIAtPointListWrapper (gpp_python_expected_improvement.cpp:401-440) -> EvaluateEIAtPointList
    (gpp_math.cpp:2305-2356)."""
    if max_num_threads > randomness_source.num_normal_rng:
        raise BoundsException("Fewer randomness_sources than max_num_threads.", randomness_source.num_normal_rng,
                              max_num_threads, 1e9)
    gp = gaussian_process
    guesses = _flat(initial_guesses, gp.dim * num_to_sample * num_multistarts).reshape(num_multistarts, num_to_sample, gp.dim)
    Xp = _being_sampled(gp, points_being_sampled, num_being_sampled)
    u = num_to_sample + max(num_being_sampled, 0)
    normals = randomness_source.normal_rng_vec[0].table(int(max_int_steps) * u)
    if num_to_sample == 1 and num_being_sampled <= 0:  # analytic evaluator (gpp_math.cpp:2317-2335)
        out = list(gp._dev.ei_analytic_batch(guesses.reshape(num_multistarts, gp.dim), float(best_so_far), want_grad=False)[0])
    else:
        out = list(gp._dev.ei_batch(guesses, Xp, int(max_int_steps), float(best_so_far), normals, want_grad=False)[0])
    status["evaluate_EI_at_point_list"] = bool(len(out) > 0 and max(out) > 0.0)
    return out


def multistart_expected_improvement_optimization(optimizer_parameters, gaussian_process, domain_bounds, points_being_sampled,
                                                 num_to_sample, num_being_sampled, best_so_far, max_int_steps,
                                                 max_num_threads, use_gpu, which_gpu, randomness_source, status):
    """MultistartExpectedImprovementOptimizationWrapper (gpp_python_expected_improvement.cpp:221-276).  Tensor-product
    domain; every live restart's EI gradient is evaluated in one batched device pass per step.  ``use_gpu`` / ``which_gpu``
    are accepted for signature compatibility (this backend always runs on its GP's device)."""
    from . import multistart
    if max_num_threads > randomness_source.num_normal_rng:
        raise BoundsException("Fewer randomness_sources than max_num_threads.", randomness_source.num_normal_rng,
                              max_num_threads, 1e9)
    _check_domain_type(optimizer_parameters)
    if int(optimizer_parameters.optimizer_type) not in (int(OptimizerTypes.null), int(OptimizerTypes.gradient_descent)):
        raise OptimalLearningException("ERROR: invalid optimizer choice. Setting all coordinates to 0.0.")
    gp = gaussian_process
    Xp = _being_sampled(gp, points_being_sampled, num_being_sampled)
    best, found = multistart.ei_optimal_points(gp._dev, optimizer_parameters, _flat(domain_bounds, 2 * gp.dim), Xp,
                                               int(num_to_sample), float(best_so_far), int(max_int_steps), randomness_source)
    kind = "gradient_descent" if int(optimizer_parameters.optimizer_type) == int(OptimizerTypes.gradient_descent) else "lhc"
    status["%s_%s_domain_found_update" % (kind, _domain_name(optimizer_parameters))] = bool(found)
    return list(np.asarray(best).ravel())


# ---- q-KG / d-KG (gpp_python_knowledge_gradient.cpp:74-154) ----
def _kg(gaussian_process, num_fidelity, optimizer_parameters, domain_bounds, discrete_pts, points_to_sample,
        points_being_sampled, num_pts, num_to_sample, num_being_sampled, max_int_steps, best_so_far, randomness_source,
        want_grad):
    gp = gaussian_process
    size = gp.dim - num_fidelity
    Xq = _flat(points_to_sample, gp.dim * num_to_sample).reshape(num_to_sample, gp.dim)
    Xp = _being_sampled(gp, points_being_sampled, num_being_sampled)
    discrete = _flat(discrete_pts, size * num_pts).reshape(num_pts, size)
    bounds = _flat(domain_bounds, 2 * size)
    m = (num_to_sample + max(num_being_sampled, 0)) * (1 + gp._g)
    M = int(max_int_steps)
    normals = randomness_source.normal_rng_vec[0].table(((M + 1) // 2) * m)
    return gp._dev.kg(_gd_params(optimizer_parameters), bounds, discrete, Xq, Xp, M, float(best_so_far), normals,
                      want_grad=want_grad, num_fidelity=int(num_fidelity))


def compute_knowledge_gradient(gaussian_process, num_fidelity, optimizer_parameters, domain_bounds, discrete_pts,
                               points_to_sample, points_being_sampled, num_pts, num_to_sample, num_being_sampled,
                               max_int_steps, best_so_far, randomness_source):
    return _kg(gaussian_process, num_fidelity, optimizer_parameters, domain_bounds, discrete_pts, points_to_sample,
               points_being_sampled, num_pts, num_to_sample, num_being_sampled, max_int_steps, best_so_far,
               randomness_source, False)["kg"]


def compute_grad_knowledge_gradient(gaussian_process, num_fidelity, optimizer_parameters, domain_bounds, discrete_pts,
                                    points_to_sample, points_being_sampled, num_pts, num_to_sample, num_being_sampled,
                                    max_int_steps, best_so_far, randomness_source):
    return list(_kg(gaussian_process, num_fidelity, optimizer_parameters, domain_bounds, discrete_pts, points_to_sample,
                    points_being_sampled, num_pts, num_to_sample, num_being_sampled, max_int_steps, best_so_far,
                    randomness_source, True)["grad"].ravel())


def evaluate_KG_at_point_list(gaussian_process, num_fidelity, optimizer_parameters, domain_bounds, discrete_being_sampled,
                              initial_guesses, num_multistarts, num_pts, num_to_sample, num_being_sampled, best_so_far,
                              max_int_steps, max_num_threads, randomness_source, status):
    """EvaluateKGAtPointList (gpp_knowledge_gradient_optimization.hpp:1090-1141) via
    gpp_python_knowledge_gradient.cpp:344-397: ``discrete_being_sampled`` = discrete points followed by the points being
    sampled, read exactly as the reference reads it (discrete block = first num_pts*(dim - num_fidelity) entries,
    points_being_sampled at offset dim*num_pts)."""
    if max_num_threads > randomness_source.num_normal_rng:
        raise BoundsException("Fewer randomness_sources than max_num_threads.", randomness_source.num_normal_rng,
                              max_num_threads, 1e9)
    gp = gaussian_process
    size = gp.dim - num_fidelity
    flat = _flat(discrete_being_sampled)
    discrete = flat[:num_pts * size].reshape(num_pts, size)
    Xp = flat[gp.dim * num_pts:gp.dim * (num_pts + num_being_sampled)].reshape(num_being_sampled, gp.dim) \
        if num_being_sampled > 0 else None
    guesses = _flat(initial_guesses, gp.dim * num_to_sample * num_multistarts).reshape(num_multistarts, num_to_sample, gp.dim)
    m = (num_to_sample + max(num_being_sampled, 0)) * (1 + gp._g)
    M = int(max_int_steps)
    normals = randomness_source.normal_rng_vec[0].table(((M + 1) // 2) * m)
    r = gp._dev.kg_batch(_gd_params(optimizer_parameters), _flat(domain_bounds)[:2 * size], discrete, guesses, Xp, M,
                         float(best_so_far), normals, want_grad=False, num_fidelity=int(num_fidelity))
    values = list(r["kg_sum"] / M)
    status["evaluate_KG_at_point_list"] = bool(len(values) > 0)
    return values


def multistart_knowledge_gradient_optimization(optimizer_parameters, optimizer_parameters_inner, gaussian_process,
                                               num_fidelity, domain_bounds, discrete_pts, points_being_sampled, num_pts,
                                               num_to_sample, num_being_sampled, best_so_far, max_int_steps,
                                               max_num_threads, randomness_source, status):
    """MultistartKnowledgeGradientOptimizationWrapper (gpp_python_knowledge_gradient.cpp:243-313).  Tensor-product domain;
    the outer optimisation runs in ``cornell_moe_amd.multistart`` with every restart's KG gradient evaluated in one batched
    device pass per GD step."""
    from . import multistart
    if max_num_threads > randomness_source.num_normal_rng:
        raise BoundsException("Fewer randomness_sources than max_num_threads.", randomness_source.num_normal_rng,
                              max_num_threads, 1e9)
    _check_domain_type(optimizer_parameters, kg=True)
    gp = gaussian_process
    size = gp.dim - num_fidelity
    discrete = _flat(discrete_pts, size * num_pts).reshape(num_pts, size)
    Xp = _being_sampled(gp, points_being_sampled, num_being_sampled)
    bounds = _flat(domain_bounds, 2 * gp.dim)
    best, found = multistart.kg_optimal_points(
        gp._dev, int(num_fidelity), optimizer_parameters, optimizer_parameters_inner, bounds, discrete, Xp,
        int(num_to_sample), float(best_so_far), int(max_int_steps), randomness_source)
    kind = "gradient_descent" if int(optimizer_parameters.optimizer_type) == int(OptimizerTypes.gradient_descent) else "lhc"
    status["%s_%s_domain_found_update" % (kind, _domain_name(optimizer_parameters))] = bool(found)
    return list(np.asarray(best).ravel())


def posterior_mean_optimization(gaussian_process, num_fidelity, optimizer_parameters, domain_bounds, initial_guess, status):
    """ComputeOptimalPosteriorMeanWrapper (gpp_python_knowledge_gradient.cpp:315-342): line-search descent on the posterior
    mean from one initial guess; returns the best point (dim - num_fidelity coordinates)."""
    from . import multistart
    _check_domain_type(optimizer_parameters, kg=True)
    best, _ = multistart.posterior_mean_optimization(gaussian_process._dev, int(num_fidelity), optimizer_parameters,
                                                     _flat(domain_bounds), _flat(initial_guess))
    return list(best)


# ---- MCMC-averaged evaluators (gpp_python_knowledge_gradient_mcmc.cpp, gpp_python_expected_improvement_mcmc.cpp) ----
class GaussianProcessMCMC(object):
    """C_GP.GaussianProcessMCMC (make_gaussian_process_mcmc, gpp_python_knowledge_gradient_mcmc.cpp:45-75): num_mcmc
    Matern-5/2 GPs over the same data; hyperparameters_list flat [num_mcmc][1 + dim] = (alpha, lengths...),
    noise_variance_list flat [num_mcmc][1 + num_derivatives]."""

    def __init__(self, hyperparameters_list, noise_variance_list, points_sampled, points_sampled_value, derivatives, num_mcmc,
                 num_derivatives, dim, num_sampled, device=None):
        self.dim = int(dim)
        self.num_mcmc = int(num_mcmc)
        self._g = int(num_derivatives)
        X = _flat(points_sampled, dim * num_sampled).reshape(num_sampled, dim)
        y = _flat(points_sampled_value, num_sampled * (1 + self._g)).reshape(num_sampled, 1 + self._g)
        hyp = _flat(hyperparameters_list, self.num_mcmc * (dim + 1)).reshape(self.num_mcmc, dim + 1)
        noise = _flat(noise_variance_list, self.num_mcmc * (1 + self._g)).reshape(self.num_mcmc, 1 + self._g)
        derivs = [int(v) for v in list(derivatives)[:self._g]]
        if device is None:  # one process per GPU: the ensemble lives on this rank's device, like GaussianProcess
            device = int(os.environ.get("LOCAL_RANK", "0")) % max(_lib.device_count(), 1)
        self._dev = _api.DeviceGPMCMC(hyp, noise, X, y, derivs, device=device)

    num_sampled = property(lambda self: self._dev.n)


def _kg_mcmc(gp_mcmc, num_fidelity, optimizer_parameters, domain_bounds, discrete_pts, points_to_sample, points_being_sampled,
             num_pts, num_to_sample, num_being_sampled, max_int_steps, best_so_far, randomness_source, want_grad):
    gp = gp_mcmc
    size = gp.dim - num_fidelity
    Xq = _flat(points_to_sample, gp.dim * num_to_sample).reshape(1, num_to_sample, gp.dim)
    Xp = _being_sampled(gp, points_being_sampled, num_being_sampled)
    discrete = _flat(discrete_pts, gp.num_mcmc * num_pts * size).reshape(gp.num_mcmc, num_pts, size)
    bounds = _flat(domain_bounds, 2 * size)
    best = _flat(best_so_far, gp.num_mcmc)
    M = int(max_int_steps)
    normals = randomness_source.normal_rng_vec[0].table(((M + 1) // 2) * (num_to_sample + max(num_being_sampled, 0)) * (1 + gp._g))
    kg, grad = gp._dev.kg_batch(_gd_params(optimizer_parameters), bounds, discrete, Xq, Xp, M, best, normals, want_grad=want_grad,
                                num_fidelity=int(num_fidelity))
    return float(kg[0]), (grad[0] if want_grad else None)


def compute_knowledge_gradient_mcmc(gaussian_process_mcmc, num_fidelity, optimizer_parameters, domain_bounds, discrete_pts,
                                    points_to_sample, points_being_sampled, num_pts, num_to_sample, num_being_sampled,
                                    max_int_steps, best_so_far, randomness_source):
    """ComputeKnowledgeGradientMCMCWrapper (gpp_python_knowledge_gradient_mcmc.cpp:77-118)."""
    return _kg_mcmc(gaussian_process_mcmc, num_fidelity, optimizer_parameters, domain_bounds, discrete_pts, points_to_sample,
                    points_being_sampled, num_pts, num_to_sample, num_being_sampled, max_int_steps, best_so_far,
                    randomness_source, False)[0]


def compute_grad_knowledge_gradient_mcmc(gaussian_process_mcmc, num_fidelity, optimizer_parameters, domain_bounds, discrete_pts,
                                         points_to_sample, points_being_sampled, num_pts, num_to_sample, num_being_sampled,
                                         max_int_steps, best_so_far, randomness_source):
    """ComputeGradKnowledgeGradientMCMCWrapper (gpp_python_knowledge_gradient_mcmc.cpp:120-165)."""
    return list(_kg_mcmc(gaussian_process_mcmc, num_fidelity, optimizer_parameters, domain_bounds, discrete_pts, points_to_sample,
                         points_being_sampled, num_pts, num_to_sample, num_being_sampled, max_int_steps, best_so_far,
                         randomness_source, True)[1].ravel())


def multistart_knowledge_gradient_mcmc_optimization(optimizer_parameters, optimizer_parameters_inner, gaussian_process_mcmc,
                                                    num_fidelity, domain_bounds, discrete_pts, points_being_sampled, num_pts,
                                                    num_to_sample, num_being_sampled, best_so_far, max_int_steps,
                                                    max_num_threads, randomness_source, status):
    """MultistartKnowledgeGradientMCMCOptimizationWrapper (gpp_python_knowledge_gradient_mcmc.cpp:256-324)."""
    from . import multistart
    if max_num_threads > randomness_source.num_normal_rng:
        raise BoundsException("Fewer randomness_sources than max_num_threads.", randomness_source.num_normal_rng,
                              max_num_threads, 1e9)
    _check_domain_type(optimizer_parameters, kg=True)
    gp = gaussian_process_mcmc
    size = gp.dim - num_fidelity
    discrete = _flat(discrete_pts, gp.num_mcmc * num_pts * size).reshape(gp.num_mcmc, num_pts, size)
    Xp = _being_sampled(gp, points_being_sampled, num_being_sampled)
    best, found = multistart.kg_mcmc_optimal_points(
        gp._dev, int(num_fidelity), optimizer_parameters, optimizer_parameters_inner, _flat(domain_bounds, 2 * gp.dim), discrete,
        Xp, int(num_to_sample), _flat(best_so_far, gp.num_mcmc), int(max_int_steps), randomness_source)
    kind = "gradient_descent" if int(optimizer_parameters.optimizer_type) == int(OptimizerTypes.gradient_descent) else "lhc"
    status["%s_%s_domain_found_update" % (kind, _domain_name(optimizer_parameters))] = bool(found)
    return list(np.asarray(best).ravel())


def evaluate_KG_mcmc_at_point_list(gaussian_process_mcmc, num_fidelity, optimizer_parameters, domain_bounds, initial_guesses,
                                   discrete_being_sampled, num_multistarts, num_pts, num_to_sample, num_being_sampled,
                                   best_so_far, max_int_steps, max_num_threads, randomness_source, status):
    """EvaluateKGMCMCAtPointListWrapper (gpp_python_knowledge_gradient_mcmc.cpp:326-383).  Argument order is the C++
    wrapper's (initial_guesses BEFORE discrete_being_sampled; the reference's own Python caller,
    cpp_wrappers/knowledge_gradient_mcmc.py:292-308, passes the two the other way round).  discrete_being_sampled =
    all GPs' discrete points [num_mcmc][num_pts][dim - num_fidelity] followed by points_being_sampled [p][dim]."""
    if max_num_threads > randomness_source.num_normal_rng:
        raise BoundsException("Fewer randomness_sources than max_num_threads.", randomness_source.num_normal_rng,
                              max_num_threads, 1e9)
    gp = gaussian_process_mcmc
    size = gp.dim - num_fidelity
    n_disc = gp.num_mcmc * num_pts * size
    packed = _flat(discrete_being_sampled, n_disc + max(num_being_sampled, 0) * gp.dim)
    discrete = packed[:n_disc].reshape(gp.num_mcmc, num_pts, size)
    Xp = packed[n_disc:].reshape(num_being_sampled, gp.dim) if num_being_sampled > 0 else None
    guesses = _flat(initial_guesses, gp.dim * num_to_sample * num_multistarts).reshape(num_multistarts, num_to_sample, gp.dim)
    M = int(max_int_steps)
    normals = randomness_source.normal_rng_vec[0].table(((M + 1) // 2) * (num_to_sample + max(num_being_sampled, 0)) * (1 + gp._g))
    kg, _ = gp._dev.kg_batch(_gd_params(optimizer_parameters), _flat(domain_bounds, 2 * size), discrete, guesses, Xp, M,
                             _flat(best_so_far, gp.num_mcmc), normals, want_grad=False, num_fidelity=int(num_fidelity))
    status["evaluate_KG_at_point_list"] = bool(len(kg) > 0 and np.max(kg) > -np.inf)
    return list(kg)


def _ei_mcmc(gp_mcmc, points_to_sample, points_being_sampled, num_to_sample, num_being_sampled, max_int_steps, best_so_far,
             randomness_source, want_grad):
    gp = gp_mcmc
    Xq = _flat(points_to_sample, gp.dim * num_to_sample).reshape(1, num_to_sample, gp.dim)
    Xp = _being_sampled(gp, points_being_sampled, num_being_sampled)
    u = num_to_sample + max(num_being_sampled, 0)
    normals = randomness_source.normal_rng_vec[0].table(int(max_int_steps) * u)
    ei, grad = gp._dev.ei_batch(Xq, Xp, int(max_int_steps), _flat(best_so_far, gp.num_mcmc), normals, want_grad=want_grad)
    return float(ei[0]), (grad[0] if want_grad else None)


def compute_expected_improvement_mcmc(gaussian_process_mcmc, points_to_sample, points_being_sampled, num_to_sample,
                                      num_being_sampled, max_int_steps, best_so_far, randomness_source):
    """ComputeExpectedImprovementMCMCWrapper (gpp_python_expected_improvement_mcmc.cpp:42-72): always Monte Carlo."""
    return _ei_mcmc(gaussian_process_mcmc, points_to_sample, points_being_sampled, num_to_sample, num_being_sampled,
                    max_int_steps, best_so_far, randomness_source, False)[0]


def compute_grad_expected_improvement_mcmc(gaussian_process_mcmc, points_to_sample, points_being_sampled, num_to_sample,
                                           num_being_sampled, max_int_steps, best_so_far, randomness_source):
    """ComputeGradExpectedImprovementMCMCWrapper (gpp_python_expected_improvement_mcmc.cpp:74-108)."""
    return list(_ei_mcmc(gaussian_process_mcmc, points_to_sample, points_being_sampled, num_to_sample, num_being_sampled,
                         max_int_steps, best_so_far, randomness_source, True)[1].ravel())


def multistart_expected_improvement_mcmc_optimization(optimizer_parameters, gaussian_process_mcmc, domain_bounds,
                                                      points_being_sampled, num_to_sample, num_being_sampled, best_so_far,
                                                      max_int_steps, max_num_threads, randomness_source, status):
    """MultistartExpectedImprovementMCMCOptimizationWrapper (gpp_python_expected_improvement_mcmc.cpp)."""
    from . import multistart
    if max_num_threads > randomness_source.num_normal_rng:
        raise BoundsException("Fewer randomness_sources than max_num_threads.", randomness_source.num_normal_rng,
                              max_num_threads, 1e9)
    _check_domain_type(optimizer_parameters)
    gp = gaussian_process_mcmc
    Xp = _being_sampled(gp, points_being_sampled, num_being_sampled)
    best, found = multistart.ei_mcmc_optimal_points(gp._dev, optimizer_parameters, _flat(domain_bounds, 2 * gp.dim), Xp,
                                                    int(num_to_sample), _flat(best_so_far, gp.num_mcmc), int(max_int_steps),
                                                    randomness_source)
    kind = "gradient_descent" if int(optimizer_parameters.optimizer_type) == int(OptimizerTypes.gradient_descent) else "lhc"
    status["%s_%s_domain_found_update" % (kind, _domain_name(optimizer_parameters))] = bool(found)
    return list(np.asarray(best).ravel())


def evaluate_EI_mcmc_at_point_list(gaussian_process_mcmc, initial_guesses, points_being_sampled, num_multistarts, num_to_sample,
                                   num_being_sampled, best_so_far, max_int_steps, max_num_threads, randomness_source, status):
    """EvaluateEIMCMCAtPointListWrapper -> EvaluateEIMCMCAtPointList (gpp_expected_improvement_mcmc_optimization.cpp:239-298):
    the analytic evaluator at num_to_sample == 1, num_being_sampled == 0."""
    if max_num_threads > randomness_source.num_normal_rng:
        raise BoundsException("Fewer randomness_sources than max_num_threads.", randomness_source.num_normal_rng,
                              max_num_threads, 1e9)
    gp = gaussian_process_mcmc
    guesses = _flat(initial_guesses, gp.dim * num_to_sample * num_multistarts).reshape(num_multistarts, num_to_sample, gp.dim)
    Xp = _being_sampled(gp, points_being_sampled, num_being_sampled)
    analytic = num_to_sample == 1 and num_being_sampled <= 0
    u = num_to_sample + max(num_being_sampled, 0)
    normals = None if analytic else randomness_source.normal_rng_vec[0].table(int(max_int_steps) * u)
    ei, _ = gp._dev.ei_batch(guesses, Xp, int(max_int_steps), _flat(best_so_far, gp.num_mcmc), normals, want_grad=False,
                             analytic=analytic)
    status["evaluate_EI_at_point_list"] = bool(len(ei) > 0 and np.max(ei) > 0.0)
    return list(ei)


# ---- log marginal likelihood (gpp_python_model_selection.cpp:43-69, 281-340) ----
_LL_CACHE = {}  # data fingerprint -> api.LogLikelihood: a sampler evaluates thousands of hyper-parameter sets on the same data


def _ll_handle(points_sampled, points_sampled_value, dim, num_sampled, derivatives, num_derivatives):
    X = _flat(points_sampled, dim * num_sampled).reshape(num_sampled, dim)
    y = _flat(points_sampled_value, num_sampled * (1 + num_derivatives)).reshape(num_sampled, 1 + num_derivatives)
    derivs = tuple(int(v) for v in list(derivatives)[:num_derivatives])
    key = (X.tobytes(), y.tobytes(), derivs)
    h = _LL_CACHE.get(key)
    if h is None:
        if len(_LL_CACHE) >= 4:
            _LL_CACHE.clear()
        h = _LL_CACHE[key] = _api.LogLikelihood(X, y, derivs)
    return h


def _check_objective(objective_type):
    if int(objective_type) != int(LogLikelihoodTypes.log_marginal_likelihood):
        raise OptimalLearningException("ERROR: invalid objective mode choice. Setting log likelihood to -DBL_MAX.")


def compute_log_likelihood(points_sampled, points_sampled_value, dim, num_sampled, objective_type, hyperparameters, derivatives,
                           num_derivatives, noise_variance):
    """ComputeLogLikelihoodWrapper (gpp_python_model_selection.cpp:43-69): log marginal likelihood with a Matern-5/2 kernel;
    hyperparameters = [alpha, [lengths]] (cpp_utils.cppify_hyperparameters)."""
    _check_objective(objective_type)
    h = _ll_handle(points_sampled, points_sampled_value, dim, num_sampled, derivatives, num_derivatives)
    hyper = np.r_[float(hyperparameters[0]), _flat(hyperparameters[1], dim), _flat(noise_variance, 1 + num_derivatives)]
    return float(h.evaluate(hyper[None, :])[0])


def compute_hyperparameter_grad_log_likelihood(points_sampled, points_sampled_value, dim, num_sampled, objective_type,
                                               hyperparameters, derivatives, num_derivatives, noise_variance):
    """ComputeHyperparameterGradLogLikelihoodWrapper (gpp_python_model_selection.cpp:88-135): list of
    1 + dim + 1 + num_derivatives partials wrt (alpha, lengths, noise variances), Matern-5/2 kernel."""
    _check_objective(objective_type)
    h = _ll_handle(points_sampled, points_sampled_value, dim, num_sampled, derivatives, num_derivatives)
    hyper = np.r_[float(hyperparameters[0]), _flat(hyperparameters[1], dim), _flat(noise_variance, 1 + num_derivatives)]
    return list(h.grad(hyper))


def evaluate_log_likelihood_at_hyperparameter_list(hyperparameter_list, points_sampled, points_sampled_value, dim, num_sampled,
                                                   objective_mode, hyperparameters, noise_variance, derivatives,
                                                   num_derivatives, num_multistarts, max_num_threads, status):
    """EvaluateLogLikelihoodAtHyperparameterListWrapper (gpp_python_model_selection.cpp:281-340): hyperparameter_list is
    flat [num_multistarts][1 + dim + 1 + num_derivatives]."""
    _check_objective(objective_mode)
    h = _ll_handle(points_sampled, points_sampled_value, dim, num_sampled, derivatives, num_derivatives)
    width = 1 + dim + 1 + num_derivatives
    vals = h.evaluate(_flat(hyperparameter_list, width * num_multistarts).reshape(num_multistarts, width))
    status["evaluate_log_marginal_likelihood_at_hyperparameter_list"] = bool(len(vals) > 0 and np.max(vals) > -np.inf)
    return list(vals)


def _hyper_guesses(randomness_source, domain_log10, count):
    """ConvertFromLogToLinearDomainAndBuildInitialGuesses (gpp_model_selection.hpp:841-858): a Latin hypercube in LOG-10 space,
    exponentiated.  (The reference draws it from randomness_source.uniform_generator, whose stream depends on its Boost version;
    here the generator of moe_latin_hypercube is seeded from the container's uniform seed -- reproducible for a given seed.)"""
    pts = _api.latin_hypercube(randomness_source._next_uniform_seed(), np.asarray(domain_log10, dtype=np.float64).reshape(-1), count)
    return 10.0 ** pts


def multistart_hyperparameter_optimization(optimizer_parameters, hyperparameter_domain, points_sampled, points_sampled_value, dim,
                                           num_sampled, hyperparameters, noise_variance, derivatives, num_derivatives,
                                           max_num_threads, randomness_source, status):
    """MultistartHyperparameterOptimizationWrapper (gpp_python_model_selection.cpp:218-279 -> DispatchHyperparameterOptimization,
    :150-216): maximum-likelihood hyper-parameters [alpha, lengths..., noise variances...] over hyperparameter_domain (LOG-10 space,
    [n_hyper][2]).  optimizer_type gradient_descent: MultistartGradientDescentHyperparameterOptimization
    (gpp_model_selection.hpp:1063-1103) -- num_multistarts Latin-hypercube guesses, restarted gradient ascent from each, all of them
    stepped together on the device (moe_ll_multistart); null: the Latin-hypercube value search (:1342-1363) over num_random_samples
    points.  status gets the reference's key."""
    opt = optimizer_parameters
    _check_objective(getattr(opt, "objective_type", LogLikelihoodTypes.log_marginal_likelihood))
    h = _ll_handle(points_sampled, points_sampled_value, dim, num_sampled, derivatives, num_derivatives)
    nh = 1 + dim + 1 + num_derivatives
    dom = _flat(hyperparameter_domain, 2 * nh).reshape(nh, 2)
    if int(opt.optimizer_type) == int(OptimizerTypes.gradient_descent):
        gd = _gd_params(opt)
        guesses = _hyper_guesses(randomness_source, dom, int(gd[0]))
        best, _, found = h.multistart(gd, dom, guesses)
        status["log_marginal_likelihood_gradient_descent_found_update"] = bool(found)
        return list(best)
    if int(opt.optimizer_type) == int(OptimizerTypes.null):
        n_lhc = int(opt.num_random_samples)
        if n_lhc <= 0:
            raise BoundsException("num_multistarts must be > 1", n_lhc, 1, 0)
        guesses = _hyper_guesses(randomness_source, dom, n_lhc)
        vals = h.evaluate(guesses)
        k = int(np.argmax(vals))   # (first of equal values, like MultistartOptimizer's strict compare)
        found = bool(vals[k] > -np.inf)
        status["log_marginal_likelihood_lhc_found_update"] = found
        return list(guesses[k] if found else guesses[0])
    raise OptimalLearningException("ERROR: invalid optimizer choice. Setting all hyperparameters to 1.0.")


def restarted_hyperparameter_optimization(optimizer_parameters, hyperparameter_domain, points_sampled, points_sampled_value, dim,
                                          num_sampled, hyperparameters, noise_variance, derivatives, num_derivatives, status):
    """RestartedGradientDescentHyperparameterOptimizationWrapper (gpp_python_model_selection.cpp:342-375 ->
    RestartedGradientDescentHyperparameterOptimizationTensor, gpp_model_selection.hpp:989-1012): restarted gradient ascent from the
    caller's CURRENT hyper-parameters (hyperparameters = [alpha, [lengths]], noise_variance) inside the LOG-10 domain; returns the
    point the ascent ends at (the reference does not compare it with the start)."""
    h = _ll_handle(points_sampled, points_sampled_value, dim, num_sampled, derivatives, num_derivatives)
    nh = 1 + dim + 1 + num_derivatives
    dom = _flat(hyperparameter_domain, 2 * nh).reshape(nh, 2)
    x0 = np.r_[float(hyperparameters[0]), _flat(hyperparameters[1], dim), _flat(noise_variance, 1 + num_derivatives)]
    gd = _gd_params(optimizer_parameters)
    if int(gd[2]) <= 0:   # max_num_restarts <= 0: the reference returns without touching its output (:995-997)
        return list(np.zeros(nh))
    end = h.ascend(gd, dom, x0)
    return list(end)


def run_cpp_tests():
    """The reference runs its C++ unit-test suites here and returns the number of failures (gpp_python_test.cpp:60-314).  This backend
    runs its device self-test (cornell_moe_amd/selftest.py): the same kinds of check -- finite-difference pings of every gradient entry
    point, analytic vs Monte-Carlo EI, linear algebra, random sources, optimiser end-to-end -- on the GPU, each against an identity, not
    against the reference (parity with the reference is the pytest suite under tests/).  Returns the number of failed checks."""
    from . import selftest
    return selftest.run()
