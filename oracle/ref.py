"""ctypes binding of oracle/_ref/libmoe_ref.so -- TEST INFRASTRUCTURE ONLY.

The library is the unmodified reference C++ core (compiled in place by oracle/Makefile) behind oracle/ref_harness.cpp.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libmoe_ref.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def available():
    return os.path.exists(_PATH)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError("oracle/_ref/libmoe_ref.so not built (run `make -C oracle ref` where /root/reference exists)")
        _lib = C.CDLL(_PATH)
        _lib.ref_last_error.restype = C.c_char_p
        _lib.ref_gp_create.restype = C.c_void_p
        _lib.ref_gp_create.argtypes = [C.c_int, C.c_double, _dp, _dp, _dp, _dp, _ip, C.c_int, C.c_int, C.c_int]
        _lib.ref_gp_destroy.argtypes = [C.c_void_p]
        _lib.ref_gp_num_sampled.argtypes = [C.c_void_p]
        _lib.ref_gp_add_points.argtypes = [C.c_void_p, _dp, _dp, C.c_int]
        _lib.ref_gp_dump.argtypes = [C.c_void_p, _dp, _dp, _dp]
        _lib.ref_gp_mix_cov.argtypes = [C.c_void_p, _dp, C.c_int, _ip, C.c_int, _dp]
        _lib.ref_gp_mean.argtypes = [C.c_void_p, _dp, C.c_int, _dp]
        _lib.ref_gp_additional_mean.argtypes = [C.c_void_p, _dp, C.c_int, _ip, C.c_int, _dp]
        _lib.ref_gp_grad_mean.argtypes = [C.c_void_p, _dp, C.c_int, _dp]
        _lib.ref_gp_var.argtypes = [C.c_void_p, _dp, C.c_int, _dp]
        _lib.ref_gp_chol_var.argtypes = [C.c_void_p, _dp, C.c_int, _dp]
        _lib.ref_gp_grad_var.argtypes = [C.c_void_p, _dp, C.c_int, C.c_int, _dp]
        _lib.ref_gp_grad_chol_var.argtypes = [C.c_void_p, _dp, C.c_int, C.c_int, _dp]
        _lib.ref_posterior_mean.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp]
        _lib.ref_covariance.argtypes = [C.c_int, C.c_int, C.c_double, _dp, _dp, _ip, C.c_int, _dp, _ip, C.c_int, _dp, _dp]
        _lib.ref_cholesky.argtypes = [C.c_int, _dp]
        _lib.ref_chol_solve.argtypes = [_dp, C.c_int, _dp]
        _lib.ref_tri_solve.argtypes = [_dp, C.c_char, C.c_int, _dp]
        _lib.ref_ei.argtypes = [C.c_void_p, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_double, _dp, _dp, _dp, _dp]
        _lib.ref_ei_analytic.argtypes = [C.c_void_p, _dp, C.c_double, _dp, _dp]
        _lib.ref_ei_multistart_analytic.argtypes = [C.c_void_p, _dp, _dp, _dp, C.c_int, C.c_double, C.POINTER(C.c_int), _dp]
        _lib.ref_ei_multistart_analytic_dom.argtypes = [C.c_void_p, _dp, _dp, _dp, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_int), _dp]
        _lib.ref_gpmcmc_create.restype = C.c_void_p
        _lib.ref_gpmcmc_create.argtypes = [_dp, _dp, C.c_int, _dp, _dp, _ip, C.c_int, C.c_int, C.c_int]
        _lib.ref_gpmcmc_destroy.argtypes = [C.c_void_p]
        _lib.ref_kg_mcmc.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, C.c_int, _dp, _dp, C.c_int, C.c_int, C.c_int, _dp, _dp,
                                     C.c_long, C.c_int, _dp, _dp]
        _lib.ref_ei_mcmc.argtypes = [C.c_void_p, _dp, _dp, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp]
        if hasattr(_lib, "ref_kg_mcmc_multistart_mt"):   # (a prebuilt oracle/_ref of an earlier round does not have it)
            _lib.ref_kg_mcmc_multistart_mt.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _dp, C.c_int, _dp, C.c_int, _dp, C.c_int,
                                                       C.c_int, C.c_int, _dp, C.c_uint, C.c_int, _ip, _dp, _dp]
        _lib.ref_ei_mcmc_multistart_analytic.argtypes = [C.c_void_p, _dp, _dp, _dp, C.c_int, _dp, C.POINTER(C.c_int), _dp]
        _lib.ref_log_likelihood.argtypes = [C.c_int, C.c_double, _dp, _dp, _dp, _dp, _ip, C.c_int, C.c_int, C.c_int, _dp]
        _lib.ref_kg.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, C.c_int, _dp, _dp, C.c_int, C.c_int, C.c_int,
                                C.c_double, _dp, C.c_long, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
        _lib.ref_kg_dom.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, C.c_int, _dp, _dp, C.c_int, C.c_int, C.c_int,
                                    C.c_double, _dp, C.c_long, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
        _lib.ref_kg_multistart_dom.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _dp, C.c_int, _dp, C.c_int, _dp, C.c_int,
                                               C.c_int, C.c_int, C.c_double, C.c_uint, C.c_int, C.POINTER(C.c_int), _dp]
        _lib.ref_normal_draws.argtypes = [C.c_uint, C.c_long, _dp]
        _lib.ref_kg_multistart.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _dp, C.c_int, _dp, C.c_int, _dp, C.c_int,
                                           C.c_int, C.c_int, C.c_double, C.c_uint, C.POINTER(C.c_int), _dp]
        _lib.ref_kg_mcmc_multistart.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _dp, C.c_int, _dp, C.c_int, _dp, C.c_int,
                                                C.c_int, C.c_int, _dp, C.c_uint, C.POINTER(C.c_int), _dp]
        _lib.ref_kg_seeded.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, C.c_int, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_double,
                                       C.c_uint, _dp]
        _lib.ref_kg_grad_batch.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, C.c_int, _dp, C.c_int, C.c_int, C.c_int,
                                           C.c_double, _dp, C.c_long, C.c_int, _dp, _dp, _dp]
    return _lib


def _d(a):
    """contiguous float64 array + its pointer (None -> NULL)."""
    if a is None:
        return None, None
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _i(a):
    if a is None or len(a) == 0:
        return None, None
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


class RefError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("reference error %d: %s" % (code, msg))
        self.code = code


def _check(rc):
    if rc != 0:
        raise RefError(rc, lib().ref_last_error().decode("utf-8", "replace"))


def covariance(cov_type, alpha, lengths, p1, d1, p2, d2):
    """(cov[(1+g1),(1+g2)] col-major flat, grad_cov[d,(1+g1),(1+g2)] flat) from the reference covariance classes."""
    L = lib()
    dim = len(p1)
    g1, g2 = len(d1), len(d2)
    cov = np.zeros((1 + g1) * (1 + g2))
    gcov = np.zeros(dim * (1 + g1) * (1 + g2))
    l_, lp = _d(lengths)
    a1, p1p = _d(p1)
    a2, p2p = _d(p2)
    i1, d1p = _i(d1)
    i2, d2p = _i(d2)
    _check(L.ref_covariance(cov_type, dim, alpha, lp, p1p, d1p, g1, p2p, d2p, g2, cov.ctypes.data_as(_dp),
                            gcov.ctypes.data_as(_dp)))
    return cov, gcov


def cholesky(a):
    a = np.array(a, dtype=np.float64, order="F")
    n = a.shape[0]
    flat = np.ascontiguousarray(a.T).ravel().copy()  # col-major flat
    rc = lib().ref_cholesky(n, flat.ctypes.data_as(_dp))
    return rc, flat.reshape(n, n).T.copy()


class RefGP(object):
    """The reference GaussianProcess (gpp_math.hpp:275-868) behind the harness."""

    def __init__(self, cov_type, alpha, lengths, X, y, noise, derivs):
        L = lib()
        X = np.ascontiguousarray(X, dtype=np.float64)
        self.n, self.d = X.shape
        self.derivs = [int(v) for v in derivs]
        self.g = len(self.derivs)
        self.N = self.n * (1 + self.g)
        self.keep = [X, _d(y)[0], _d(noise)[0], _d(lengths)[0], _i(self.derivs)[0]]
        dp = self.keep[4].ctypes.data_as(_ip) if self.keep[4] is not None else None
        self.h = L.ref_gp_create(cov_type, alpha, self.keep[3].ctypes.data_as(_dp), X.ctypes.data_as(_dp),
                                 self.keep[1].ctypes.data_as(_dp), self.keep[2].ctypes.data_as(_dp), dp, self.g, self.d,
                                 self.n)
        if not self.h:
            raise RefError(4, L.ref_last_error().decode("utf-8", "replace"))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().ref_gp_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def dump(self):
        K = np.zeros(self.N * self.N)
        kiy = np.zeros(self.N)
        mean = C.c_double(0.0)
        lib().ref_gp_dump(self.h, K.ctypes.data_as(_dp), kiy.ctypes.data_as(_dp), C.byref(mean))
        return K.reshape(self.N, self.N).T.copy(), kiy, mean.value  # K_chol as [row, col]

    def add_points(self, pts, vals):
        pts, pp = _d(pts)
        vals, vp = _d(vals)
        k = pts.reshape(-1, self.d).shape[0]
        _check(lib().ref_gp_add_points(self.h, pp, vp, k))
        self.n += k
        self.N = self.n * (1 + self.g)

    def mix_cov(self, pts, derivs2=()):
        pts, pp = _d(pts)
        k = pts.reshape(-1, self.d).shape[0]
        i2, d2p = _i(list(derivs2))
        g2 = len(derivs2)
        out = np.zeros(self.N * k * (1 + g2))
        lib().ref_gp_mix_cov(self.h, pp, k, d2p, g2, out.ctypes.data_as(_dp))
        return out.reshape(k * (1 + g2), self.N).T.copy()  # [row=N, col]

    def _q(self, fn, pts, size, *extra):
        pts, pp = _d(pts)
        k = pts.reshape(-1, self.d).shape[0]
        out = np.zeros(size(k))
        _check(fn(self.h, pp, k, *extra, out.ctypes.data_as(_dp)))
        return out

    def mean(self, pts):
        return self._q(lib().ref_gp_mean, pts, lambda k: k)

    def additional_mean(self, pts, derivs2=()):
        i2, d2p = _i(list(derivs2))
        g2 = len(derivs2)
        return self._q(lib().ref_gp_additional_mean, pts, lambda k: k * (1 + g2), d2p, g2)

    def grad_mean(self, pts):
        return self._q(lib().ref_gp_grad_mean, pts, lambda k: self.d * k * (1 + self.g))

    def var(self, pts):
        return self._q(lib().ref_gp_var, pts, lambda k: (k * (1 + self.g)) ** 2)

    def chol_var(self, pts):
        return self._q(lib().ref_gp_chol_var, pts, lambda k: (k * (1 + self.g)) ** 2)

    def grad_var(self, pts, nd):
        return self._q(lib().ref_gp_grad_var, pts, lambda k: self.d * (k * (1 + self.g)) ** 2 * nd, nd)

    def grad_chol_var(self, pts, nd):
        return self._q(lib().ref_gp_grad_chol_var, pts, lambda k: self.d * (k * (1 + self.g)) ** 2 * nd, nd)

    def posterior_mean(self, pt, num_fidelity=0):
        pt, pp = _d(pt)
        val = C.c_double(0.0)
        grad = np.zeros(self.d - num_fidelity)
        _check(lib().ref_posterior_mean(self.h, num_fidelity, pp, C.byref(val), grad.ctypes.data_as(_dp)))
        return val.value, grad

    def ei(self, Xq, Xp, M, best_so_far, normals, want_grad=True):
        Xq, qp = _d(Xq)
        q = Xq.reshape(-1, self.d).shape[0]
        if Xp is None or len(Xp) == 0:
            p, Xp_, pp = 0, None, None
        else:
            Xp_, pp = _d(Xp)
            p = Xp_.reshape(-1, self.d).shape[0]
        normals, npp = _d(normals)
        assert normals.size >= M * (q + p)
        ei = C.c_double(0.0)
        sec = C.c_double(0.0)
        grad = np.zeros(q * self.d) if want_grad else None
        _check(lib().ref_ei(self.h, qp, pp, q, p, M, best_so_far, npp, C.byref(ei),
                            grad.ctypes.data_as(_dp) if want_grad else None, C.byref(sec)))
        return ei.value, (grad.reshape(q, self.d) if want_grad else None), sec.value

    def ei_analytic(self, pt, best_so_far):
        """OnePotentialSampleExpectedImprovementEvaluator: (EI, grad [d])."""
        pt, pp = _d(pt)
        ei = C.c_double(0.0)
        grad = np.zeros(self.d)
        _check(lib().ref_ei_analytic(self.h, pp, best_so_far, C.byref(ei), grad.ctypes.data_as(_dp)))
        return ei.value, grad

    def ei_multistart_analytic(self, gd, bounds, starts, best_so_far, domain_type=0):
        """ComputeOptimalPointsToSampleViaMultistartGradientDescent at q = 1, p = 0: (best_point [d], found).
        domain_type 1: over SimplexIntersectTensorProductDomain."""
        gd, gdp = _d(gd)
        bounds, bp = _d(bounds)
        starts, sp = _d(starts)
        S = starts.reshape(-1, self.d).shape[0]
        assert S >= 20, "the reference pops its top-20 queue unconditionally"
        found = C.c_int(0)
        best = np.zeros(self.d)
        _check(lib().ref_ei_multistart_analytic_dom(self.h, gdp, bp, sp, S, best_so_far, int(domain_type), C.byref(found),
                                                    best.ctypes.data_as(_dp)))
        return best, bool(found.value)

    def kg(self, gd, bounds, discrete, Xq, Xp, M, best_so_far, normals, want_grad=True, num_fidelity=0, details=False,
           domain_type=0):
        """Returns dict(kg, grad[q,d], best_point[M,d], seconds=(state, eval), ...).  domain_type 1: the inner optimisations over
        SimplexIntersectTensorProductDomain (what the reference builds for DomainTypes::kSimplex)."""
        gd, gdp = _d(gd)
        bounds, bp = _d(bounds)
        discrete, dp = _d(discrete)
        P = discrete.reshape(-1, self.d - num_fidelity).shape[0]
        Xq, qp = _d(Xq)
        q = Xq.reshape(-1, self.d).shape[0]
        if Xp is None or len(Xp) == 0:
            p, Xp_, pp = 0, None, None
        else:
            Xp_, pp = _d(Xp)
            p = Xp_.reshape(-1, self.d).shape[0]
        m = (q + p) * (1 + self.g)
        normals, npp = _d(normals)
        assert normals.size >= ((M + 1) // 2) * m, "normal table too small"
        kg = C.c_double(0.0)
        grad = np.zeros(q * self.d)
        best_point = np.zeros(M * self.d)
        tsm = np.zeros(m)
        chol = np.zeros(m * m)
        gchol = np.zeros(self.d * m * m * q)
        cic = np.zeros(m * M)
        sec = np.zeros(2)
        P_ = lambda a: a.ctypes.data_as(_dp)  # noqa: E731
        _check(lib().ref_kg_dom(self.h, num_fidelity, gdp, bp, dp, P, qp, pp, q, p, M, best_so_far, npp, normals.size,
                                1 if want_grad else 0, int(domain_type), C.byref(kg), P_(grad), P_(best_point), P_(tsm), P_(chol),
                                P_(gchol), P_(cic), P_(sec)))
        out = dict(kg=kg.value, grad=grad.reshape(q, self.d) if want_grad else None,
                   best_point=best_point.reshape(M, self.d), seconds=(sec[0], sec[1]))
        if details:
            out.update(to_sample_mean=tsm, chol_var=chol.reshape(m, m).T.copy(), grad_chol=gchol,
                       chol_inverse_cov=cic.reshape(M, m))
        return out

    def kg_multistart(self, gd_outer, gd_inner, bounds, discrete, starts, Xp, M, best_so_far, seed, num_fidelity=0, domain_type=0):
        """ComputeKGOptimalPointsToSampleViaMultistartGradientDescent (one thread) with NormalRNG(seed):
        (best_points [q,d], found).  starts[S][q][d], S >= 20.  The normal table the run consumed is normal_draws(seed, ...).
        domain_type 1: outer AND inner domain SimplexIntersectTensorProductDomain."""
        gdo, gop = _d(gd_outer)
        gdi, gip = _d(gd_inner)
        bounds, bp = _d(bounds)
        inner_bounds = np.ascontiguousarray(bounds.reshape(-1)[: 2 * (self.d - num_fidelity)])
        discrete, dp = _d(discrete)
        P = discrete.reshape(-1, self.d - num_fidelity).shape[0]
        starts = np.ascontiguousarray(starts, dtype=np.float64)
        S, q, _ = starts.shape
        assert S >= 20, "the reference pops its top-20 queue unconditionally"
        if Xp is None or len(Xp) == 0:
            p, pp = 0, None
        else:
            Xp_, pp = _d(Xp)
            p = Xp_.reshape(-1, self.d).shape[0]
        found = C.c_int(0)
        best = np.zeros(q * self.d)
        _check(lib().ref_kg_multistart_dom(self.h, num_fidelity, gop, gip, bp, inner_bounds.ctypes.data_as(_dp), dp, P,
                                           starts.ctypes.data_as(_dp), S, pp, q, p, M, best_so_far, seed, int(domain_type),
                                           C.byref(found), best.ctypes.data_as(_dp)))
        return best.reshape(q, self.d), bool(found.value)

    def kg_seeded(self, gd, bounds, discrete, Xq, Xp, M, best_so_far, seed, num_fidelity=0):
        """ComputeKnowledgeGradient on a fresh state drawing from NormalRNG(seed) -- the route the reference's Python entry
        points take (gpp_python_knowledge_gradient.cpp:74-154) -- instead of an injected table."""
        gd, gdp = _d(gd)
        bounds, bp = _d(bounds)
        discrete, dp = _d(discrete)
        P = discrete.reshape(-1, self.d - num_fidelity).shape[0]
        Xq, qp = _d(Xq)
        q = Xq.reshape(-1, self.d).shape[0]
        if Xp is None or len(Xp) == 0:
            p, pp = 0, None
        else:
            Xp_, pp = _d(Xp)
            p = Xp_.reshape(-1, self.d).shape[0]
        kg = C.c_double(0.0)
        _check(lib().ref_kg_seeded(self.h, num_fidelity, gdp, bp, dp, P, qp, pp, q, p, M, best_so_far, int(seed), C.byref(kg)))
        return kg.value

    def kg_grad_batch(self, gd, bounds, discrete, Xq_all, M, best_so_far, normals, num_threads, num_fidelity=0):
        gd, gdp = _d(gd)
        bounds, bp = _d(bounds)
        discrete, dp = _d(discrete)
        P = discrete.reshape(-1, self.d - num_fidelity).shape[0]
        Xq_all = np.ascontiguousarray(Xq_all, dtype=np.float64)
        R, q, _ = Xq_all.shape
        normals, npp = _d(normals)
        kg = np.zeros(R)
        grad = np.zeros(R * q * self.d)
        wall = C.c_double(0.0)
        _check(lib().ref_kg_grad_batch(self.h, num_fidelity, gdp, bp, dp, P, Xq_all.ctypes.data_as(_dp), R, q, M,
                                       best_so_far, npp, normals.size, num_threads, kg.ctypes.data_as(_dp),
                                       grad.ctypes.data_as(_dp), C.byref(wall)))
        return kg, grad.reshape(R, q, self.d), wall.value


def normal_draws(seed, count):
    """The first `count` draws of the reference's NormalRNG(seed) (mt19937 + the shimmed normal distribution)."""
    out = np.zeros(int(count))
    _check(lib().ref_normal_draws(int(seed), int(count), out.ctypes.data_as(_dp)))
    return out


def num_procs():
    return lib().ref_num_procs()


def log_likelihood(cov_type, alpha, lengths, X, y, noise, derivs):
    """LogMarginalLikelihoodEvaluator::ComputeLogLikelihood on a fresh state (gpp_python_model_selection.cpp:43-69)."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    n, d = X.shape
    derivs = [int(v) for v in derivs]
    ya, yp = _d(y)
    na, np_ = _d(noise)
    la, lp = _d(lengths)
    da, dp = _i(derivs)
    val = C.c_double(0.0)
    _check(lib().ref_log_likelihood(cov_type, float(alpha), lp, X.ctypes.data_as(_dp), yp, np_, dp, len(derivs), d, n, C.byref(val)))
    return val.value


def ll_multistart(cov_type, alpha, lengths, X, y, noise, derivs, gd, domain_log10, initial_guesses, num_threads=1):
    """MultistartGradientDescentHyperparameterOptimization from caller-supplied linear-space guesses (ref_harness.cpp:
    ref_ll_multistart): (best [1 + d + 1 + g], best log likelihood, found)."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    n, d = X.shape
    lengths, lp = _d(lengths)
    ya, yp = _d(y)
    noise, nop = _d(noise)
    dv, dvp = _i(list(derivs))
    g = len(derivs)
    gd, gdp = _d(gd)
    dom, dmp = _d(domain_log10)
    x0 = np.ascontiguousarray(initial_guesses, dtype=np.float64).reshape(-1, 1 + d + 1 + g)
    found, val = C.c_int(0), C.c_double(0.0)
    best = np.zeros(1 + d + 1 + g)
    L = lib()
    L.ref_ll_multistart.argtypes = [C.c_int, C.c_double, _dp, _dp, _dp, _dp, _ip, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, C.c_int,
                                    C.c_int, _ip, _dp, _dp]
    _check(L.ref_ll_multistart(int(cov_type), float(alpha), lp, X.ctypes.data_as(_dp), yp, nop, dvp, g, d, n, gdp, dmp,
                               x0.ctypes.data_as(_dp), x0.shape[0], int(num_threads), C.byref(found), best.ctypes.data_as(_dp),
                               C.byref(val)))
    return best, val.value, bool(found.value)


def log_likelihood_grad(cov_type, alpha, lengths, X, y, noise, derivs):
    """LogMarginalLikelihoodEvaluator::ComputeGradLogLikelihood on a fresh state (gpp_python_model_selection.cpp:88-135).
    Returns [1 + d + 1 + g] partials wrt (alpha, lengths, noise variances)."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    n, d = X.shape
    derivs = [int(v) for v in derivs]
    ya, yp = _d(y)
    na, np_ = _d(noise)
    la, lp = _d(lengths)
    da, dp = _i(derivs)
    out = np.zeros(1 + d + 1 + len(derivs))
    fn = lib().ref_log_likelihood_grad
    fn.argtypes = [C.c_int, C.c_double, _dp, _dp, _dp, _dp, _ip, C.c_int, C.c_int, C.c_int, _dp]
    fn.restype = C.c_int
    rc = fn(cov_type, float(alpha), lp, X.ctypes.data_as(_dp), yp, np_, dp, len(derivs), d, n, out.ctypes.data_as(_dp))
    _check(rc)
    return out


class RefGPMCMC(object):
    """The reference GaussianProcessMCMC (gpp_knowledge_gradient_mcmc_optimization.hpp:140-198): one Matern-5/2 GP per
    hyper-parameter sample over the same data.  hypers [num_mcmc][1 + d] = (alpha, lengths), noises [num_mcmc][1 + g]."""

    def __init__(self, hypers, noises, X, y, derivs):
        X = np.ascontiguousarray(X, dtype=np.float64)
        self.n, self.d = X.shape
        self.derivs = [int(v) for v in derivs]
        self.g = len(self.derivs)
        hypers, hp = _d(hypers)
        noises, nop = _d(noises)
        self.num_mcmc = hypers.reshape(-1, self.d + 1).shape[0]
        ya, yp = _d(y)
        da, dp = _i(self.derivs)
        self.h = lib().ref_gpmcmc_create(hp, nop, self.num_mcmc, X.ctypes.data_as(_dp), yp, dp, self.g, self.d, self.n)
        if not self.h:
            raise RefError(4, lib().ref_last_error().decode("utf-8", "replace"))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().ref_gpmcmc_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def kg(self, gd, bounds, discrete_all, Xq, Xp, M, best_so_far, normals, want_grad=True, num_fidelity=0):
        """KnowledgeGradientMCMCEvaluator: (KG, grad [q][d] or None).  discrete_all [num_mcmc][P][d-f]; best_so_far [num_mcmc]."""
        gd, gdp = _d(gd)
        bounds, bp = _d(bounds)
        discrete_all, dp = _d(discrete_all)
        P = discrete_all.reshape(self.num_mcmc, -1, self.d - num_fidelity).shape[1]
        Xq, qp = _d(Xq)
        q = Xq.reshape(-1, self.d).shape[0]
        if Xp is None or len(Xp) == 0:
            p, pp = 0, None
        else:
            Xp, pp = _d(Xp)
            p = Xp.reshape(-1, self.d).shape[0]
        best, bsp = _d(best_so_far)
        normals, npp = _d(normals)
        kg = C.c_double(0.0)
        grad = np.zeros(q * self.d)
        _check(lib().ref_kg_mcmc(self.h, num_fidelity, gdp, bp, dp, P, qp, pp, q, p, M, bsp, npp, normals.size,
                                 1 if want_grad else 0, C.byref(kg), grad.ctypes.data_as(_dp)))
        return kg.value, (grad.reshape(q, self.d) if want_grad else None)

    def ei(self, Xq, Xp, M, best_so_far, normals, want_grad=True):
        Xq, qp = _d(Xq)
        q = Xq.reshape(-1, self.d).shape[0]
        if Xp is None or len(Xp) == 0:
            p, pp = 0, None
        else:
            Xp, pp = _d(Xp)
            p = Xp.reshape(-1, self.d).shape[0]
        best, bsp = _d(best_so_far)
        normals, npp = _d(normals)
        assert normals.size >= M * (q + p)
        ei = C.c_double(0.0)
        grad = np.zeros(q * self.d)
        _check(lib().ref_ei_mcmc(self.h, qp, pp, q, p, M, bsp, npp, C.byref(ei), grad.ctypes.data_as(_dp) if want_grad else None))
        return ei.value, (grad.reshape(q, self.d) if want_grad else None)

    def ei_multistart_analytic(self, gd, bounds, starts, best_so_far):
        """ComputeEIMCMCOptimalPointsToSampleViaMultistartGradientDescent at q = 1, p = 0: (best_point [d], found)."""
        gd, gdp = _d(gd)
        bounds, bp = _d(bounds)
        starts, sp = _d(starts)
        S = starts.reshape(-1, self.d).shape[0]
        assert S >= 20, "the reference pops its top-20 queue unconditionally"
        best_so_far, bsp = _d(best_so_far)
        found = C.c_int(0)
        best = np.zeros(self.d)
        _check(lib().ref_ei_mcmc_multistart_analytic(self.h, gdp, bp, sp, S, bsp, C.byref(found), best.ctypes.data_as(_dp)))
        return best, bool(found.value)

    def kg_multistart(self, gd_outer, gd_inner, bounds, discrete_all, starts, Xp, M, best_so_far, seed, num_fidelity=0):
        """ComputeKGMCMCOptimalPointsToSampleViaMultistartGradientDescent (one thread) with NormalRNG(seed):
        (best_points [q,d], found).  discrete_all[num_mcmc][P][d - f]; best_so_far[num_mcmc]; starts[S][q][d], S >= 20."""
        gdo, gop = _d(gd_outer)
        gdi, gip = _d(gd_inner)
        bounds, bp = _d(bounds)
        inner_bounds = np.ascontiguousarray(bounds.reshape(-1)[: 2 * (self.d - num_fidelity)])
        discrete_all, dp = _d(discrete_all)
        P = discrete_all.reshape(self.num_mcmc, -1, self.d - num_fidelity).shape[1]
        starts = np.ascontiguousarray(starts, dtype=np.float64)
        S, q, _ = starts.shape
        assert S >= 20, "the reference pops its top-20 queue unconditionally"
        if Xp is None or len(Xp) == 0:
            p, pp = 0, None
        else:
            Xp_, pp = _d(Xp)
            p = Xp_.reshape(-1, self.d).shape[0]
        best_so_far, bsp = _d(best_so_far)
        found = C.c_int(0)
        best = np.zeros(q * self.d)
        _check(lib().ref_kg_mcmc_multistart(self.h, num_fidelity, gop, gip, bp, inner_bounds.ctypes.data_as(_dp), dp, P,
                                            starts.ctypes.data_as(_dp), S, pp, q, p, M, bsp, seed, C.byref(found),
                                            best.ctypes.data_as(_dp)))
        return best.reshape(q, self.d), bool(found.value)

    def kg_multistart_mt(self, gd_outer, gd_inner, bounds, discrete_all, starts, Xp, M, best_so_far, seed, num_threads, num_fidelity=0):
        """The same under `num_threads` OpenMP threads (the reference's own drivers pass 20): (best_points [q,d], found, wall seconds)."""
        gdo, gop = _d(gd_outer)
        gdi, gip = _d(gd_inner)
        bounds, bp = _d(bounds)
        inner_bounds = np.ascontiguousarray(bounds.reshape(-1)[: 2 * (self.d - num_fidelity)])
        discrete_all, dp = _d(discrete_all)
        P = discrete_all.reshape(self.num_mcmc, -1, self.d - num_fidelity).shape[1]
        starts = np.ascontiguousarray(starts, dtype=np.float64)
        S, q, _ = starts.shape
        assert S >= 20, "the reference pops its top-20 queue unconditionally"
        if Xp is None or len(Xp) == 0:
            p, pp = 0, None
        else:
            Xp_, pp = _d(Xp)
            p = Xp_.reshape(-1, self.d).shape[0]
        best_so_far, bsp = _d(best_so_far)
        found, wall = C.c_int(0), C.c_double(0.0)
        best = np.zeros(q * self.d)
        _check(lib().ref_kg_mcmc_multistart_mt(self.h, num_fidelity, gop, gip, bp, inner_bounds.ctypes.data_as(_dp), dp, P,
                                               starts.ctypes.data_as(_dp), S, pp, q, p, M, bsp, seed, int(num_threads), C.byref(found),
                                               best.ctypes.data_as(_dp), C.byref(wall)))
        return best.reshape(q, self.d), bool(found.value), wall.value
