"""ctypes binding of oracle/libmoe_oracle.so (the plain-C restatement, oracle/moe_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(cornell_moe_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libmoe_oracle.so")
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_lp = C.POINTER(C.c_long)
ORC_MAX_DIM = 64


class _Cov(C.Structure):
    _fields_ = [("type", C.c_int), ("dim", C.c_int), ("alpha", C.c_double), ("lengths_sq", C.c_double * ORC_MAX_DIM)]


def build():
    """(Re)build libmoe_oracle.so with gcc (needs no reference tree)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "moe_oracle.c")
        if not os.path.exists(_PATH) or os.path.getmtime(_PATH) < os.path.getmtime(src):
            build()
        L = C.CDLL(_PATH)
        L.orc_gp_create.restype = C.c_void_p
        L.orc_gp_create.argtypes = [C.c_int, C.c_double, _dp, _dp, _dp, _dp, _ip, C.c_int, C.c_int, C.c_int]
        L.orc_gp_destroy.argtypes = [C.c_void_p]
        L.orc_gp_N.argtypes = [C.c_void_p]
        L.orc_gp_dump.argtypes = [C.c_void_p, _dp, _dp, _dp]
        L.orc_gp_mix_cov.argtypes = [C.c_void_p, _dp, C.c_int, _ip, C.c_int, _dp]
        L.orc_gp_additional_mean.argtypes = [C.c_void_p, _dp, C.c_int, _ip, C.c_int, _dp]
        L.orc_gp_grad_additional_mean.argtypes = [C.c_void_p, _dp, C.c_int, _ip, C.c_int, _dp]
        for name in ("orc_gp_mean", "orc_gp_grad_mean", "orc_gp_var", "orc_gp_chol_var"):
            getattr(L, name).argtypes = [C.c_void_p, _dp, C.c_int, _dp]
        L.orc_gp_grad_var.argtypes = [C.c_void_p, _dp, C.c_int, C.c_int, _dp]
        L.orc_gp_grad_chol_var.argtypes = [C.c_void_p, _dp, C.c_int, C.c_int, _dp]
        L.orc_covariance.argtypes = [C.POINTER(_Cov), _dp, _ip, C.c_int, _dp, _ip, C.c_int, _dp]
        L.orc_grad_covariance.argtypes = [C.POINTER(_Cov), _dp, _ip, C.c_int, _dp, _ip, C.c_int, _dp]
        L.orc_cholesky.argtypes = [C.c_int, _dp]
        L.orc_chol_solve.argtypes = [_dp, C.c_int, _dp]
        L.orc_tri_solve.argtypes = [_dp, C.c_char, C.c_int, C.c_int, _dp]
        L.orc_ei.argtypes = [C.c_void_p, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_double, _dp, _dp, _dp]
        L.orc_log_likelihood.argtypes = [C.c_int, C.c_double, _dp, _dp, _dp, _dp, _ip, C.c_int, C.c_int, C.c_int, _dp]
        L.orc_ei_analytic.argtypes = [C.c_void_p, _dp, C.c_double, _dp, _dp]
        L.orc_kg.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, C.c_int, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_double,
                             _dp, C.c_int, _dp, _dp, _dp, _lp]
        L.orc_kg_head.argtypes = L.orc_kg.argtypes + [_dp]
        _lib = L
    return _lib


def _d(a):
    if a is None:
        return None, None
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _i(a):
    if a is None or len(a) == 0:
        return None, None
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


def _cov(cov_type, alpha, lengths):
    c = _Cov()
    c.type = cov_type
    c.dim = len(lengths)
    c.alpha = alpha
    for i, l in enumerate(lengths):
        c.lengths_sq[i] = float(l) * float(l)
    return c


def covariance(cov_type, alpha, lengths, p1, d1, p2, d2):
    L = lib()
    c = _cov(cov_type, alpha, lengths)
    dim, g1, g2 = len(p1), len(d1), len(d2)
    cov = np.zeros((1 + g1) * (1 + g2))
    gcov = np.zeros(dim * (1 + g1) * (1 + g2))
    a1, p1p = _d(p1)
    a2, p2p = _d(p2)
    i1, d1p = _i(list(d1))
    i2, d2p = _i(list(d2))
    L.orc_covariance(C.byref(c), p1p, d1p, g1, p2p, d2p, g2, cov.ctypes.data_as(_dp))
    L.orc_grad_covariance(C.byref(c), p1p, d1p, g1, p2p, d2p, g2, gcov.ctypes.data_as(_dp))
    return cov, gcov


def cholesky(a):
    a = np.array(a, dtype=np.float64)
    n = a.shape[0]
    flat = np.ascontiguousarray(a.T).ravel().copy()
    rc = lib().orc_cholesky(n, flat.ctypes.data_as(_dp))
    return rc, flat.reshape(n, n).T.copy()


def chol_solve(Lmat, b):
    Lmat = np.array(Lmat, dtype=np.float64)
    n = Lmat.shape[0]
    flat = np.ascontiguousarray(Lmat.T).ravel().copy()
    x = np.array(b, dtype=np.float64).copy()
    lib().orc_chol_solve(flat.ctypes.data_as(_dp), n, x.ctypes.data_as(_dp))
    return x


class SingularMatrix(RuntimeError):
    pass


class OrcGP(object):
    def __init__(self, cov_type, alpha, lengths, X, y, noise, derivs):
        L = lib()
        X = np.ascontiguousarray(X, dtype=np.float64)
        self.n, self.d = X.shape
        self.derivs = [int(v) for v in derivs]
        self.g = len(self.derivs)
        self.N = self.n * (1 + self.g)
        ya, yp = _d(y)
        na, np_ = _d(noise)
        la, lp = _d(lengths)
        da, dp = _i(self.derivs)
        self.h = L.orc_gp_create(cov_type, alpha, lp, X.ctypes.data_as(_dp), yp, np_, dp, self.g, self.d, self.n)
        if not self.h:
            raise SingularMatrix("K singular")

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().orc_gp_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def dump(self):
        K = np.zeros(self.N * self.N)
        kiy = np.zeros(self.N)
        mean = C.c_double(0.0)
        lib().orc_gp_dump(self.h, K.ctypes.data_as(_dp), kiy.ctypes.data_as(_dp), C.byref(mean))
        return K.reshape(self.N, self.N).T.copy(), kiy, mean.value

    def mix_cov(self, pts, derivs2=()):
        pts, pp = _d(pts)
        k = pts.reshape(-1, self.d).shape[0]
        i2, d2p = _i(list(derivs2))
        g2 = len(derivs2)
        out = np.zeros(self.N * k * (1 + g2))
        lib().orc_gp_mix_cov(self.h, pp, k, d2p, g2, out.ctypes.data_as(_dp))
        return out.reshape(k * (1 + g2), self.N).T.copy()

    def _q(self, fn, pts, size, *extra):
        pts, pp = _d(pts)
        k = pts.reshape(-1, self.d).shape[0]
        out = np.zeros(size(k))
        rc = fn(self.h, pp, k, *extra, out.ctypes.data_as(_dp))
        return out, rc

    def mean(self, pts):
        return self._q(lib().orc_gp_mean, pts, lambda k: k)[0]

    def additional_mean(self, pts, derivs2=()):
        i2, d2p = _i(list(derivs2))
        g2 = len(derivs2)
        return self._q(lib().orc_gp_additional_mean, pts, lambda k: k * (1 + g2), d2p, g2)[0]

    def grad_additional_mean(self, pts, derivs2=()):
        i2, d2p = _i(list(derivs2))
        g2 = len(derivs2)
        return self._q(lib().orc_gp_grad_additional_mean, pts, lambda k: self.d * k * (1 + g2), d2p, g2)[0]

    def grad_mean(self, pts):
        return self._q(lib().orc_gp_grad_mean, pts, lambda k: self.d * k * (1 + self.g))[0]

    def var(self, pts):
        return self._q(lib().orc_gp_var, pts, lambda k: (k * (1 + self.g)) ** 2)[0]

    def chol_var(self, pts):
        out, rc = self._q(lib().orc_gp_chol_var, pts, lambda k: (k * (1 + self.g)) ** 2)
        if rc:
            raise SingularMatrix("variance singular at minor %d" % rc)
        return out

    def grad_var(self, pts, nd):
        return self._q(lib().orc_gp_grad_var, pts, lambda k: self.d * (k * (1 + self.g)) ** 2 * nd, nd)[0]

    def grad_chol_var(self, pts, nd):
        out, rc = self._q(lib().orc_gp_grad_chol_var, pts, lambda k: self.d * (k * (1 + self.g)) ** 2 * nd, nd)
        if rc:
            raise SingularMatrix("variance singular at minor %d" % rc)
        return out

    def ei(self, Xq, Xp, M, best_so_far, normals, want_grad=True):
        Xq, qp = _d(Xq)
        q = Xq.reshape(-1, self.d).shape[0]
        if Xp is None or len(Xp) == 0:
            p, pp = 0, None
        else:
            Xp, pp = _d(Xp)
            p = Xp.reshape(-1, self.d).shape[0]
        normals, npp = _d(normals)
        ei = C.c_double(0.0)
        grad = np.zeros(q * self.d)
        rc = lib().orc_ei(self.h, qp, pp, q, p, M, best_so_far, npp, C.byref(ei),
                          grad.ctypes.data_as(_dp) if want_grad else None)
        if rc:
            raise SingularMatrix("variance singular at minor %d" % rc)
        return ei.value, (grad.reshape(q, self.d) if want_grad else None)

    def ei_analytic(self, pt, best_so_far, want_grad=True):
        pt, pp = _d(pt)
        ei = C.c_double(0.0)
        grad = np.zeros(self.d)
        lib().orc_ei_analytic(self.h, pp, best_so_far, C.byref(ei), grad.ctypes.data_as(_dp) if want_grad else None)
        return ei.value, (grad if want_grad else None)

    def kg(self, gd, bounds, discrete, Xq, Xp, M, best_so_far, normals, want_grad=True, num_fidelity=0, head=None):
        """head [q][d]: the points the evaluating state was BUILT at (the discretised set stays frozen there when the reference's
        multistart drivers move a state with SetCurrentPoint); None = a fresh state at Xq."""
        gd, gdp = _d(gd)
        bounds, bp = _d(bounds)
        discrete, dp = _d(discrete)
        P = discrete.reshape(-1, self.d - num_fidelity).shape[0]
        Xq, qp = _d(Xq)
        q = Xq.reshape(-1, self.d).shape[0]
        if Xp is None or len(Xp) == 0:
            p, pp = 0, None
        else:
            Xp, pp = _d(Xp)
            p = Xp.reshape(-1, self.d).shape[0]
        normals, npp = _d(normals)
        kg = C.c_double(0.0)
        grad = np.zeros(q * self.d)
        best_point = np.zeros(M * self.d)
        counters = (C.c_long * 2)()
        hp = None
        if head is not None:
            head, hp = _d(head)
        rc = lib().orc_kg_head(self.h, num_fidelity, gdp, bp, dp, P, qp, pp, q, p, M, best_so_far, npp, 1 if want_grad else 0,
                               C.byref(kg), grad.ctypes.data_as(_dp), best_point.ctypes.data_as(_dp), counters, hp)
        if rc:
            raise SingularMatrix("singular at minor %d" % rc)
        return dict(kg=kg.value, grad=grad.reshape(q, self.d) if want_grad else None,
                    best_point=best_point.reshape(M, self.d), mean_evals=counters[0], grad_evals=counters[1])


def log_likelihood(cov_type, alpha, lengths, X, y, noise, derivs):
    """orc_log_likelihood: log marginal likelihood of the data under the GP prior (gpp_model_selection.cpp:540-612)."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    n, d = X.shape
    derivs = [int(v) for v in derivs]
    ya, yp = _d(y)
    na, np_ = _d(noise)
    la, lp = _d(lengths)
    da, dp = _i(derivs)
    val = C.c_double(0.0)
    rc = lib().orc_log_likelihood(cov_type, float(alpha), lp, X.ctypes.data_as(_dp), yp, np_, dp, len(derivs), d, n, C.byref(val))
    if rc:
        raise SingularMatrix("K singular at minor %d" % rc)
    return val.value


def log_likelihood_grad(cov_type, alpha, lengths, X, y, noise, derivs):
    """orc_log_likelihood_grad: hyper-parameter gradient of the log marginal likelihood (gpp_model_selection.cpp:629-677).
    Returns [1 + d + 1 + g] partials wrt (alpha, lengths, noise variances)."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    n, d = X.shape
    derivs = [int(v) for v in derivs]
    ya, yp = _d(y)
    na, np_ = _d(noise)
    la, lp = _d(lengths)
    da, dp = _i(derivs)
    out = np.zeros(1 + d + 1 + len(derivs))
    fn = lib().orc_log_likelihood_grad
    fn.argtypes = [C.c_int, C.c_double, _dp, _dp, _dp, _dp, _ip, C.c_int, C.c_int, C.c_int, _dp]
    fn.restype = C.c_int
    rc = fn(cov_type, float(alpha), lp, X.ctypes.data_as(_dp), yp, np_, dp, len(derivs), d, n, out.ctypes.data_as(_dp))
    if rc:
        raise SingularMatrix("K singular at minor %d" % rc)
    return out


class OrcGPMCMC(object):
    """Restatement of the MCMC-averaged evaluators on top of OrcGP (numpy level): GaussianProcessMCMC builds one Matern-5/2 GP
    per hyper-parameter sample (gpp_knowledge_gradient_mcmc_optimization.cpp:24-49); the evaluators average the per-GP
    results and, for KG, divide by the fidelity cost (:84-180; gpp_expected_improvement_mcmc_optimization.cpp:48-88)."""

    def __init__(self, hypers, noises, X, y, derivs):
        X = np.ascontiguousarray(X, dtype=np.float64)
        self.n, self.d = X.shape
        self.derivs = [int(v) for v in derivs]
        self.g = len(self.derivs)
        hypers = np.asarray(hypers, dtype=np.float64).reshape(-1, self.d + 1)
        noises = np.asarray(noises, dtype=np.float64).reshape(-1, 1 + self.g)
        self.num_mcmc = hypers.shape[0]
        self.gps = [OrcGP(1, float(hypers[i, 0]), hypers[i, 1:], X, y, noises[i], self.derivs) for i in range(self.num_mcmc)]

    @staticmethod
    def cost(Xq, num_fidelity):
        """ComputeCost / ComputeGradCost (:84-127): the largest product of the fidelity coordinates over the q points."""
        Xq = np.asarray(Xq, dtype=np.float64)
        q, d = Xq.shape
        grad = np.zeros((q, d))
        if num_fidelity == 0:
            return 1.0, grad
        cost, index = 0.0, -1
        for i in range(q):
            pc = float(np.prod(Xq[i, d - num_fidelity:]))
            if cost < pc:
                cost, index = pc, i
        for j in range(d - num_fidelity, d):
            grad[index, j] = cost / Xq[index, j]
        return cost, grad

    def kg(self, gd, bounds, discrete_all, Xq, Xp, M, best_so_far, normals, want_grad=True, num_fidelity=0, head=None):
        Xq = np.asarray(Xq, dtype=np.float64).reshape(-1, self.d)
        discrete_all = np.asarray(discrete_all, dtype=np.float64).reshape(self.num_mcmc, -1, self.d - num_fidelity)
        kg, grad = 0.0, np.zeros_like(Xq)
        for i, gp in enumerate(self.gps):
            r = gp.kg(gd, bounds, discrete_all[i], Xq, Xp, M, float(best_so_far[i]), normals, want_grad=want_grad,
                      num_fidelity=num_fidelity, head=head)
            kg += r["kg"]
            if want_grad:
                grad += r["grad"]
        cost, gcost = self.cost(Xq, num_fidelity)
        kg_mean = kg / self.num_mcmc
        if not want_grad:
            return kg_mean / cost, None
        grad = (grad / self.num_mcmc * cost - kg_mean * gcost) / (cost * cost)
        return kg_mean / cost, grad

    def ei(self, Xq, Xp, M, best_so_far, normals, want_grad=True):
        Xq = np.asarray(Xq, dtype=np.float64).reshape(-1, self.d)
        ei, grad = 0.0, np.zeros_like(Xq)
        for i, gp in enumerate(self.gps):
            v, g = gp.ei(Xq, Xp, M, float(best_so_far[i]), normals, want_grad=want_grad)
            ei += v
            if want_grad:
                grad += g
        return ei / self.num_mcmc, (grad / self.num_mcmc if want_grad else None)

    def ei_analytic(self, pt, best_so_far, want_grad=True):
        """OnePotentialSampleExpectedImprovementMCMCEvaluator (gpp_expected_improvement_mcmc_optimization.cpp:136-176)."""
        ei, grad = 0.0, np.zeros(self.d)
        for i, gp in enumerate(self.gps):
            v, g = gp.ei_analytic(pt, float(best_so_far[i]), want_grad=want_grad)
            ei += v
            if want_grad:
                grad += g
        return ei / self.num_mcmc, (grad / self.num_mcmc if want_grad else None)
