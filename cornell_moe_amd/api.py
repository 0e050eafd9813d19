"""numpy-level wrappers over the C ABI (include/moe_hip.h).  Thin: argument marshalling + error mapping only.

Exceptions mirror the reference's Python-visible classes (gpp_python.cpp:71-124, 285-379):
OptimalLearningException, BoundsException(value, min, max), InvalidValueException(value, truth, tolerance),
SingularMatrixException(num_rows, leading_minor_index).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import dp, ip


class OptimalLearningException(Exception):
    pass


class BoundsException(OptimalLearningException):
    def __init__(self, message, value=0.0, min=0.0, max=0.0):  # noqa: A002  (names follow gpp_python.cpp:315-317)
        super(BoundsException, self).__init__(message)
        self.value, self.min, self.max = value, min, max


class InvalidValueException(OptimalLearningException):
    def __init__(self, message, value=0.0, truth=0.0, tolerance=0.0):
        super(InvalidValueException, self).__init__(message)
        self.value, self.truth, self.tolerance = value, truth, tolerance


class SingularMatrixException(OptimalLearningException):
    def __init__(self, message, num_rows=0, leading_minor_index=0):
        super(SingularMatrixException, self).__init__(message)
        self.num_rows, self.leading_minor_index = int(num_rows), int(leading_minor_index)


def _raise(err):
    msg = err.message.decode("utf-8", "replace")
    p = list(err.payload)
    if err.code == _lib.MOE_ERR_BOUNDS:
        raise BoundsException(msg, p[0], p[1], p[2])
    if err.code == _lib.MOE_ERR_INVALID_VALUE:
        raise InvalidValueException(msg, p[0], p[1], p[2])
    if err.code == _lib.MOE_ERR_SINGULAR:
        raise SingularMatrixException(msg, p[0], p[1])
    raise OptimalLearningException(msg)


def _check(rc, err):
    if rc != 0:
        _raise(err)


def _d(a):
    if a is None:
        return None, None
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(dp)


def fp64_rate(device=0):
    """moe_debug_fp64_rate: sustained FP64 FMA rate of the chip in TFLOP/s."""
    v = C.c_double(0.0)
    err = _lib.MoeError()
    _check(_lib.load().moe_debug_fp64_rate(int(device), C.byref(v), C.byref(err)), err)
    return v.value


def kxx_build_probe(log=None, device=0):
    """bench.py's `roofline_cov_build_kxx`: the GP's own covariance assembly where SURVEY 8(d) says it is the meaningful HBM
    measurement -- K(X, X) with derivative-observation rows at C5 (n = 2000, d = 12, g = 3: N = 8000, 512 MB) and at C5 with all
    12 partial derivatives observed (N = 26 000, 5.4 GB) -- against the 8 TB/s HBM peak."""
    from .workloads import make_workload
    out = {"bound": "hbm", "peak": 8000.0, "unit": "GB/s", "kernel": "cov_build_points_kernel (GpDev::rebuild's launch)", "cases": []}
    for derivs in ((0, 1, 2), tuple(range(12))):
        w = make_workload("C5", M=2, derivs=derivs)
        G = DeviceGP(w.hyperparameters, w.X, w.y, w.noise, w.derivs, device=device)
        ms, nbytes = G.kxx_build_probe(repeat=5 if len(derivs) > 3 else 20)
        gbs = nbytes / (ms * 1e-3) / 1e9
        out["cases"].append({"n": w.n, "d": w.d, "g": w.g, "N": w.n * (1 + w.g), "avg_launch_ms": ms, "bytes_per_launch": nbytes,
                             "achieved": gbs, "frac": gbs / 8000.0})
        if log:
            log("K(X,X) build N=%d: %.3f ms, %.0f GB/s" % (w.n * (1 + w.g), ms, gbs))
        G.close()
    out["achieved"] = out["cases"][0]["achieved"]
    out["frac"] = out["cases"][0]["frac"]
    return out


def pool_held_bytes():
    """moe_pool_held_bytes: device bytes the library's pool holds (released by destroyed objects, not yet reused)."""
    return int(_lib.load().moe_pool_held_bytes())


def pool_trim():
    """moe_pool_trim: give every pooled device / pinned buffer back to the runtime."""
    _lib.load().moe_pool_trim()


def multistart_trace():
    """moe_multistart_trace: rows (kind 0 values / 1 gradients, items, ms) of the last outer optimisation of this process."""
    L = _lib.load()
    n = L.moe_multistart_trace(None, 0)
    out = np.zeros((max(n, 1), 3))
    L.moe_multistart_trace(out.ctypes.data_as(dp), n)
    return out[:n]


def set_reference_quirks(on):
    """moe_set_reference_quirks: 1 = the multistart drivers reproduce the reference's execution, defects included (default);
    0 = the drivers as the reference intends them (fresh states, all q points move); -1 = follow MOE_REFERENCE_QUIRKS."""
    _lib.load().moe_set_reference_quirks(int(on))


def get_reference_quirks():
    return bool(_lib.load().moe_get_reference_quirks())


def set_ensemble_launches(on):
    """moe_set_ensemble_launches: 1 = the MCMC-averaged KG evaluators issue every kernel once for all ensemble members (default),
    0 = member by member, -1 = follow MOE_ENS_LAUNCH.  Same bits either way."""
    _lib.load().moe_set_ensemble_launches(int(on))


def ensemble_launch_stats():
    """(merged evaluations, member-by-member fall-backs, launches issued by merged evaluations, member launches they stand for)"""
    import ctypes
    out = (ctypes.c_longlong * 4)()
    _lib.load().moe_ensemble_launch_stats(out)
    return tuple(int(v) for v in out)


def normal_draws(seed, count):
    out = np.empty(int(count), dtype=np.float64)
    _lib.load().moe_normal_draws(C.c_uint(int(seed) & 0xFFFFFFFF), int(count), out.ctypes.data_as(dp))
    return out


def latin_hypercube(seed, bounds, num_points):
    """moe_latin_hypercube: [num_points][dim] points, one per slice of every edge (gpp_random.cpp:173-194)."""
    bounds, bp = _d(bounds)
    dim = bounds.size // 2
    out = np.zeros((int(num_points), dim))
    _lib.load().moe_latin_hypercube(C.c_uint(int(seed) & 0xFFFFFFFF), bp, dim, int(num_points), out.ctypes.data_as(dp))
    return out


def debug_cholesky(a, device=0):
    """Device Cholesky + inverse factor of an SPD matrix (parity probe); returns (L, Linv) as [row, col] arrays."""
    a = np.array(a, dtype=np.float64)
    n = a.shape[0]
    flat = np.ascontiguousarray(a.T).ravel()
    chol = np.zeros(n * n)
    inv = np.zeros(n * n)
    info = C.c_int(0)
    err = _lib.MoeError()
    _check(_lib.load().moe_debug_cholesky(n, flat.ctypes.data_as(dp), int(device), chol.ctypes.data_as(dp),
                                          inv.ctypes.data_as(dp), C.byref(info), C.byref(err)), err)
    return chol.reshape(n, n).T.copy(), inv.reshape(n, n).T.copy()


def debug_math(x, device=0):
    """(exp(-x), sqrt(x)) evaluated by the device fast-math routines, x >= 0."""
    x, xp = _d(x)
    e = np.zeros(x.size)
    r = np.zeros(x.size)
    err = _lib.MoeError()
    _check(_lib.load().moe_debug_math(x.size, xp, int(device), e.ctypes.data_as(dp), r.ctypes.data_as(dp), C.byref(err)), err)
    return e, r


class DeviceGP(object):
    """Device-resident GP: handle around moe_gp_t (replaces the reference's C_GP.GaussianProcess object)."""

    def __init__(self, hyperparameters, X, y, noise_variance, derivatives=(), cov_type=_lib.COV_MATERN_NU_2P5, device=0):
        L = _lib.load()
        X = np.ascontiguousarray(X, dtype=np.float64)
        if X.ndim != 2:
            raise BoundsException("points_sampled must be 2-D [num_sampled][dim]")
        self.n, self.d = X.shape
        self.derivatives = [int(v) for v in derivatives]
        self.g = len(self.derivatives)
        y = np.ascontiguousarray(y, dtype=np.float64).reshape(self.n, 1 + self.g)
        hyper, hp = _d(hyperparameters)
        noise, npn = _d(noise_variance)
        if hyper.size != 1 + self.d:
            raise InvalidValueException("hyperparameters must be [alpha, lengths...]", hyper.size, 1 + self.d, 0)
        if noise.size != 1 + self.g:
            raise InvalidValueException("noise_variance must have 1 + num_derivatives entries", noise.size, 1 + self.g, 0)
        dv = np.ascontiguousarray(self.derivatives, dtype=np.int32)
        self._h = C.c_void_p(None)
        err = _lib.MoeError()
        rc = L.moe_gp_create(hp, int(cov_type), X.ctypes.data_as(dp), y.ctypes.data_as(dp), npn,
                             dv.ctypes.data_as(ip) if self.g else None, self.g, self.d, self.n, int(device),
                             C.byref(self._h), C.byref(err))
        _check(rc, err)
        self.device = device

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.load().moe_gp_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- bookkeeping ----
    @property
    def N(self):
        return self.n * (1 + self.g)

    def add_points(self, pts, vals):
        pts, pp = _d(pts)
        vals, vp = _d(vals)
        k = pts.reshape(-1, self.d).shape[0]
        err = _lib.MoeError()
        try:
            _check(_lib.load().moe_gp_add_points(self._h, pp, vp, k, C.byref(err)), err)
        finally:  # a failed append rolls the handle back: always mirror what the device holds
            self.n = int(_lib.load().moe_gp_num_sampled(self._h))

    def get_factor(self):
        N = self.N
        K = np.zeros(N * N)
        kiy = np.zeros(N)
        mean = C.c_double(0.0)
        err = _lib.MoeError()
        _check(_lib.load().moe_gp_get_factor(self._h, K.ctypes.data_as(dp), kiy.ctypes.data_as(dp), C.byref(mean),
                                             C.byref(err)), err)
        return K.reshape(N, N).T.copy(), kiy, mean.value

    # ---- posterior queries (raw column-major outputs, flat) ----
    def _q(self, fn, pts, size, *extra):
        pts, pp = _d(pts)
        k = pts.reshape(-1, self.d).shape[0]
        out = np.zeros(size(k))
        err = _lib.MoeError()
        _check(fn(self._h, pp, k, *extra, out.ctypes.data_as(dp), C.byref(err)), err)
        return out

    def mean(self, pts):
        return self._q(_lib.load().moe_gp_mean, pts, lambda k: k)

    def additional_mean(self, pts):
        return self._q(_lib.load().moe_gp_additional_mean, pts, lambda k: k)

    def grad_mean(self, pts):
        return self._q(_lib.load().moe_gp_grad_mean, pts, lambda k: self.d * k * (1 + self.g))

    def variance(self, pts):
        return self._q(_lib.load().moe_gp_variance, pts, lambda k: (k * (1 + self.g)) ** 2)

    def cholesky_variance(self, pts):
        return self._q(_lib.load().moe_gp_cholesky_variance, pts, lambda k: (k * (1 + self.g)) ** 2)

    def grad_variance(self, pts, num_derivs):
        return self._q(_lib.load().moe_gp_grad_variance, pts, lambda k: self.d * (k * (1 + self.g)) ** 2 * num_derivs,
                       int(num_derivs))

    def grad_cholesky_variance(self, pts, num_derivs):
        return self._q(_lib.load().moe_gp_grad_cholesky_variance, pts,
                       lambda k: self.d * (k * (1 + self.g)) ** 2 * num_derivs, int(num_derivs))

    def mix_covariance(self, pts, derivs2=()):
        pts, pp = _d(pts)
        k = pts.reshape(-1, self.d).shape[0]
        g2 = len(derivs2)
        dv = np.ascontiguousarray(list(derivs2), dtype=np.int32)
        out = np.zeros(self.N * k * (1 + g2))
        err = _lib.MoeError()
        _check(_lib.load().moe_gp_mix_covariance(self._h, pp, k, dv.ctypes.data_as(ip) if g2 else None, g2,
                                                 out.ctypes.data_as(dp), C.byref(err)), err)
        return out.reshape(k * (1 + g2), self.N).T.copy()

    def cov_build_probe(self, pts, repeat=10):
        pts, pp = _d(pts)
        k = pts.reshape(-1, self.d).shape[0]
        ms, nbytes = C.c_double(0.0), C.c_double(0.0)
        err = _lib.MoeError()
        _check(_lib.load().moe_cov_build_probe(self._h, pp, k, int(repeat), C.byref(ms), C.byref(nbytes), C.byref(err)), err)
        return ms.value, nbytes.value

    def kxx_build_probe(self, repeat=10):
        """moe_kxx_build_probe: (average ms per launch, algorithmic bytes per launch) of the GP's own K(X, X) assembly."""
        ms, nbytes = C.c_double(0.0), C.c_double(0.0)
        err = _lib.MoeError()
        _check(_lib.load().moe_kxx_build_probe(self._h, int(repeat), C.byref(ms), C.byref(nbytes), C.byref(err)), err)
        return ms.value, nbytes.value

    def posterior_mean(self, point, num_fidelity=0, want_grad=True):
        point, pp = _d(point)
        val = C.c_double(0.0)
        grad = np.zeros(self.d - num_fidelity)
        err = _lib.MoeError()
        _check(_lib.load().moe_posterior_mean(self._h, int(num_fidelity), pp, C.byref(val),
                                              grad.ctypes.data_as(dp) if want_grad else None, C.byref(err)), err)
        return val.value, (grad if want_grad else None)

    # ---- acquisition functions ----
    def ei(self, Xq, Xp, num_mc, best_so_far, normals, want_grad=True, want_value=True):
        Xq, qp = _d(Xq)
        q = Xq.reshape(-1, self.d).shape[0]
        if Xp is None or np.size(Xp) == 0:
            p, ppp = 0, None
        else:
            Xp, ppp = _d(Xp)
            p = Xp.reshape(-1, self.d).shape[0]
        normals, npn = _d(normals)
        if normals.size < num_mc * (q + p):
            raise InvalidValueException("normal table too small", normals.size, num_mc * (q + p), 0)
        ei = C.c_double(0.0)
        grad = np.zeros(q * self.d)
        err = _lib.MoeError()
        _check(_lib.load().moe_ei(self._h, qp, ppp, q, p, int(num_mc), float(best_so_far), npn,
                                  C.byref(ei) if want_value else None, grad.ctypes.data_as(dp) if want_grad else None,
                                  C.byref(err)), err)
        return (ei.value if want_value else None), (grad.reshape(q, self.d) if want_grad else None)

    def ei_batch(self, Xq_all, Xp, num_mc, best_so_far, normals, want_grad=True):
        """moe_ei_batch: Xq_all [E][q][dim] -> (ei [E], grad [E][q][dim] or None)."""
        Xq_all = np.ascontiguousarray(Xq_all, dtype=np.float64)
        E, q, _ = Xq_all.shape
        if Xp is None or np.size(Xp) == 0:
            p, ppp = 0, None
        else:
            Xp, ppp = _d(Xp)
            p = Xp.reshape(-1, self.d).shape[0]
        normals, npn = _d(normals)
        if normals.size < num_mc * (q + p):
            raise InvalidValueException("normal table too small", normals.size, num_mc * (q + p), 0)
        ei = np.zeros(E)
        grad = np.zeros(E * q * self.d)
        err = _lib.MoeError()
        _check(_lib.load().moe_ei_batch(self._h, Xq_all.ctypes.data_as(dp), E, ppp, q, p, int(num_mc), float(best_so_far), npn,
                                        ei.ctypes.data_as(dp), grad.ctypes.data_as(dp) if want_grad else None, C.byref(err)),
               err)
        return ei, (grad.reshape(E, q, self.d) if want_grad else None)

    def ei_analytic_batch(self, points, best_so_far, want_grad=True):
        """moe_ei_analytic_batch: points [E][dim] -> (ei [E], grad [E][dim] or None)."""
        points = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, self.d)
        E = points.shape[0]
        ei = np.zeros(E)
        grad = np.zeros((E, self.d))
        err = _lib.MoeError()
        _check(_lib.load().moe_ei_analytic_batch(self._h, points.ctypes.data_as(dp), E, float(best_so_far),
                                                 ei.ctypes.data_as(dp), grad.ctypes.data_as(dp) if want_grad else None,
                                                 C.byref(err)), err)
        return ei, (grad if want_grad else None)

    def ei_multistart(self, outer_params, bounds, starts, Xp, num_mc, best_so_far, normals, gradient_ascent=True):
        """moe_ei_multistart: starts [S][q][dim] -> (best_points [q][dim], best_ei, found)."""
        go = self._gd(outer_params)
        bounds, bp = _d(bounds)
        starts = np.ascontiguousarray(starts, dtype=np.float64)
        S, q, _ = starts.shape
        if Xp is None or np.size(Xp) == 0:
            p, ppp = 0, None
        else:
            Xp, ppp = _d(Xp)
            p = Xp.reshape(-1, self.d).shape[0]
        npn = None
        if normals is not None:
            normals, npn = _d(normals)
            if normals.size < num_mc * (q + p):
                raise InvalidValueException("normal table too small", normals.size, num_mc * (q + p), 0)
        best = np.zeros(q * self.d)
        best_ei = C.c_double(0.0)
        found = C.c_int(0)
        err = _lib.MoeError()
        _check(_lib.load().moe_ei_multistart(self._h, C.byref(go), bp, starts.ctypes.data_as(dp), S, ppp, q, p, int(num_mc),
                                             float(best_so_far), npn, 1 if gradient_ascent else 0, best.ctypes.data_as(dp),
                                             C.byref(best_ei), C.byref(found), C.byref(err)), err)
        return best.reshape(q, self.d), best_ei.value, bool(found.value)

    @staticmethod
    def _gd(params):
        if isinstance(params, _lib.GdParams):
            return params
        g = _lib.GdParams()
        (g.num_multistarts, g.max_num_steps, g.max_num_restarts, g.num_steps_averaged) = [int(v) for v in params[:4]]
        (g.gamma, g.pre_mult, g.max_relative_change, g.tolerance) = [float(v) for v in params[4:8]]
        g.domain_type = int(params[8]) if len(params) > 8 else 0   # (outer optimisers: 1 = simplex intersection, moe_hip.h)
        return g

    def kg(self, inner_params, bounds, discrete, Xq, Xp, num_mc, best_so_far, normals, want_grad=True, num_fidelity=0,
           first_sample=0, num_local=None, want_best_points=False):
        """One KG evaluation (or an even-aligned MC shard of it).  Returns dict(kg_sum, grad_sum, kg, grad, stats, ...);
        `kg`/`grad` are the normalised values assuming this call covered all num_mc samples."""
        L = _lib.load()
        gd = self._gd(inner_params)
        bounds, bp = _d(bounds)
        discrete, dpp = _d(discrete)
        if not 0 <= num_fidelity < self.d:
            raise BoundsException("num_fidelity out of range", num_fidelity, 0, self.d - 1)
        P = discrete.reshape(-1, self.d - num_fidelity).shape[0]
        Xq, qp = _d(Xq)
        q = Xq.reshape(-1, self.d).shape[0]
        if Xp is None or np.size(Xp) == 0:
            p, ppp = 0, None
        else:
            Xp, ppp = _d(Xp)
            p = Xp.reshape(-1, self.d).shape[0]
        m = (q + p) * (1 + self.g)
        normals, npn = _d(normals)
        if normals.size < ((num_mc + 1) // 2) * m:
            raise InvalidValueException("normal table too small", normals.size, ((num_mc + 1) // 2) * m, 0)
        if num_local is None:
            num_local = num_mc - first_sample
        kg_sum = C.c_double(0.0)
        grad = np.zeros(q * self.d)
        best = np.zeros(num_local * self.d) if want_best_points else None
        stats = _lib.KgStats()
        err = _lib.MoeError()
        rc = L.moe_kg(self._h, int(num_fidelity), C.byref(gd), bp, dpp, P, qp, ppp, q, p, int(num_mc), float(best_so_far),
                      npn, int(first_sample), int(num_local), 1 if want_grad else 0, C.byref(kg_sum),
                      grad.ctypes.data_as(dp), best.ctypes.data_as(dp) if want_best_points else None, C.byref(stats),
                      C.byref(err))
        _check(rc, err)
        out = dict(kg_sum=kg_sum.value, grad_sum=grad.reshape(q, self.d) if want_grad else None,
                   kg=kg_sum.value / num_mc, grad=(grad.reshape(q, self.d) / num_mc) if want_grad else None,
                   mean_evals=stats.posterior_mean_evals, grad_evals=stats.posterior_grad_evals,
                   ms_state=stats.ms_state, ms_mc=stats.ms_mc, ms_tail=stats.ms_tail)
        if want_best_points:
            out["best_point"] = best.reshape(num_local, self.d)
        return out

    def kg_batch(self, inner_params, bounds, discrete, Xq_all, Xp, num_mc, best_so_far, normals, want_grad=True,
                 num_fidelity=0, first_sample=0, num_local=None):
        L = _lib.load()
        gd = self._gd(inner_params)
        bounds, bp = _d(bounds)
        discrete, dpp = _d(discrete)
        if not 0 <= num_fidelity < self.d:
            raise BoundsException("num_fidelity out of range", num_fidelity, 0, self.d - 1)
        P = discrete.reshape(-1, self.d - num_fidelity).shape[0]
        Xq_all = np.ascontiguousarray(Xq_all, dtype=np.float64)
        R, q, _ = Xq_all.shape
        if Xp is None or np.size(Xp) == 0:
            p, ppp = 0, None
        else:
            Xp, ppp = _d(Xp)
            p = Xp.reshape(-1, self.d).shape[0]
        normals, npn = _d(normals)
        if num_local is None:
            num_local = num_mc - first_sample
        kg_sum = np.zeros(R)
        grad = np.zeros(R * q * self.d)
        stats = _lib.KgStats()
        err = _lib.MoeError()
        rc = L.moe_kg_batch(self._h, int(num_fidelity), C.byref(gd), bp, dpp, P, Xq_all.ctypes.data_as(dp), R, ppp, q, p,
                            int(num_mc), float(best_so_far), npn, int(first_sample), int(num_local),
                            1 if want_grad else 0, kg_sum.ctypes.data_as(dp), grad.ctypes.data_as(dp), C.byref(stats),
                            C.byref(err))
        _check(rc, err)
        return dict(kg_sum=kg_sum, grad_sum=grad.reshape(R, q, self.d), mean_evals=stats.posterior_mean_evals,
                    grad_evals=stats.posterior_grad_evals, ms_state=stats.ms_state, ms_mc=stats.ms_mc, ms_tail=stats.ms_tail)

    def kg_multistart(self, outer_params, inner_params, bounds, discrete, starts, Xp, num_mc, best_so_far, normals,
                      gradient_ascent=True, num_fidelity=0, comm=None):
        """moe_kg_multistart: starts [S][q][dim] -> (best_points [q][dim], best_kg, found).  comm (r5: a dist.Exchange, or None):
        moe_kg_multistart_comm -- the restarts of every batched evaluation are dealt to the ranks, one all-gather each; every rank
        returns the single-rank result bit for bit."""
        L = _lib.load()
        go, gi = self._gd(outer_params), self._gd(inner_params)
        bounds, bp = _d(bounds)
        discrete, dpp = _d(discrete)
        if not 0 <= num_fidelity < self.d:
            raise BoundsException("num_fidelity out of range", num_fidelity, 0, self.d - 1)
        P = discrete.reshape(-1, self.d - num_fidelity).shape[0]
        starts = np.ascontiguousarray(starts, dtype=np.float64)
        S, q, _ = starts.shape
        if Xp is None or np.size(Xp) == 0:
            p, ppp = 0, None
        else:
            Xp, ppp = _d(Xp)
            p = Xp.reshape(-1, self.d).shape[0]
        normals, npn = _d(normals)
        best = np.zeros(q * self.d)
        best_kg = C.c_double(0.0)
        found = C.c_int(0)
        err = _lib.MoeError()
        if comm is not None and comm.world > 1:
            rc = L.moe_kg_multistart_comm(self._h, C.byref(comm.c_struct), int(num_fidelity), C.byref(go), C.byref(gi), bp, dpp, P,
                                          starts.ctypes.data_as(dp), S, ppp, q, p, int(num_mc), float(best_so_far), npn,
                                          1 if gradient_ascent else 0, best.ctypes.data_as(dp), C.byref(best_kg), C.byref(found),
                                          C.byref(err))
            comm.reraise()  # (an exception inside the exchange callback cannot cross the C frames: it is kept and raised here)
            _check(rc, err)
        else:
            _check(L.moe_kg_multistart(self._h, int(num_fidelity), C.byref(go), C.byref(gi), bp, dpp, P, starts.ctypes.data_as(dp), S,
                                       ppp, q, p, int(num_mc), float(best_so_far), npn, 1 if gradient_ascent else 0,
                                       best.ctypes.data_as(dp), C.byref(best_kg), C.byref(found), C.byref(err)), err)
        return best.reshape(q, self.d), best_kg.value, bool(found.value)

    def posterior_mean_optimize(self, params, bounds, initial_guess, num_fidelity=0):
        g = self._gd(params)
        bounds, bp = _d(bounds)
        x0, xp = _d(initial_guess)
        out = np.zeros(self.d - num_fidelity)
        val = C.c_double(0.0)
        err = _lib.MoeError()
        _check(_lib.load().moe_posterior_mean_optimize(self._h, int(num_fidelity), C.byref(g), bp, xp, out.ctypes.data_as(dp),
                                                       C.byref(val), C.byref(err)), err)
        return out, val.value

    def last_kernel_ms(self):
        out = np.zeros(5)
        _lib.load().moe_last_kernel_ms(self._h, out.ctypes.data_as(dp))
        return dict(mc=out[0], cov_build=out[1], tail=out[2], state=out[3], total=out[4])


    def last_kernel_info(self):
        out = (C.c_int * 8)()
        _lib.load().moe_last_kernel_info(self._h, out)
        keys = ("variant", "xlds", "waves", "tr", "weight_table", "fused_tail", "blocks", "prep")
        info = dict(zip(keys, [int(v) for v in out]))
        info["far_frame"], info["wide_frame"], info["lane"] = (info["xlds"] >> 1) & 1, (info["xlds"] >> 2) & 1, (info["xlds"] >> 3) & 1
        info["xlds"] &= 1
        return info


def kg_batch_multi(gps, shard, inner_params, bounds, discrete, Xq_all, Xp, num_mc, best_so_far, normals, want_grad=True,
                   num_fidelity=0):
    """moe_kg_batch_multi: the same GP on several devices (`gps`: DeviceGP objects built with device = 0 .. W-1), one host
    thread per device inside the library; shard = "restarts" (evaluation e on handle e % W) or "mc" (every handle takes an
    even-aligned slice of the samples, host-side fixed-order sum).  Returns the dict kg_batch returns."""
    L = _lib.load()
    g0 = gps[0]
    gd = DeviceGP._gd(inner_params)
    bounds, bp = _d(bounds)
    discrete, dpp = _d(discrete)
    P = discrete.reshape(-1, g0.d - num_fidelity).shape[0]
    Xq_all = np.ascontiguousarray(Xq_all, dtype=np.float64)
    R, q, _ = Xq_all.shape
    if Xp is None or np.size(Xp) == 0:
        p, ppp = 0, None
    else:
        Xp, ppp = _d(Xp)
        p = Xp.reshape(-1, g0.d).shape[0]
    normals, npn = _d(normals)
    arr = (C.c_void_p * len(gps))(*[g._h.value for g in gps])
    kg = np.zeros(R)
    grad = np.zeros(R * q * g0.d)
    stats = _lib.KgStats()
    err = _lib.MoeError()
    _check(L.moe_kg_batch_multi(arr, len(gps), {"restarts": 0, "mc": 1}[shard], int(num_fidelity), C.byref(gd), bp, dpp, P,
                                Xq_all.ctypes.data_as(dp), R, ppp, q, p, int(num_mc), float(best_so_far), npn,
                                1 if want_grad else 0, kg.ctypes.data_as(dp), grad.ctypes.data_as(dp), C.byref(stats),
                                C.byref(err)), err)
    return dict(kg_sum=kg, grad_sum=grad.reshape(R, q, g0.d) if want_grad else None, mean_evals=stats.posterior_mean_evals,
                grad_evals=stats.posterior_grad_evals)


class DeviceGPMCMC(object):
    """num_mcmc device GPs over the same data, one per hyper-parameter sample (replaces C_GP.GaussianProcessMCMC,
    gpp_knowledge_gradient_mcmc_optimization.cpp:24-49: Matern-5/2, hypers [num_mcmc][1 + dim] = (alpha, lengths),
    noises [num_mcmc][1 + num_derivatives]).  `members` restricts construction to a subset of the GP indices (the GP-index
    shard of a multi-GPU run); the MCMC-averaged evaluators below then return that subset's share."""

    def __init__(self, hypers, noises, X, y, derivatives=(), device=0, members=None):
        X = np.ascontiguousarray(X, dtype=np.float64)
        self.n, self.d = X.shape
        self.derivatives = [int(v) for v in derivatives]
        self.g = len(self.derivatives)
        hypers = np.ascontiguousarray(hypers, dtype=np.float64).reshape(-1, self.d + 1)
        noises = np.ascontiguousarray(noises, dtype=np.float64).reshape(-1, 1 + self.g)
        self.total_num_mcmc = hypers.shape[0]
        self.members = list(range(self.total_num_mcmc)) if members is None else [int(i) for i in members]
        self.gps = [DeviceGP(hypers[i], X, y, noises[i], self.derivatives, cov_type=_lib.COV_MATERN_NU_2P5, device=device)
                    for i in self.members]
        self._arr = (C.c_void_p * max(len(self.gps), 1))(*[gp._h.value for gp in self.gps])

    num_mcmc = property(lambda self: len(self.gps))

    def _common(self, Xq_all, Xp):
        Xq_all = np.ascontiguousarray(Xq_all, dtype=np.float64)
        E, q, _ = Xq_all.shape
        if Xp is None or np.size(Xp) == 0:
            p, ppp = 0, None
        else:
            Xp, ppp = _d(Xp)
            p = Xp.reshape(-1, self.d).shape[0]
        return Xq_all, E, q, Xp, p, ppp

    def _local(self, per_gp):
        """Rows of a per-GP array [total_num_mcmc][...] that belong to this object's members."""
        a = np.asarray(per_gp, dtype=np.float64).reshape(self.total_num_mcmc, -1)
        return np.ascontiguousarray(a[self.members])

    def kg_batch(self, inner_params, bounds, discrete_all, Xq_all, Xp, num_mc, best_so_far, normals, want_grad=True,
                 num_fidelity=0, finalize=True):
        """moe_kg_mcmc_batch: Xq_all [E][q][dim], discrete_all [total_num_mcmc][P][dim - f], best_so_far [total_num_mcmc]
        -> (kg [E], grad [E][q][dim] or None).  finalize=False returns this object's members' plain sums."""
        Xq_all, E, q, Xp, p, ppp = self._common(Xq_all, Xp)
        g = DeviceGP._gd(inner_params)
        bounds, bp = _d(bounds)
        disc = self._local(discrete_all)
        if not 0 <= num_fidelity < self.d:
            raise BoundsException("num_fidelity out of range", num_fidelity, 0, self.d - 1)
        P = disc.shape[1] // (self.d - num_fidelity)
        best = self._local(best_so_far).ravel()
        normals, npn = _d(normals)
        m = (q + p) * (1 + self.g)
        if normals.size < ((num_mc + 1) // 2) * m:
            raise InvalidValueException("normal table too small", normals.size, ((num_mc + 1) // 2) * m, 0)
        kg = np.zeros(E)
        grad = np.zeros((E, q, self.d))
        err = _lib.MoeError()
        _check(_lib.load().moe_kg_mcmc_batch(self._arr, len(self.gps), int(num_fidelity), C.byref(g), bp, disc.ctypes.data_as(dp),
                                             P, Xq_all.ctypes.data_as(dp), E, ppp, q, p, int(num_mc), best.ctypes.data_as(dp),
                                             npn, 1 if finalize else 0, self.total_num_mcmc, kg.ctypes.data_as(dp),
                                             grad.ctypes.data_as(dp) if want_grad else None, C.byref(err)), err)
        return kg, (grad if want_grad else None)

    def kg_finalize(self, kg_sum, grad_sum, Xq_all, num_fidelity=0):
        """moe_kg_mcmc_finalize on all-reduced sums: mean over total_num_mcmc, fidelity cost and its gradient term."""
        Xq_all = np.ascontiguousarray(Xq_all, dtype=np.float64)
        E, q, _ = Xq_all.shape
        kg = np.array(kg_sum, dtype=np.float64, copy=True).reshape(E)
        grad = None if grad_sum is None else np.array(grad_sum, dtype=np.float64, copy=True).reshape(E, q, self.d)
        rc = _lib.load().moe_kg_mcmc_finalize(kg.ctypes.data_as(dp), grad.ctypes.data_as(dp) if grad is not None else None,
                                              Xq_all.ctypes.data_as(dp), E, q, self.d, int(num_fidelity), self.total_num_mcmc)
        if rc:
            raise BoundsException("moe_kg_mcmc_finalize: bad argument", rc, 0, 0)
        return kg, grad

    def ei_batch(self, Xq_all, Xp, num_mc, best_so_far, normals, want_grad=True, analytic=False):
        """moe_ei_mcmc_batch: (ei [E], grad [E][q][dim] or None), averaged over this object's members."""
        Xq_all, E, q, Xp, p, ppp = self._common(Xq_all, Xp)
        best = self._local(best_so_far).ravel()
        npn = None
        if not analytic:
            normals, npn = _d(normals)
            if normals.size < num_mc * (q + p):
                raise InvalidValueException("normal table too small", normals.size, num_mc * (q + p), 0)
        ei = np.zeros(E)
        grad = np.zeros((E, q, self.d))
        err = _lib.MoeError()
        _check(_lib.load().moe_ei_mcmc_batch(self._arr, len(self.gps), Xq_all.ctypes.data_as(dp), E, ppp, q, p, int(num_mc),
                                             best.ctypes.data_as(dp), npn, 1 if analytic else 0, ei.ctypes.data_as(dp),
                                             grad.ctypes.data_as(dp) if want_grad else None, C.byref(err)), err)
        return ei, (grad if want_grad else None)

    def kg_multistart(self, outer_params, inner_params, bounds, discrete_all, starts, Xp, num_mc, best_so_far, normals,
                      gradient_ascent=True, num_fidelity=0, comm=None):
        """moe_kg_mcmc_multistart: starts [S][q][dim] -> (best_points [q][dim], best_kg, found).  comm (r5: a dist.Exchange):
        moe_kg_mcmc_multistart_comm -- this object holds members rank, rank + world, ... (DeviceGPMCMC(members=dist.shard_members(...)));
        discrete_all / best_so_far are the WHOLE ensemble's, the local rows are picked here; the per-member values of every batched
        evaluation are exchanged and added in global member order: the single-rank result bit for bit."""
        starts, S, q, Xp, p, ppp = self._common(starts, Xp)
        go, gi = DeviceGP._gd(outer_params), DeviceGP._gd(inner_params)
        bounds, bp = _d(bounds)
        disc = self._local(discrete_all)
        if not 0 <= num_fidelity < self.d:
            raise BoundsException("num_fidelity out of range", num_fidelity, 0, self.d - 1)
        P = disc.shape[1] // (self.d - num_fidelity)
        best = self._local(best_so_far).ravel()
        normals, npn = _d(normals)
        out = np.zeros(q * self.d)
        val = C.c_double(0.0)
        found = C.c_int(0)
        err = _lib.MoeError()
        if comm is not None and comm.world > 1:
            if self.members != list(range(comm.rank, self.total_num_mcmc, comm.world)):
                raise InvalidValueException("member-sharded optimisation: this rank must hold members rank, rank + world, ...",
                                            len(self.members), 0, 0)
            rc = _lib.load().moe_kg_mcmc_multistart_comm(self._arr, len(self.gps), self.total_num_mcmc, C.byref(comm.c_struct),
                                                         int(num_fidelity), C.byref(go), C.byref(gi), bp, disc.ctypes.data_as(dp), P,
                                                         starts.ctypes.data_as(dp), S, ppp, q, p, int(num_mc),
                                                         best.ctypes.data_as(dp), npn, 1 if gradient_ascent else 0,
                                                         out.ctypes.data_as(dp), C.byref(val), C.byref(found), C.byref(err))
            comm.reraise()
            _check(rc, err)
            return out.reshape(q, self.d), val.value, bool(found.value)
        _check(_lib.load().moe_kg_mcmc_multistart(self._arr, len(self.gps), int(num_fidelity), C.byref(go), C.byref(gi), bp,
                                                  disc.ctypes.data_as(dp), P, starts.ctypes.data_as(dp), S, ppp, q, p, int(num_mc),
                                                  best.ctypes.data_as(dp), npn, 1 if gradient_ascent else 0,
                                                  out.ctypes.data_as(dp), C.byref(val), C.byref(found), C.byref(err)), err)
        return out.reshape(q, self.d), val.value, bool(found.value)

    def ei_multistart(self, outer_params, bounds, starts, Xp, num_mc, best_so_far, normals, gradient_ascent=True):
        """moe_ei_mcmc_multistart: starts [S][q][dim] -> (best_points [q][dim], best_ei, found)."""
        starts, S, q, Xp, p, ppp = self._common(starts, Xp)
        go = DeviceGP._gd(outer_params)
        bounds, bp = _d(bounds)
        best = self._local(best_so_far).ravel()
        npn = None
        if normals is not None:
            normals, npn = _d(normals)
        out = np.zeros(q * self.d)
        val = C.c_double(0.0)
        found = C.c_int(0)
        err = _lib.MoeError()
        _check(_lib.load().moe_ei_mcmc_multistart(self._arr, len(self.gps), C.byref(go), bp, starts.ctypes.data_as(dp), S, ppp, q, p,
                                                  int(num_mc), best.ctypes.data_as(dp), npn, 1 if gradient_ascent else 0,
                                                  out.ctypes.data_as(dp), C.byref(val), C.byref(found), C.byref(err)), err)
        return out.reshape(q, self.d), val.value, bool(found.value)


def kg_multistart_multi(gps, outer_params, inner_params, bounds, discrete, starts, Xp, num_mc, best_so_far, normals,
                        gradient_ascent=True, num_fidelity=0):
    """moe_kg_multistart_multi (r5): the outer optimiser with its restarts dealt to `gps` -- DeviceGP objects holding the same GP on
    devices 0 .. W-1 -- one host thread per handle, the exchange in shared memory.  (best_points [q][dim], best_kg, found): bit for
    bit DeviceGP.kg_multistart's."""
    g0 = gps[0]
    go, gi = DeviceGP._gd(outer_params), DeviceGP._gd(inner_params)
    bounds, bp = _d(bounds)
    discrete, dpp = _d(discrete)
    P = discrete.reshape(-1, g0.d - num_fidelity).shape[0]
    starts = np.ascontiguousarray(starts, dtype=np.float64)
    S, q, _ = starts.shape
    if Xp is None or np.size(Xp) == 0:
        p, ppp = 0, None
    else:
        Xp, ppp = _d(Xp)
        p = Xp.reshape(-1, g0.d).shape[0]
    normals, npn = _d(normals)
    arr = (C.c_void_p * len(gps))(*[g._h.value for g in gps])
    best = np.zeros(q * g0.d)
    val, found, err = C.c_double(0.0), C.c_int(0), _lib.MoeError()
    _check(_lib.load().moe_kg_multistart_multi(arr, len(gps), int(num_fidelity), C.byref(go), C.byref(gi), bp, dpp, P,
                                               starts.ctypes.data_as(dp), S, ppp, q, p, int(num_mc), float(best_so_far), npn,
                                               1 if gradient_ascent else 0, best.ctypes.data_as(dp), C.byref(val), C.byref(found),
                                               C.byref(err)), err)
    return best.reshape(q, g0.d), val.value, bool(found.value)


def kg_mcmc_multistart_multi(mcmc, num_workers, outer_params, inner_params, bounds, discrete_all, starts, Xp, num_mc, best_so_far,
                             normals, gradient_ascent=True, num_fidelity=0):
    """moe_kg_mcmc_multistart_multi (r5): `mcmc` = a DeviceGPMCMC holding the WHOLE ensemble (member g built on device
    g % num_workers by the caller); worker k -- a host thread -- takes members k, k + num_workers, ...  Result: bit for bit
    DeviceGPMCMC.kg_multistart's."""
    starts, S, q, Xp, p, ppp = mcmc._common(starts, Xp)
    go, gi = DeviceGP._gd(outer_params), DeviceGP._gd(inner_params)
    bounds, bp = _d(bounds)
    disc = mcmc._local(discrete_all)
    P = disc.shape[1] // (mcmc.d - num_fidelity)
    best = mcmc._local(best_so_far).ravel()
    normals, npn = _d(normals)
    out = np.zeros(q * mcmc.d)
    val, found, err = C.c_double(0.0), C.c_int(0), _lib.MoeError()
    _check(_lib.load().moe_kg_mcmc_multistart_multi(mcmc._arr, len(mcmc.gps), int(num_workers), int(num_fidelity), C.byref(go),
                                                    C.byref(gi), bp, disc.ctypes.data_as(dp), P, starts.ctypes.data_as(dp), S, ppp,
                                                    q, p, int(num_mc), best.ctypes.data_as(dp), npn, 1 if gradient_ascent else 0,
                                                    out.ctypes.data_as(dp), C.byref(val), C.byref(found), C.byref(err)), err)
    return out.reshape(q, mcmc.d), val.value, bool(found.value)


class LogLikelihood(object):
    """Log marginal likelihood of fixed data under varying hyper-parameters (moe_ll_*): the evaluator a hyper-parameter
    sampler calls thousands of times (LogMarginalLikelihoodEvaluator, gpp_model_selection.cpp:540-612)."""

    def __init__(self, X, y, derivatives=(), cov_type=_lib.COV_MATERN_NU_2P5, device=0):
        X = np.ascontiguousarray(X, dtype=np.float64)
        self.n, self.d = X.shape
        self.derivatives = [int(v) for v in derivatives]
        self.g = len(self.derivatives)
        y = np.ascontiguousarray(y, dtype=np.float64).reshape(self.n, 1 + self.g)
        dv = np.ascontiguousarray(self.derivatives, dtype=np.int32)
        self._h = C.c_void_p(None)
        err = _lib.MoeError()
        _check(_lib.load().moe_ll_create(int(cov_type), X.ctypes.data_as(dp), y.ctypes.data_as(dp),
                                         dv.ctypes.data_as(ip) if self.g else None, self.g, self.d, self.n, int(device),
                                         C.byref(self._h), C.byref(err)), err)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.load().moe_ll_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def evaluate(self, hyperparameters_all):
        """hyperparameters_all [num_sets][1 + dim + 1 + g] = (alpha, lengths, noise variances) -> values [num_sets]
        (-inf where K + noise is singular)."""
        h = np.ascontiguousarray(hyperparameters_all, dtype=np.float64).reshape(-1, 1 + self.d + 1 + self.g)
        out = np.zeros(h.shape[0])
        err = _lib.MoeError()
        _check(_lib.load().moe_ll_evaluate(self._h, h.ctypes.data_as(dp), h.shape[0], out.ctypes.data_as(dp), C.byref(err)), err)
        return out

    def multistart(self, gd_params, domain_log10, initial_guesses):
        """moe_ll_multistart: maximum-likelihood hyper-parameters by restarted gradient ascent from initial_guesses [S][1 + dim + 1 + g]
        (linear space) inside domain_log10 [n_hyper][2] (log-10 space) -> (best [n_hyper], best log likelihood, found)."""
        nh = 1 + self.d + 1 + self.g
        g = DeviceGP._gd(gd_params)
        dom, dpp = _d(domain_log10)
        x0 = np.ascontiguousarray(initial_guesses, dtype=np.float64).reshape(-1, nh)
        out = np.zeros(nh)
        val, found, err = C.c_double(0.0), C.c_int(0), _lib.MoeError()
        _check(_lib.load().moe_ll_multistart(self._h, C.byref(g), dpp, x0.ctypes.data_as(dp), x0.shape[0], out.ctypes.data_as(dp),
                                             C.byref(val), C.byref(found), C.byref(err)), err)
        return out, val.value, bool(found.value)

    def ascend(self, gd_params, domain_log10, x0):
        """moe_ll_ascend: the end point of the restarted gradient ascent from x0 [1 + dim + 1 + g] (linear space)."""
        nh = 1 + self.d + 1 + self.g
        g = DeviceGP._gd(gd_params)
        dom, dpp = _d(domain_log10)
        x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(nh)
        out = np.zeros(nh)
        err = _lib.MoeError()
        _check(_lib.load().moe_ll_ascend(self._h, C.byref(g), dpp, x0.ctypes.data_as(dp), out.ctypes.data_as(dp), C.byref(err)), err)
        return out

    def grad(self, hyperparameters):
        """d log p / d (alpha, lengths, noise variances) at one hyper-parameter set [1 + dim + 1 + g] (moe_ll_grad)."""
        h = np.ascontiguousarray(hyperparameters, dtype=np.float64).reshape(1 + self.d + 1 + self.g)
        out = np.zeros(h.size)
        err = _lib.MoeError()
        _check(_lib.load().moe_ll_grad(self._h, h.ctypes.data_as(dp), out.ctypes.data_as(dp), C.byref(err)), err)
        return out
