"""ctypes binding of libmoe_hip.so (include/moe_hip.h).  There is NO fallback: a missing library or a missing GPU raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MOE_LIB_PATH") or os.path.join(_HERE, "lib", "libmoe_hip.so")  # (MOE_LIB_PATH: A/B builds, tools only)

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)

MOE_OK, MOE_ERR_RUNTIME, MOE_ERR_BOUNDS, MOE_ERR_INVALID_VALUE, MOE_ERR_SINGULAR = 0, 1, 2, 3, 4
COV_SQUARE_EXPONENTIAL, COV_MATERN_NU_2P5 = 0, 1


class MoeError(C.Structure):
    _fields_ = [("code", C.c_int), ("message", C.c_char * 480), ("payload", C.c_double * 3)]


class GdParams(C.Structure):
    _fields_ = [("num_multistarts", C.c_int), ("max_num_steps", C.c_int), ("max_num_restarts", C.c_int),
                ("num_steps_averaged", C.c_int), ("gamma", C.c_double), ("pre_mult", C.c_double),
                ("max_relative_change", C.c_double), ("tolerance", C.c_double), ("domain_type", C.c_int)]


class KgStats(C.Structure):
    _fields_ = [("posterior_mean_evals", C.c_longlong), ("posterior_grad_evals", C.c_longlong), ("ms_state", C.c_double),
                ("ms_mc", C.c_double), ("ms_tail", C.c_double)]


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, dp, dp, C.c_int)  # moe_allgather_fn


class Comm(C.Structure):  # moe_comm_t
    _fields_ = [("rank", C.c_int), ("world", C.c_int), ("allgather", ALLGATHER_FN), ("ctx", C.c_void_p)]


# every symbol include/moe_hip.h declares: (restype, argtypes)
_EP = C.POINTER(MoeError)
_GP = C.c_void_p
_GPA = C.POINTER(C.c_void_p)  # const moe_gp_t* const*: the handles of an MCMC ensemble
SIGNATURES = {
    "moe_version": (C.c_char_p, []),
    "moe_device_count": (C.c_int, [ip]),
    "moe_device_arch": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "moe_gp_create": (C.c_int, [dp, C.c_int, dp, dp, dp, ip, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_GP), _EP]),
    "moe_gp_destroy": (C.c_int, [_GP]),
    "moe_gp_dim": (C.c_int, [_GP]),
    "moe_gp_num_sampled": (C.c_int, [_GP]),
    "moe_gp_num_derivatives": (C.c_int, [_GP]),
    "moe_gp_add_points": (C.c_int, [_GP, dp, dp, C.c_int, _EP]),
    "moe_gp_get_factor": (C.c_int, [_GP, dp, dp, dp, _EP]),
    "moe_gp_mean": (C.c_int, [_GP, dp, C.c_int, dp, _EP]),
    "moe_gp_additional_mean": (C.c_int, [_GP, dp, C.c_int, dp, _EP]),
    "moe_gp_grad_mean": (C.c_int, [_GP, dp, C.c_int, dp, _EP]),
    "moe_gp_variance": (C.c_int, [_GP, dp, C.c_int, dp, _EP]),
    "moe_gp_cholesky_variance": (C.c_int, [_GP, dp, C.c_int, dp, _EP]),
    "moe_gp_grad_variance": (C.c_int, [_GP, dp, C.c_int, C.c_int, dp, _EP]),
    "moe_gp_grad_cholesky_variance": (C.c_int, [_GP, dp, C.c_int, C.c_int, dp, _EP]),
    "moe_posterior_mean": (C.c_int, [_GP, C.c_int, dp, dp, dp, _EP]),
    "moe_normal_draws": (C.c_int, [C.c_uint, C.c_longlong, dp]),
    "moe_ei": (C.c_int, [_GP, dp, dp, C.c_int, C.c_int, C.c_int, C.c_double, dp, dp, dp, _EP]),
    "moe_ei_batch": (C.c_int, [_GP, dp, C.c_int, dp, C.c_int, C.c_int, C.c_int, C.c_double, dp, dp, dp, _EP]),
    "moe_ei_analytic_batch": (C.c_int, [_GP, dp, C.c_int, C.c_double, dp, dp, _EP]),
    "moe_ei_multistart": (C.c_int, [_GP, C.POINTER(GdParams), dp, dp, C.c_int, dp, C.c_int, C.c_int, C.c_int, C.c_double, dp,
                                    C.c_int, dp, dp, ip, _EP]),
    "moe_kg": (C.c_int, [_GP, C.c_int, C.POINTER(GdParams), dp, dp, C.c_int, dp, dp, C.c_int, C.c_int, C.c_int, C.c_double,
                         dp, C.c_int, C.c_int, C.c_int, dp, dp, dp, C.POINTER(KgStats), _EP]),
    "moe_kg_batch": (C.c_int, [_GP, C.c_int, C.POINTER(GdParams), dp, dp, C.c_int, dp, C.c_int, dp, C.c_int, C.c_int,
                               C.c_int, C.c_double, dp, C.c_int, C.c_int, C.c_int, dp, dp, C.POINTER(KgStats), _EP]),
    "moe_kg_batch_multi": (C.c_int, [_GPA, C.c_int, C.c_int, C.c_int, C.POINTER(GdParams), dp, dp, C.c_int, dp, C.c_int, dp,
                                     C.c_int, C.c_int, C.c_int, C.c_double, dp, C.c_int, dp, dp, C.POINTER(KgStats), _EP]),
    "moe_kg_multistart": (C.c_int, [_GP, C.c_int, C.POINTER(GdParams), C.POINTER(GdParams), dp, dp, C.c_int, dp, C.c_int, dp,
                                    C.c_int, C.c_int, C.c_int, C.c_double, dp, C.c_int, dp, dp, ip, _EP]),
    "moe_kg_mcmc_batch": (C.c_int, [_GPA, C.c_int, C.c_int, C.POINTER(GdParams), dp, dp, C.c_int, dp, C.c_int, dp, C.c_int,
                                    C.c_int, C.c_int, dp, dp, C.c_int, C.c_int, dp, dp, _EP]),
    "moe_kg_mcmc_finalize": (C.c_int, [dp, dp, dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "moe_ei_mcmc_batch": (C.c_int, [_GPA, C.c_int, dp, C.c_int, dp, C.c_int, C.c_int, C.c_int, dp, dp, C.c_int, dp, dp, _EP]),
    "moe_kg_mcmc_multistart": (C.c_int, [_GPA, C.c_int, C.c_int, C.POINTER(GdParams), C.POINTER(GdParams), dp, dp, C.c_int, dp,
                                         C.c_int, dp, C.c_int, C.c_int, C.c_int, dp, dp, C.c_int, dp, dp, ip, _EP]),
    "moe_multistart_trace": (C.c_int, [dp, C.c_int]),
    "moe_pool_held_bytes": (C.c_longlong, []),
    "moe_pool_trim": (C.c_int, []),
    "moe_debug_sharded_items": (C.c_int, [C.POINTER(Comm), C.c_int, C.c_int, C.c_double, C.c_int, dp, _EP]),
    "moe_rccl_unique_id": (C.c_int, [C.c_char_p, _EP]),
    "moe_rccl_create": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), _EP]),
    "moe_rccl_comm": (C.c_int, [C.c_void_p, C.POINTER(Comm)]),
    "moe_rccl_allreduce_sum": (C.c_int, [C.c_void_p, dp, C.c_int, _EP]),
    "moe_rccl_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_double)]),
    "moe_rccl_destroy": (None, [C.c_void_p]),
    "moe_kg_multistart_comm": (C.c_int, [_GP, C.POINTER(Comm), C.c_int, C.POINTER(GdParams), C.POINTER(GdParams), dp, dp, C.c_int,
                                         dp, C.c_int, dp, C.c_int, C.c_int, C.c_int, C.c_double, dp, C.c_int, dp, dp, ip, _EP]),
    "moe_kg_mcmc_multistart_comm": (C.c_int, [_GPA, C.c_int, C.c_int, C.POINTER(Comm), C.c_int, C.POINTER(GdParams),
                                              C.POINTER(GdParams), dp, dp, C.c_int, dp, C.c_int, dp, C.c_int, C.c_int, C.c_int,
                                              dp, dp, C.c_int, dp, dp, ip, _EP]),
    "moe_kg_multistart_multi": (C.c_int, [_GPA, C.c_int, C.c_int, C.POINTER(GdParams), C.POINTER(GdParams), dp, dp, C.c_int, dp,
                                          C.c_int, dp, C.c_int, C.c_int, C.c_int, C.c_double, dp, C.c_int, dp, dp, ip, _EP]),
    "moe_kg_mcmc_multistart_multi": (C.c_int, [_GPA, C.c_int, C.c_int, C.c_int, C.POINTER(GdParams), C.POINTER(GdParams), dp, dp,
                                               C.c_int, dp, C.c_int, dp, C.c_int, C.c_int, C.c_int, dp, dp, C.c_int, dp, dp, ip,
                                               _EP]),
    "moe_ei_mcmc_multistart": (C.c_int, [_GPA, C.c_int, C.POINTER(GdParams), dp, dp, C.c_int, dp, C.c_int, C.c_int, C.c_int, dp,
                                         dp, C.c_int, dp, dp, ip, _EP]),
    "moe_ll_create": (C.c_int, [C.c_int, dp, dp, ip, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), _EP]),
    "moe_ll_destroy": (C.c_int, [C.c_void_p]),
    "moe_ll_evaluate": (C.c_int, [C.c_void_p, dp, C.c_int, dp, _EP]),
    "moe_ll_grad": (C.c_int, [C.c_void_p, dp, dp, _EP]),
    "moe_ll_ascend": (C.c_int, [C.c_void_p, C.POINTER(GdParams), dp, dp, dp, _EP]),
    "moe_ll_multistart": (C.c_int, [C.c_void_p, C.POINTER(GdParams), dp, dp, C.c_int, dp, dp, ip, _EP]),
    "moe_posterior_mean_optimize": (C.c_int, [_GP, C.c_int, C.POINTER(GdParams), dp, dp, dp, dp, _EP]),
    "moe_latin_hypercube": (C.c_int, [C.c_uint, dp, C.c_int, C.c_int, dp]),
    "moe_gp_mix_covariance": (C.c_int, [_GP, dp, C.c_int, ip, C.c_int, dp, _EP]),
    "moe_cov_build_probe": (C.c_int, [_GP, dp, C.c_int, C.c_int, dp, dp, _EP]),
    "moe_debug_cholesky": (C.c_int, [C.c_int, dp, C.c_int, dp, dp, ip, _EP]),
    "moe_debug_math": (C.c_int, [C.c_int, dp, C.c_int, dp, dp, _EP]),
    "moe_kxx_build_probe": (C.c_int, [_GP, C.c_int, dp, dp, _EP]),
    "moe_debug_fp64_rate": (C.c_int, [C.c_int, dp, _EP]),
    "moe_set_reference_quirks": (C.c_int, [C.c_int]),
    "moe_get_reference_quirks": (C.c_int, []),
    "moe_set_ensemble_launches": (C.c_int, [C.c_int]),
    "moe_ensemble_launch_stats": (C.c_int, [C.POINTER(C.c_longlong)]),
    "moe_last_kernel_ms": (C.c_int, [_GP, dp]),
    "moe_last_kernel_info": (C.c_int, [_GP, C.POINTER(C.c_int)]),
}

_lib = None


def load():
    """Load libmoe_hip.so (building it is __graft_entry__.build()'s / `python -m cornell_moe_amd.build`'s job)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: run `python -m cornell_moe_amd.build` (needs hipcc); there is no CPU fallback"
                              % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def device_count():
    n = C.c_int(0)
    load().moe_device_count(C.byref(n))
    return n.value


def require_gpu():
    if device_count() <= 0:
        raise RuntimeError("libmoe_hip: no HIP device visible and there is no CPU fallback")
