"""CPU tests (-m "not gpu"): the C-ABI library builds, loads and exports every symbol include/moe_hip.h declares;
compute entry points fail LOUDLY (no CPU fallback) when no GPU is visible."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from cornell_moe_amd import _lib, api, build as moe_build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    moe_build.build()  # hipcc cross-compiles gfx950 without a GPU
    return _lib.load()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "moe_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(moe_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported(lib):
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), "libmoe_hip.so does not export %s" % name
    assert set(declared) == set(_lib.SIGNATURES.keys()), "ctypes signatures out of sync with include/moe_hip.h"


def test_version_and_struct_layout(lib):
    assert b"gfx950" in lib.moe_version()
    assert C.sizeof(_lib.GdParams) == 4 * 4 + 4 * 8 + 8   # (+ domain_type, r4: an int padded to the struct's 8-byte alignment)
    assert C.sizeof(_lib.MoeError) == 4 + 480 + 4 + 3 * 8  # int, char[480], pad to 8, double[3]


def test_normal_draws_deterministic(lib):
    a = api.normal_draws(3141, 1001)
    b = api.normal_draws(3141, 1001)
    c = api.normal_draws(3142, 1001)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    big = api.normal_draws(7, 200000)
    assert abs(big.mean()) < 0.01 and abs(big.std() - 1.0) < 0.01


def test_no_silent_cpu_fallback(lib):
    """Without a GPU every compute entry point must raise; with one this test is a no-op (covered by -m gpu tests)."""
    if _lib.device_count() > 0:
        pytest.skip("GPU present")
    X = np.random.default_rng(0).uniform(size=(10, 2))
    with pytest.raises(api.OptimalLearningException) as ei:
        api.DeviceGP([1.0, 0.5, 0.5], X, np.zeros((10, 1)), [0.1])
    assert "no CPU fallback" in str(ei.value)
    with pytest.raises(api.OptimalLearningException):
        api.debug_cholesky(np.eye(3))
    # GPP.run_cpp_tests() is a device self-test (r5): without a device it must not report "0 failures"
    from cornell_moe_amd import GPP
    with pytest.raises(RuntimeError) as er:
        GPP.run_cpp_tests()
    assert "no CPU fallback" in str(er.value)


def test_null_arguments_are_errors_not_crashes(lib):
    """NULL handles / NULL mandatory arguments come back as MOE_ERR_* codes through every handle-taking entry point
    that can be reached without a device (the checks run before any device work)."""
    err = _lib.MoeError()
    assert lib.moe_gp_dim(None) == -1 and lib.moe_gp_num_sampled(None) == -1 and lib.moe_gp_num_derivatives(None) == -1
    assert lib.moe_last_kernel_ms(None, None) == _lib.MOE_ERR_RUNTIME
    assert lib.moe_device_arch(0, None, 0) == _lib.MOE_ERR_BOUNDS
    out = np.zeros(4)
    dp = _lib.dp
    assert lib.moe_gp_mean(None, out.ctypes.data_as(dp), 1, out.ctypes.data_as(dp), C.byref(err)) == _lib.MOE_ERR_RUNTIME
    assert b"NULL GP handle" in err.message
    assert lib.moe_kg_batch(None, 0, None, None, None, 0, None, 1, None, 1, 0, 2, 0.0, None, 0, 2, 1, None, None, None,
                            C.byref(err)) == _lib.MOE_ERR_RUNTIME
    assert lib.moe_kg(None, 0, None, None, None, 0, None, None, 1, 0, 2, 0.0, None, 0, 2, 1, None, None, None, None,
                      C.byref(err)) == _lib.MOE_ERR_RUNTIME
    assert lib.moe_kg_batch_multi(None, 0, 0, 0, None, None, None, 0, None, 1, None, 1, 0, 2, 0.0, None, 1, None, None, None,
                                  C.byref(err)) == _lib.MOE_ERR_RUNTIME
    assert lib.moe_gp_add_points(None, None, None, 1, C.byref(err)) == _lib.MOE_ERR_RUNTIME
