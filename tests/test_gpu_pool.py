"""GPU tests of the library's device-memory pool (common.hpp: DevicePool; include/moe_hip.h: moe_pool_held_bytes, moe_pool_trim): objects
built on recycled buffers, pinned staging buffers and streams give the results of objects built on fresh ones, bit for bit -- no kernel may
rely on what a fresh hipMalloc happens to contain."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_PROBE = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
from cornell_moe_amd import api
from cornell_moe_amd.workloads import make_workload
w = make_workload(seed=77, n=150, d=5, q=3, M=48, P=6, derivs=(1,))
out = []
for rep in range(3):
    G = api.DeviceGP(w.hyperparameters, w.X, w.y, w.noise, w.derivs)
    best = float(G.additional_mean(w.discrete).min())
    r = G.kg((1, 6, 1, 3, 0.0, 1.0, 0.1, 1e-10), w.bounds, w.discrete, w.Xq, None, w.M, best, w.kg_normals)
    var = G.cholesky_variance(w.Xq)
    out.append((float(r["kg_sum"]).hex(), np.asarray(r["grad_sum"]).tobytes().hex(), np.asarray(var).tobytes().hex()))
    G.close()
    if rep == 1:
        api.pool_trim()
print(repr(out))
print(api.pool_held_bytes())
"""


def _run(env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    p = subprocess.run([sys.executable, "-c", _PROBE % ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = p.stdout.strip().splitlines()
    return eval(lines[-2]), int(lines[-1])


def test_pooled_buffers_give_the_results_of_fresh_ones():
    pooled, held = _run({"MOE_POOL": "1"})
    fresh, held_off = _run({"MOE_POOL": "0"})
    # MOE_POOL_POISON=1: a released block is filled with 0xFF bytes (NaN doubles, -1 ints) before it is pooled
    poisoned, _ = _run({"MOE_POOL": "1", "MOE_POOL_POISON": "1"})
    assert held > 0 and held_off == 0
    # build 0 on fresh memory, build 1 on the blocks build 0 released, build 2 after a trim: all the same, the same without the pool, and
    # the same when every recycled block comes back as NaNs
    assert pooled[0] == pooled[1] == pooled[2] == fresh[0] == fresh[1] == fresh[2] == poisoned[0] == poisoned[1] == poisoned[2]


def test_pool_accounting_and_trim():
    from cornell_moe_amd import _lib, api
    from cornell_moe_amd.workloads import make_workload
    _lib.require_gpu()
    if os.environ.get("MOE_POOL", "1") == "0":
        pytest.skip("the pool is switched off in this environment (MOE_POOL=0)")
    api.pool_trim()
    assert api.pool_held_bytes() == 0
    w = make_workload(seed=78, n=300, d=4, q=2, M=16, P=4)
    G = api.DeviceGP(w.hyperparameters, w.X, w.y, w.noise, w.derivs)
    in_use = api.pool_held_bytes()
    G.close()
    released = api.pool_held_bytes()
    # the GP's factor and inverse factor alone: 2 x (n + head-room)^2 doubles
    assert released - in_use >= 2 * 8 * 300 * 300
    G2 = api.DeviceGP(w.hyperparameters, w.X, w.y, w.noise, w.derivs)  # takes the same blocks back
    assert api.pool_held_bytes() <= in_use + 4096
    G2.close()
    api.pool_trim()
    assert api.pool_held_bytes() == 0
