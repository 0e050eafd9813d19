"""Time moe_gp_add_points: the rank-k block-row append vs the full rebuild (MOE_GP_APPEND=0), at a few sizes.
   python tools/add_points_time.py            (run once per setting of MOE_GP_APPEND)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cornell_moe_amd import api  # noqa: E402

rng = np.random.default_rng(0)
print("MOE_GP_APPEND =", os.environ.get("MOE_GP_APPEND", "(default: on)"))
for n, d, k, derivs in ((1000, 8, 4, ()), (4000, 8, 4, ()), (8000, 8, 4, ()), (2000, 12, 4, (0, 1, 2))):
    X = rng.uniform(size=(n + 10 * k, d))
    y = np.sin(3 * X).sum(1, keepdims=True)
    if derivs:
        y = np.hstack([y] + [3 * np.cos(3 * X[:, [j]]) for j in derivs])
    gp = api.DeviceGP([1.0] + [0.7] * d, X[:n], y[:n], [0.01] * (1 + len(derivs)), derivatives=derivs)
    at = n
    ts = []
    for r in range(10):
        t0 = time.perf_counter()
        gp.add_points(X[at:at + k], y[at:at + k])
        ts.append(time.perf_counter() - t0)
        at += k
    print("n=%5d g=%d N=%6d  add %d points: median %.3f ms (first %.3f)" % (n, len(derivs), n * (1 + len(derivs)), k,
                                                                          1e3 * np.median(ts[1:]), 1e3 * ts[0]))
