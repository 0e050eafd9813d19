"""GP construction at the C5 sizes (covariance build + Cholesky + explicit inverse factor + K^-1 y): wall time per build, the
factorisation's share of the FP64 matrix peak, and the same with the one-level factorisation (MOE_CHOL_TWO_LEVEL_MIN huge).
    python tools/chol_time.py [g ...]      g = observed derivatives at n = 2000, d = 12 (3 -> N = 8000, 12 -> N = 26 000)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cornell_moe_amd.api import DeviceGP  # noqa: E402

PEAK = 78.6e12
for g in [int(v) for v in sys.argv[1:]] or [3, 12]:
    rng = np.random.default_rng(1005)
    n, d = 2000, 12
    X = rng.uniform(size=(n, d))
    derivs = tuple(range(g))
    y = np.zeros((n, 1 + g))
    y[:, 0] = np.sin(3 * X).sum(1) + 0.1 * rng.uniform(size=n)
    for a in range(g):
        y[:, 1 + a] = 3 * np.cos(3 * X[:, a])
    N = n * (1 + g)
    for mode, env in (("two-level + MFMA rank-512 update", None), ("one-level (round 1)", "1000000000")):
        if env is None:
            os.environ.pop("MOE_CHOL_TWO_LEVEL_MIN", None)
        else:
            os.environ["MOE_CHOL_TWO_LEVEL_MIN"] = env
        if env is not None and N > 10000:
            continue  # (the one-level path needs minutes there)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            G = DeviceGP(np.r_[1.0, np.full(d, 0.7)], X, y, np.full(1 + g, 0.01), derivs)
            ts.append(time.perf_counter() - t0)
            del G
        best = min(ts[1:])
        flops = 2.0 * N ** 3 / 3.0  # factorisation N^3/3 + inverse factor N^3/3 (multiply-adds counted as 2)
        print("n=%d g=%d N=%d  %-34s build %.2f ms (first %.0f ms): %.1f TFLOP/s = %.2f of the FP64 matrix peak for factor + inverse"
              % (n, g, N, mode, 1e3 * best, 1e3 * ts[0], flops / best / 1e12, flops / best / PEAK), flush=True)
