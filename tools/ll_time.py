"""Log-likelihood evaluation rate (moe_ll_evaluate / moe_ll_grad) at the benchmark sizes: what a hyper-parameter MCMC loop
(emcee over compute_log_likelihood) pays per proposal.   python tools/ll_time.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cornell_moe_amd import api  # noqa: E402

for n, d, g, batch in ((1000, 8, 0, 16), (2000, 12, 3, 4)):
    rng = np.random.default_rng(3)
    X = rng.uniform(size=(n, d))
    y = rng.normal(size=(n, 1 + g))
    LL = api.LogLikelihood(X, y, tuple(range(g)))
    th = np.r_[1.0, np.full(d, 0.7), np.full(1 + g, 0.05)]
    sets = np.array([th * (1.0 + 0.01 * i) for i in range(batch)])
    for label, fn, per in (("evaluate x1", lambda: LL.evaluate(sets[:1]), 1), ("evaluate x%d" % batch, lambda: LL.evaluate(sets), batch),
                           ("grad", lambda: LL.grad(th), 1)):
        v0 = fn()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            fn()
        dt = (time.perf_counter() - t0) / reps
        N = n * (1 + g)
        print("n=%d d=%d g=%d N=%d  %-12s %8.2f ms per call, %8.2f ms per hyper-parameter set  (factorisation alone: %.1f TFLOP/s)"
              % (n, d, g, N, label, 1e3 * dt, 1e3 * dt / per, per * N ** 3 / 3.0 / dt / 1e12), flush=True)
        print("      first values: %s" % np.array2string(np.ravel(v0)[:3], precision=12), flush=True)
