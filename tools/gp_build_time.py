"""GP construction (covariance build + blocked Cholesky + explicit inverse factor + K^-1 y) wall time at the BASELINE shapes."""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from cornell_moe_amd.api import DeviceGP  # noqa: E402
from cornell_moe_amd.workloads import make_workload  # noqa: E402

for name in sys.argv[1:] or ["C2", "C3", "C5"]:
    w = make_workload(name)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        G = DeviceGP(w.hyperparameters, w.X, w.y, w.noise, w.derivs)
        ts.append(time.perf_counter() - t0)
        del G
    print("%s: N = %d  build %.2f ms (first %.1f ms)" % (name, w.n * (1 + len(w.derivs)), 1e3 * min(ts[1:]), 1e3 * ts[0]))

from cornell_moe_amd.api import LogLikelihood  # noqa: E402
for name in ["C2", "C3"]:
    w = make_workload(name)
    LL = LogLikelihood(w.X, w.y, w.derivs)
    sets = np.tile(np.r_[w.hyperparameters, w.noise], (20, 1)) * np.linspace(0.8, 1.2, 20)[:, None]
    LL.evaluate(sets[:2])
    t0 = time.perf_counter()
    LL.evaluate(sets)
    print("%s: log marginal likelihood %.2f ms per hyper-parameter set" % (name, 1e3 * (time.perf_counter() - t0) / 20))
    big = np.tile(np.r_[w.hyperparameters, w.noise], (256, 1)) * np.linspace(0.7, 1.3, 256)[:, None]
    LL.evaluate(big[:64])
    for k in (1, 8, 64, 256):
        t0 = time.perf_counter()
        LL.evaluate(big[:k])
        print("%s:   %3d sets per call: %.3f ms per set" % (name, k, 1e3 * (time.perf_counter() - t0) / k))
    LL.grad(sets[0])
    t0 = time.perf_counter()
    for k in range(10):
        LL.grad(sets[k])
    print("%s: hyper-parameter gradient of it %.2f ms per set (factorisation included)" % (name, 1e2 * (time.perf_counter() - t0)))
