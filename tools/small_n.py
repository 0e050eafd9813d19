"""q-KG gradient throughput at the small training-set sizes typical of a BO run (n = 50 ... 400), batches of 64 evaluations."""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from cornell_moe_amd.api import DeviceGP  # noqa: E402
from cornell_moe_amd.workloads import make_workload  # noqa: E402

for n, d, q, M in ((50, 4, 2, 1000), (100, 4, 2, 1000), (200, 6, 4, 2000), (400, 8, 4, 4000)):
    w = make_workload(seed=7, n=n, d=d, q=q, M=M, P=10, derivs=(), num_restarts=64)
    G = DeviceGP(w.hyperparameters, w.X, w.y, w.noise, ())
    best = float(G.additional_mean(w.discrete).min())
    for _ in range(3):
        t0 = time.perf_counter()
        r = G.kg_batch(w.inner_gd, w.bounds, w.discrete, w.Xq_restarts, None, w.M, best, w.kg_normals)
        dt = time.perf_counter() - t0
    km = G.last_kernel_ms()
    print("n=%4d d=%d q=%d M=%5d: %8.0f evals/s (%.3f ms/eval: mc %.3f tail %.3f state %.3f); passes/sample %.1f + %.1f" % (
        n, d, q, M, 64 / dt, 1e3 * dt / 64, km["mc"], km["tail"], km["state"], r["mean_evals"] / (64.0 * M),
        r["grad_evals"] / (64.0 * M)), flush=True)
