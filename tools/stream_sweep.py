"""Shapes whose coordinate table does not fit LDS next to the weight slabs (the streamed wave-per-sample kernel): q-KG value + gradient
per evaluation, 8 evaluations per call.   python tools/stream_sweep.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cornell_moe_amd.api import DeviceGP  # noqa: E402
from cornell_moe_amd.workloads import make_workload  # noqa: E402

CASES = ((1000, 13, 4, 2000, ()), (1000, 16, 4, 2000, ()), (1500, 8, 4, 10000, ()), (2000, 8, 4, 10000, ()), (3000, 8, 4, 10000, ()),
         (800, 12, 8, 4000, (0, 1, 2)), (1200, 12, 8, 4000, (0, 1, 2)))
for n, d, q, M, derivs in CASES:
    w = make_workload(seed=31 + n + d, n=n, d=d, q=q, M=M, P=10, derivs=derivs, num_restarts=8)
    G = DeviceGP(w.hyperparameters, w.X, w.y, w.noise, derivs)
    best = float(G.additional_mean(w.discrete).min())
    G.kg_batch(w.inner_gd, w.bounds, w.discrete, w.Xq_restarts, None, w.M, best, w.kg_normals)
    t0 = time.perf_counter()
    r = G.kg_batch(w.inner_gd, w.bounds, w.discrete, w.Xq_restarts, None, w.M, best, w.kg_normals)
    dt = (time.perf_counter() - t0) / 8
    km = G.last_kernel_ms()
    info = G.last_kernel_info()
    print("n=%4d d=%2d g=%d q=%d M=%5d: %.3f ms/eval (mc %.3f tail %.3f state %.3f) passes %.1f + %.1f  variant %d xlds %d waves %d tiles-in-LDS/tr %d"
          % (n, d, len(derivs), q, M, 1e3 * dt, km["mc"], km["tail"], km["state"], r["mean_evals"] / (8.0 * M), r["grad_evals"] / (8.0 * M),
             info["variant"], info["xlds"], info["waves"], info["tr"]), flush=True)
    G.close()
