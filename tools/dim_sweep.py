"""q-KG value + gradient rate against the dimension (where the per-dimension instantiations of the MC kernels stand):
n = 300 and n = 1000, q = 4, M = 2000, 16 evaluations per call.   python tools/dim_sweep.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cornell_moe_amd.api import DeviceGP  # noqa: E402
from cornell_moe_amd.workloads import make_workload  # noqa: E402

for n in (300, 1000):
    for d in (4, 8, 12, 13, 16, 20, 24, 32):
        w = make_workload(seed=900 + d, n=n, d=d, q=4, M=2000, P=10, derivs=(), num_restarts=16)
        G = DeviceGP(w.hyperparameters, w.X, w.y, w.noise, ())
        best = float(G.additional_mean(w.discrete).min())
        for _ in range(3):  # (the first calls of a process also pay for lazy code-object loading and the first allocations)
            G.kg_batch(w.inner_gd, w.bounds, w.discrete, w.Xq_restarts, None, w.M, best, w.kg_normals)
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            r = G.kg_batch(w.inner_gd, w.bounds, w.discrete, w.Xq_restarts, None, w.M, best, w.kg_normals)
        dt = (time.perf_counter() - t0) / reps / 16
        km = G.last_kernel_ms()
        info = G.last_kernel_info()
        print("n=%4d d=%2d: %8.0f evals/s  %.3f ms/eval (mc %.3f tail %.3f state %.3f)  passes/sample %.1f + %.1f  kernel: variant %d xlds %d waves %d"
              % (n, d, 1.0 / dt, 1e3 * dt, km["mc"], km["tail"], km["state"], r["mean_evals"] / (16.0 * w.M), r["grad_evals"] / (16.0 * w.M),
                 info["variant"], info["xlds"], info["waves"]), flush=True)
        G.close()
