"""d-KG (derivative observations) at small and mid sizes: wall per evaluation next to the device phases.   python tools/dkg_sweep.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cornell_moe_amd.api import DeviceGP  # noqa: E402
from cornell_moe_amd.workloads import make_workload  # noqa: E402

CASES = ((100, 6, 4, 2000, (0,), 16), (300, 12, 8, 4000, (0, 1, 2), 8), (500, 12, 8, 4000, (0, 1, 2), 8), (400, 8, 4, 4000, (0, 1), 16),
         (600, 8, 4, 4000, (0, 1, 2, 3), 16), (800, 12, 8, 4000, (0, 1, 2), 8), (1200, 12, 8, 4000, (0, 1, 2), 8))
for n, d, q, M, derivs, E in CASES:
    w = make_workload(seed=31 + n + d, n=n, d=d, q=q, M=M, P=10, derivs=derivs, num_restarts=E)
    G = DeviceGP(w.hyperparameters, w.X, w.y, w.noise, derivs)
    best = float(G.additional_mean(w.discrete).min())
    for _ in range(3):
        t0 = time.perf_counter()
        G.kg_batch(w.inner_gd, w.bounds, w.discrete, w.Xq_restarts, None, w.M, best, w.kg_normals)
        dt = (time.perf_counter() - t0) / E
    km = G.last_kernel_ms()
    info = G.last_kernel_info()
    print("n=%4d d=%2d g=%d q=%d M=%d, %2d per call: %.3f ms/eval wall (device: mc %.3f tail %.3f state %.3f) variant %d xlds %d waves %d"
          % (n, d, len(derivs), q, M, E, 1e3 * dt, km["mc"], km["tail"], km["state"], info["variant"], info["xlds"], info["waves"]), flush=True)
    G.close()
