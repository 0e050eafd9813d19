import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from cornell_moe_amd.api import DeviceGP
from cornell_moe_amd.workloads import make_workload
over = dict(kv.split("=") for kv in sys.argv[1].split(",")) if len(sys.argv) > 1 and sys.argv[1] else {}
over = {k: int(v) for k, v in over.items()}
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t0 = time.perf_counter()
w = make_workload("C5", num_restarts=R, **over)
G = DeviceGP(w.hyperparameters, w.X, w.y, w.noise, w.derivs)
print("GP build n=%d d=%d g=%d N=%d: %.3f s" % (w.n, w.d, w.g, w.n * (1 + w.g), time.perf_counter() - t0), flush=True)
best = float(G.additional_mean(w.discrete).min())
for i in range(2):
    t0 = time.perf_counter()
    r = G.kg_batch(w.inner_gd, w.bounds, w.discrete, w.Xq_restarts, None, w.M, best, w.kg_normals)
    dt = time.perf_counter() - t0
    km = G.last_kernel_ms()
    print("rep %d: %.2f ms/eval wall; mc %.3f cov %.3f tail %.3f state %.3f; passes/sample value %.2f grad %.2f; kg %.6g |grad| %.4g"
          % (i, 1e3 * dt / R, km["mc"], km["cov_build"], km["tail"], km["state"], r["mean_evals"] / (R * w.M),
             r["grad_evals"] / (R * w.M), r["kg_sum"][0] / w.M, np.abs(r["grad_sum"][0]).max() / w.M), flush=True)
