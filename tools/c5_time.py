"""C5 (d-KG, n = 2000, d = 12, g = 3, q = 8, M = 20 000) per evaluation, by MC kernel variant.   python tools/c5_time.py [batch = 4]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cornell_moe_amd.api import DeviceGP  # noqa: E402
from cornell_moe_amd.workloads import make_workload  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4
w = make_workload("C5", num_restarts=E)
G = DeviceGP(w.hyperparameters, w.X, w.y, w.noise, w.derivs)
best = float(G.additional_mean(w.discrete).min())
ref = None
for variant in ("2", "1"):
    os.environ["MOE_KG_VARIANT"] = variant
    for i in range(3):
        t0 = time.perf_counter()
        r = G.kg_batch(w.inner_gd, w.bounds, w.discrete, w.Xq_restarts[:E], None, w.M, best, w.kg_normals)
        dt = time.perf_counter() - t0
    km = G.last_kernel_ms()
    info = G.last_kernel_info()
    print("variant %s: %.2f ms/eval wall (%.1f evals/s); device mc %.3f tail %.3f state %.3f cov %.3f; passes %.1f + %.1f; info %s"
          % (variant, 1e3 * dt / E, E / dt, km["mc"], km["tail"], km["state"], km["cov_build"], r["mean_evals"] / (E * w.M),
             r["grad_evals"] / (E * w.M), info), flush=True)
    if ref is None:
        ref = r
    else:
        import numpy as np
        kk = "kg_all" if "kg_all" in r else [k for k in r if k.startswith("kg")][0]
        gk = [k for k in r if k.startswith("grad") and hasattr(r[k], "shape")][0]
        print("   variant 1 vs 2 (%s, %s): max rel KG diff %.2e, max grad diff / max |grad| %.2e" % (kk, gk,
            float(np.max(np.abs(np.asarray(r[kk]) - np.asarray(ref[kk])) / np.abs(np.asarray(ref[kk])))),
            float(np.abs(r[gk] - ref[gk]).max() / np.abs(ref[gk]).max())))
