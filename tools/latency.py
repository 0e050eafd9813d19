"""Latency of the small, launch-bound entry points (posterior queries, q-EI at C2, single q-KG evaluation at C3)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cornell_moe_amd.api import DeviceGP
from cornell_moe_amd.workloads import make_workload


def med(fn, reps=50):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))


w = make_workload("C2")
G = DeviceGP(w.hyperparameters, w.X, w.y, w.noise, ())
best = float(np.min(w.y[:, 0])) + 0.5
print("C2 mean(1 pt)            %.3f ms" % med(lambda: G.mean(w.query[:1])))
print("C2 variance(2 pts)       %.3f ms" % med(lambda: G.variance(w.query[:2])))
print("C2 posterior_mean+grad   %.3f ms" % med(lambda: G.posterior_mean(w.query[0])))
print("C2 EI value              %.3f ms" % med(lambda: G.ei(w.Xq, None, w.M, best, w.ei_normals, want_grad=False)))
print("C2 EI value+grad         %.3f ms" % med(lambda: G.ei(w.Xq, None, w.M, best, w.ei_normals)))
w3 = make_workload("C3")
G3 = DeviceGP(w3.hyperparameters, w3.X, w3.y, w3.noise, ())
b3 = float(G3.additional_mean(w3.discrete).min())
print("C3 KG value (1 eval)     %.3f ms" % med(lambda: G3.kg(w3.inner_gd, w3.bounds, w3.discrete, w3.Xq, None, w3.M, b3, w3.kg_normals, want_grad=False), 10))
print("C3 KG value+grad (1)     %.3f ms" % med(lambda: G3.kg(w3.inner_gd, w3.bounds, w3.discrete, w3.Xq, None, w3.M, b3, w3.kg_normals), 10))
print("   last kernel ms:", {k: round(float(v), 4) for k, v in G3.last_kernel_ms().items()})
