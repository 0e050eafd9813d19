"""rocprofv3 driver: C3 q-KG value + gradient, ONE evaluation per call, 6 calls."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cornell_moe_amd.api import DeviceGP
from cornell_moe_amd.workloads import make_workload
w = make_workload("C3", num_restarts=1)
G = DeviceGP(w.hyperparameters, w.X, w.y, w.noise, w.derivs)
best = float(G.additional_mean(w.discrete).min())
for _ in range(6):
    G.kg_batch(w.inner_gd, w.bounds, w.discrete, w.Xq_restarts, None, w.M, best, w.kg_normals)
print(G.last_kernel_ms())
