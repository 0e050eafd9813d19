"""rocprofv3 driver: d-KG at a mid size (n = 500, d = 12, q = 8, g = 3, M = 4000), 8 evaluations per call, 4 calls."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cornell_moe_amd.api import DeviceGP
from cornell_moe_amd.workloads import make_workload
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
w = make_workload(seed=31 + n + 12, n=n, d=12, q=8, M=4000, P=10, derivs=(0, 1, 2), num_restarts=8)
G = DeviceGP(w.hyperparameters, w.X, w.y, w.noise, (0, 1, 2))
best = float(G.additional_mean(w.discrete).min())
for _ in range(4):
    G.kg_batch(w.inner_gd, w.bounds, w.discrete, w.Xq_restarts, None, w.M, best, w.kg_normals)
print(G.last_kernel_ms())
