import sys
sys.path.insert(0, '/root/repo')
from cornell_moe_amd.api import DeviceGP
from cornell_moe_amd.workloads import make_workload
w = make_workload("C3", num_restarts=8)
G = DeviceGP(w.hyperparameters, w.X, w.y, w.noise, ())
best = float(G.additional_mean(w.discrete).min())
for gd in ((1, 6, 0, 3, 0.0, 1.0, 0.1, 1e-10), (1, 1, 1, 3, 0.0, 1.0 / 1024, 0.1, 1e-10), (1, 6, 1, 3, 0.0, 1.0, 0.1, 1e-10)):
    for _ in range(2):
        r = G.kg_batch(gd, w.bounds, w.discrete, w.Xq_restarts, None, w.M, best, w.kg_normals, want_grad=False)
    km = G.last_kernel_ms()
    print(gd[1:3], gd[5], "mc %.4f ms/eval" % km["mc"], "passes", r["mean_evals"] / 8e4, r["grad_evals"] / 8e4)
