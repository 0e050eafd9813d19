import sys
import numpy as np
sys.path.insert(0, '/root/repo')
from cornell_moe_amd import api
from cornell_moe_amd.workloads import make_workload
from oracle import orc
w = make_workload(seed=33, n=80, d=3, q=2, M=400, P=7, derivs=(), p=1)
O = orc.OrcGP(0, w.alpha, w.lengths, w.X, w.y, w.noise, ())
G = api.DeviceGP(w.hyperparameters, w.X, w.y, w.noise, (), cov_type=0)
best = float(np.median(w.y[:, 0]))
pts = np.random.default_rng(1).uniform(size=(30, 3))
want = np.array([O.ei_analytic(p, best, want_grad=False)[0] for p in pts])
for E in (1, 2, 5, 30):
    a = G.ei_analytic_batch(pts[:E], best, want_grad=False)[0]
    b, g = G.ei_analytic_batch(pts[:E], best, want_grad=True)
    print(E, np.abs(a - want[:E]).max(), np.abs(b - want[:E]).max())
mu = G.mean(pts[:3]); var = [G.variance(pts[k:k+1]) for k in range(3)]
print(mu, O.mean(pts[:3]), var, [O.var(pts[k:k+1]) for k in range(3)])
