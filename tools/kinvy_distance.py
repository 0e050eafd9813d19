"""Distance of the device's K^-1 (y - mean) from the reference's on the golden cases (why the tolerance is 1e-10)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cornell_moe_amd import api
from helpers import load_golden, rel
cases, _ = load_golden()
out = []
for c in cases:
    i = c.inp
    gp = api.DeviceGP(np.concatenate([[float(i["alpha"])], i["lengths"]]), i["X"], i["y"], i["noise"], list(i["derivs"]), cov_type=int(i["cov_type"]))
    K, kiy, mean = gp.get_factor()
    Lr = c.out["K_chol"]
    cond = (np.abs(np.diag(Lr)).max() / np.abs(np.diag(Lr)).min()) ** 2
    out.append((rel(kiy, c.out["K_inv_y"]), rel(np.tril(K), Lr), cond, i["X"].shape[0] * (1 + len(i["derivs"]))))
for r in out:
    print("N=%4d  K^-1 y rel. distance %.2e   factor %.2e   (diag(L) ratio)^2 %.1e" % (r[3], r[0], r[1], r[2]))
print("max K^-1 y distance %.2e, min %.2e" % (max(r[0] for r in out), min(r[0] for r in out)))
