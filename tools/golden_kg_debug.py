"""Per golden KG case: error against the reference fixture under the kernel variants (which kernel took it, errors vs tolerances)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_golden, kg_tolerances, TOL
from cornell_moe_amd import api
cases, _ = load_golden()
for c in cases:
    i = c.inp
    hyper = np.concatenate([[float(i["alpha"])], i["lengths"]])
    gp = api.DeviceGP(hyper, i["X"], i["y"], i["noise"], list(i["derivs"]), cov_type=int(i["cov_type"]))
    Xp = i["Xp"] if int(i["p"]) > 0 else None
    r = gp.kg(i["inner_gd"], i["bounds"], i["discrete"], i["Xq"], Xp, int(i["M"]), float(i["best_so_far"]), i["kg_normals"], want_best_points=True)
    info = gp.last_kernel_info()
    gtol, ptol = kg_tolerances(c)
    ekg = abs(r["kg"] - float(c.out["kg"])) / abs(float(c.out["kg"]))
    eg = np.abs(r["grad"] - c.out["grad_kg"]).max()
    ep = np.abs(r["best_point"] - c.out["kg_best_point"]).max()
    print("case %2d n=%3d d=%d g=%d gd=%s M=%d  variant %d lane %d waves %2d | kg rel %.1e (tol %.0e)  grad %.2e (tol %.1e)  points %.2e (tol %.0e) %s"
          % (c.index, i["X"].shape[0], i["X"].shape[1], len(i["derivs"]), tuple(int(v) for v in i["inner_gd"][:3]), int(i["M"]), info["variant"], info["lane"], info["waves"],
             ekg, TOL["kg"], eg, gtol, ep, ptol, "" if (ekg <= TOL["kg"] and eg <= gtol and ep <= ptol) else "<-- FAIL"), flush=True)
