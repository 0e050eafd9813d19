"""Reproduce one case of tools/fuzz_parity.py with diagnostics:  python tools/fuzz_case.py <seed> <case>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cornell_moe_amd import api  # noqa: E402
from cornell_moe_amd.workloads import make_workload  # noqa: E402
from oracle import orc, ref  # noqa: E402

seed, want = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for case in range(want + 1):
    d = int(rng.integers(1, 17))
    g = int(rng.integers(0, min(4, d) + 1)) if rng.uniform() < 0.5 else 0
    derivs = tuple(int(v) for v in rng.permutation(d)[:g])
    umax = 64 // (1 + g)
    q = int(rng.integers(1, min(4, umax) + 1))
    p = int(rng.integers(0, min(3, umax - q) + 1))
    n = int(rng.integers(1, 300))
    P = int(rng.integers(1, 13))
    M = int(rng.integers(1, 65))
    cov = int(rng.integers(0, 2))
    f = int(rng.integers(0, d)) if rng.uniform() < 0.3 else 0
    gd = (1, int(rng.integers(1, 8)), int(rng.integers(1, 3)), 3, float(rng.choice([0.0, 0.5, 1.0])),
          float(rng.choice([1.0, 0.3, 2.0])), float(rng.choice([0.1, 0.5, 1.0])), float(rng.choice([1e-10, 1e-6])))
    w = make_workload(seed=10_000 + case, n=n, d=d, q=q, M=M, P=P, derivs=derivs, p=p)
    shift = scale = None
    if rng.uniform() < 0.4:
        shift = rng.choice([-50.0, 10.0, 100.0, 1000.0], size=d) * (rng.uniform(size=d) < 0.7)
        scale = rng.choice([0.1, 1.0, 10.0], size=d)
        for name in ("X", "Xq", "Xp", "discrete"):
            setattr(w, name, shift + scale * getattr(w, name))
        w.lengths = w.lengths * scale
        w.hyperparameters = np.concatenate([[w.alpha], w.lengths])
        w.bounds = np.column_stack([shift, shift + scale]).reshape(-1)
print("case", want, dict(n=n, d=d, q=q, p=p, derivs=derivs, f=f, P=P, M=M, cov=cov, gd=gd), "shift", shift, "scale", scale)
disc = w.discrete[:, :d - f]
bounds = w.bounds[:2 * (d - f)]
O = orc.OrcGP(cov, w.alpha, w.lengths, w.X, w.y, w.noise, derivs)
G = api.DeviceGP(w.hyperparameters, w.X, w.y, w.noise, derivs, cov_type=cov)
pts = np.vstack([w.Xq, w.Xp]) if p else w.Xq
print("mean  dev", G.mean(pts)[:4], "\n      orc", O.mean(pts)[:4])
print("var   dev", G.variance(pts)[:3], "\n      orc", O.var(pts)[:3])
full = np.hstack([disc, np.ones((disc.shape[0], f))])
best = float(O.additional_mean(full).min())
Xp = w.Xp if p else None
ro = O.kg(gd, bounds, disc, w.Xq, Xp, M, best, w.kg_normals, num_fidelity=f)
if ref.available():  # (r4) the reference itself next to its restatement
    R = ref.RefGP(cov, w.alpha, w.lengths, w.X, w.y, w.noise, list(derivs))
    rr = R.kg(gd, bounds, disc, w.Xq, Xp, M, best, w.kg_normals, num_fidelity=f)
    print("restatement vs reference: kg", ro["kg"], rr["kg"], "max |dx*|", float(np.abs(ro["best_point"] - rr["best_point"]).max()),
          "max |dgrad|", float(np.abs(ro["grad"] - rr["grad"]).max()))
for variant in ("0", "1", "2"):
    os.environ["MOE_KG_VARIANT"] = variant
    rg = G.kg(gd, bounds, disc, w.Xq, Xp, M, best, w.kg_normals, num_fidelity=f, want_best_points=True)
    print("variant", variant, "kg dev", rg["kg"], "orc", ro["kg"], "grad evals", rg["grad_evals"], ro["grad_evals"])
    bad = ~np.isfinite(rg["best_point"]).all(axis=1)
    print("  non-finite best points:", int(bad.sum()), " max |dx*|", float(np.nanmax(np.abs(rg["best_point"] - ro["best_point"]))))
    print("  grad dev", np.asarray(rg["grad"]).ravel()[:4], "orc", np.asarray(ro["grad"]).ravel()[:4])
    off = np.nonzero(np.abs(rg["best_point"] - ro["best_point"]).max(axis=1) > 1e-8)[0]
    for i in off[:4]:
        print("  sample", int(i), "dev x*", rg["best_point"][i], "\n             orc x*", ro["best_point"][i])
if len(sys.argv) > 3:  # extra environment for a third run of the wave-per-sample kernel, e.g. MOE_KG_DOT_MAX_RADIUS2=0
    for kv in sys.argv[3:]:
        k, v = kv.split("=")
        os.environ[k] = v
    os.environ["MOE_KG_VARIANT"] = "0"
    rg = G.kg(gd, bounds, disc, w.Xq, Xp, M, best, w.kg_normals, num_fidelity=f, want_best_points=True)
    off = np.nonzero(np.abs(rg["best_point"] - ro["best_point"]).max(axis=1) > 1e-6)[0]
    print("with", sys.argv[3:], ": kg dev", rg["kg"], "orc", ro["kg"], "samples off:", off)
