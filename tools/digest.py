"""Bit-level digest of a spread of KG / EI results: run under two builds of the library (MOE_LIB_PATH) and compare the output lines.
    python tools/digest.py            (one line per case: sha1 of the raw result bytes)"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cornell_moe_amd import api  # noqa: E402
from cornell_moe_amd.workloads import make_workload  # noqa: E402

CASES = [
    ("q-KG C3-like", dict(seed=3, n=300, d=8, q=4, M=640, P=10)),
    ("q-KG tiny", dict(seed=4, n=30, d=2, q=4, M=128, P=11)),
    ("q-KG m=8", dict(seed=5, n=200, d=4, q=6, M=200, P=8, p=2)),
    ("q-KG m=12 (unfused tail)", dict(seed=6, n=150, d=5, q=12, M=96, P=6)),
    ("d-KG g=2", dict(seed=7, n=120, d=4, q=2, M=128, P=6, derivs=(0, 2))),
    ("d-KG g=3 q=8", dict(seed=8, n=200, d=12, q=8, M=64, P=10, derivs=(0, 1, 2))),
    ("q-KG many samples", dict(seed=9, n=100, d=3, q=2, M=40000, P=5)),
]


def h(*arrays):
    m = hashlib.sha1()
    for a in arrays:
        m.update(np.ascontiguousarray(np.asarray(a, dtype=np.float64)).tobytes())
    return m.hexdigest()[:16]


for name, kw in CASES:
    w = make_workload(num_restarts=5, **kw)
    G = api.DeviceGP(w.hyperparameters, w.X, w.y, w.noise, w.derivs)
    best = float(G.additional_mean(w.discrete).min())
    Xp = w.Xp if w.p else None
    one = G.kg(w.inner_gd, w.bounds, w.discrete, w.Xq, Xp, w.M, best, w.kg_normals)
    val = G.kg(w.inner_gd, w.bounds, w.discrete, w.Xq, Xp, w.M, best, w.kg_normals, want_grad=False)
    bat = G.kg_batch(w.inner_gd, w.bounds, w.discrete, w.Xq_restarts, Xp, w.M, best, w.kg_normals)
    half = (w.M // 4) * 2
    sh = G.kg(w.inner_gd, w.bounds, w.discrete, w.Xq, Xp, w.M, best, w.kg_normals, first_sample=half, num_local=w.M - half)
    print("%-28s one %s value-only %s batch %s shard %s  passes %d/%d" % (
        name, h(one["kg_sum"], one["grad_sum"]), h(val["kg_sum"]), h(bat["kg_sum"], bat["grad_sum"]), h(sh["kg_sum"], sh["grad_sum"]),
        one["mean_evals"], one["grad_evals"]), flush=True)
w = make_workload(seed=10, n=60, d=3, q=3, M=64, P=6)
hy = np.stack([w.hyperparameters * (1.0 + 0.05 * i) for i in range(4)])
mc = api.DeviceGPMCMC(hy, np.tile(w.noise, (4, 1)), w.X, w.y, ())
disc_all = np.stack([w.discrete] * 4)
bests = np.array([float(g.additional_mean(w.discrete).min()) for g in mc.gps])
kg, grad = mc.kg_batch(w.inner_gd, w.bounds, disc_all, w.Xq_restarts[:3], None, w.M, bests, w.kg_normals)
print("%-28s %s" % ("KG-MCMC batch", h(kg, grad)))
