"""Diagnostics for cornell_moe_amd/selftest.py: the quantities behind the checks that compare with a tolerance."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cornell_moe_amd import api, selftest as st

rng, X, y, hyper = st._problem(seed=12)
G = api.DeviceGP(hyper, X, y, [0.01])
m, d = 3, 3
pts = rng.uniform(size=(m, d))
for chol in (False, True):
    raw = (G.grad_cholesky_variance(pts, m) if chol else G.grad_variance(pts, m)).reshape(m, m, m, d)
    fd = st._fd(lambda p: st._variance_matrix(G, p, chol), pts)
    for order in ("[p][col][row]", "[p][row][col]"):
        worst = 0.0
        for p in range(m):
            for i in range(m):
                for j in range(i + 1):
                    a = raw[p, j, i] if order == "[p][col][row]" else raw[p, i, j]
                    worst = max(worst, np.abs(a - fd[i, j, p]).max())
        print("chol", chol, order, "worst abs diff", worst, "scale", np.abs(fd).max())

rng, X, y, hyper = st._problem(seed=13)
G = api.DeviceGP(hyper, X, y, [0.01])
x = rng.uniform(size=(1, 3))
for best in (float(G.mean(x)[0]),):
    ea, ga = G.ei_analytic_batch(x, best)
    M = 200000
    em, gm = G.ei(x, None, M, best, api.normal_draws(5, M))
    sd = np.sqrt(max(G.variance(x)[0], 1e-30))
    print("EI best", best, "analytic", ea[0], "mc", em, "diff", em - ea[0], "5 sd/sqrt(M)", 5 * sd / np.sqrt(M), "grad a", ga[0], "grad mc", gm[0])

rng, X, y, hyper = st._problem(seed=17)
G = api.DeviceGP(hyper, X, y, [0.01])
bounds = np.tile([0.0, 1.0], 3)
x0 = rng.uniform(size=3)
x1, v1 = G.posterior_mean_optimize((1, 40, 2, 3, 0.0, 1.0, 0.2, 1e-9), bounds, x0)
print("posterior mean: start", G.posterior_mean(x0, want_grad=False)[0], "end", v1, x1)
disc = rng.uniform(size=(60, 3))
best = float(G.additional_mean(disc).min())
starts = rng.uniform(0.2, 0.8, size=(6, 2, 3))
M = 32
Z = rng.standard_normal((M // 2, 2))
inner = (1, 4, 1, 3, 0.0, 1.0, 0.1, 1e-10)
api.set_reference_quirks(0)
pt, val, found = G.kg_multistart((6, 6, 1, 3, 0.7, 0.02, 0.05, 1e-12), inner, bounds, disc, starts, None, M, best, Z)
api.set_reference_quirks(1)
pt = np.asarray(pt)
at_starts = G.kg_batch(inner, bounds, disc, starts, None, M, best, Z, want_grad=False)["kg_sum"] / M
at_end = G.kg_batch(inner, bounds, disc, pt.reshape(1, 2, 3), None, M, best, Z, want_grad=False)["kg_sum"][0] / M
print("kg multistart: found", found, "val", val, "at_end", at_end, "at_starts", at_starts, "pt", pt)
