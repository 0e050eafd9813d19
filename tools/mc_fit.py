"""Fit the MC kernel's time per sample = fixed + (value passes) * a + (gradient passes) * b by varying the inner optimiser's
step count at the headline shape (C3, 8 evaluations per launch)."""
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from cornell_moe_amd.api import DeviceGP  # noqa: E402
from cornell_moe_amd.workloads import make_workload  # noqa: E402

w = make_workload("C3", num_restarts=8)
G = DeviceGP(w.hyperparameters, w.X, w.y, w.noise, ())
best = float(G.additional_mean(w.discrete).min())
rows = []
import itertools
for steps, pre in itertools.product((1, 3, 6, 12), (1.0, 0.125, 0.015625)):  # small pre_mult: fewer Armijo halvings per step
    gd = (1, steps, 1, 3, 0.0, pre, 0.1, 1e-10)
    for _ in range(2):
        r = G.kg_batch(gd, w.bounds, w.discrete, w.Xq_restarts, None, w.M, best, w.kg_normals)
    km = G.last_kernel_ms()
    S = r["mean_evals"] / (8.0 * w.M)
    Gp = r["grad_evals"] / (8.0 * w.M)
    rows.append((steps, km["mc"], S, Gp))
    print("steps %2d pre_mult %-8g: mc %.4f ms/eval, value passes %.2f, grad passes %.2f" % (steps, pre, km["mc"], S, Gp), flush=True)
A = np.array([[1.0, s, g] for _, _, s, g in rows])
y = np.array([t for _, t, _, _ in rows])
coef, res, _, _ = np.linalg.lstsq(A, y, rcond=None)
print("fit: fixed %.4f ms, per value pass %.5f ms, per grad pass %.5f ms (per evaluation of 10^4 samples); residual %s" % (
    coef[0], coef[1], coef[2], res))
ntiles = 16
per_wave_ns = 2048.0 / 1e4 * 1e6 / 2.0  # ms per eval -> ns of SIMD time per sample (2048 waves share 1024 SIMDs)
print("per sample-wave on its SIMD share: fixed %.0f ns, value pass %.0f ns (%.1f ns per tile), grad pass %.0f ns (%.1f per tile)" % (
    coef[0] * per_wave_ns, coef[1] * per_wave_ns, coef[1] * per_wave_ns / ntiles, coef[2] * per_wave_ns,
    coef[2] * per_wave_ns / ntiles))
