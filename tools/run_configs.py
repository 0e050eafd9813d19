"""Measure every BASELINE.json configuration that fits one GPU, with the reference CPU path beside it where that is
affordable: prints one JSON object (kept under profiles/ as rNN_configs.json).

  C1  GP posterior mean/var, n=200 d=2 (plumbing)            GPU vs reference, 100 query points
  C2  q-EI value+grad, n=500 d=4 q=2, 1k MC                   GPU evals/s vs reference (1 core)
  C3  q-KG value+grad, n=1000 d=8 q=4, 10k MC (headline)      see bench.py; here: single evaluation and batch of 8
  C4  64 multistarts x C3 on ONE GPU                          one moe_kg_batch of 64 evaluations
  C5  d-KG, n=2000 d=12 q=8 g=3, 20k MC                       GPU only (the reference needs ~hours: see note)
Also: covariance-build HBM roofline at sizes >> cache through moe_cov_build_probe."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cornell_moe_amd.api import DeviceGP  # noqa: E402
from cornell_moe_amd.workloads import make_workload  # noqa: E402

try:
    from oracle import ref
    HAVE_REF = ref.available()
except Exception:  # pragma: no cover
    HAVE_REF = False


def timeit(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


out = {}

# ---- C1 ----
w = make_workload("C1")
t0 = time.perf_counter()
G = DeviceGP(w.hyperparameters, w.X, w.y, w.noise, ())
t_build = time.perf_counter() - t0
t_mean = timeit(lambda: G.mean(w.query))
t_var = timeit(lambda: G.variance(w.query[:20]))
c1 = {"gpu_build_s": t_build, "gpu_mean_100pts_s": t_mean, "gpu_var_20pts_s": t_var}
if HAVE_REF:
    t0 = time.perf_counter()
    R = ref.RefGP(1, w.alpha, w.lengths, w.X, w.y, w.noise, ())
    c1["ref_build_s"] = time.perf_counter() - t0
    c1["ref_mean_100pts_s"] = timeit(lambda: R.mean(w.query))
    c1["ref_var_20pts_s"] = timeit(lambda: R.var(w.query[:20]))
    c1["max_rel_err_mean"] = float(np.abs(G.mean(w.query) - R.mean(w.query)).max() / np.abs(R.mean(w.query)).max())
out["C1"] = c1

# ---- C2 ----
w = make_workload("C2")
G = DeviceGP(w.hyperparameters, w.X, w.y, w.noise, ())
best = float(np.min(w.y[:, 0])) + 0.5
t = timeit(lambda: G.ei(w.Xq, None, w.M, best, w.ei_normals), reps=20)
c2 = {"gpu_ei_grad_evals_per_s": 1.0 / t, "gpu_ms_per_eval": 1e3 * t}
Xq64 = np.random.default_rng(2).uniform(0.05, 0.95, size=(64,) + w.Xq.shape)
tb = timeit(lambda: G.ei_batch(Xq64, None, w.M, best, w.ei_normals), reps=10)
c2["gpu_batch_64_evals_per_s"] = 64.0 / tb
# SURVEY 8(d)'s roofline entry for q-EI: M (u^2 + 2u + 2 q d u) flops over 8 [M u + d u^2 q] bytes per evaluation -- 40 kflop and
# 16 KB at C2: the evaluation is launch / sync latency (one sample per lane, 4 workgroups), so the fraction says how far a
# latency-bound call is from either roof, nothing about the kernel
u_, q_, d_ = w.q, w.q, w.d
fl = w.M * (u_ * u_ + 2 * u_ + 2 * q_ * d_ * u_)
by = 8.0 * (w.M * u_ + d_ * u_ * u_ * q_)
c2["roofline"] = {"bound": "hbm", "algorithmic_bytes_per_eval": by, "algorithmic_flops_per_eval": fl,
                  "achieved_GB_per_s_batch64": by * 64.0 / tb / 1e9, "frac_of_8TBs_batch64": by * 64.0 / tb / 8e12,
                  "achieved_GFLOP_per_s_batch64": fl * 64.0 / tb / 1e9,
                  "note": "latency-bound: 16 KB and 40 kflop per evaluation; one at a time %.0f us per call" % (1e6 * t)}
if HAVE_REF:
    R = ref.RefGP(1, w.alpha, w.lengths, w.X, w.y, w.noise, ())
    tr = timeit(lambda: R.ei(w.Xq, None, w.M, best, w.ei_normals), reps=5)
    c2["ref_1core_evals_per_s"] = 1.0 / tr
    eo, go = R.ei(w.Xq, None, w.M, best, w.ei_normals)[:2]
    eg, gg = G.ei(w.Xq, None, w.M, best, w.ei_normals)
    c2["rel_err_ei"] = abs(eo - eg) / max(abs(eo), 1e-300)
    c2["max_abs_err_grad"] = float(np.abs(go - gg).max())
out["C2"] = c2

# ---- C3 / C4 ----
w = make_workload("C3", num_restarts=64)
G = DeviceGP(w.hyperparameters, w.X, w.y, w.noise, ())
best = float(G.additional_mean(w.discrete).min())
c3 = {}
for R_ in (1, 8, 64):
    t = timeit(lambda: G.kg_batch(w.inner_gd, w.bounds, w.discrete, w.Xq_restarts[:R_], None, w.M, best, w.kg_normals), reps=3)
    c3["batch_%d_evals_per_s" % R_] = R_ / t
    c3["batch_%d_ms_per_eval" % R_] = 1e3 * t / R_
out["C3"] = {k: v for k, v in c3.items() if not k.startswith("batch_64")}
out["C4_one_gpu"] = {"wall_s_64_multistarts": 64.0 / c3["batch_64_evals_per_s"], "evals_per_s": c3["batch_64_evals_per_s"]}
# covariance build N x M roofline at the C3 tail shape, M = 80000 columns (640 MB >> caches)
ms, nbytes = G.cov_build_probe(np.random.default_rng(0).uniform(size=(80000, 8)), repeat=10)
out["cov_build_N1000xM80000"] = {"ms": ms, "GB_per_s": nbytes / ms / 1e6, "frac_of_8TBs": nbytes / ms / 1e6 / 8000.0}

# ---- MCMC-averaged q-KG: what examples/main.py runs -- 16 hyper-parameter samples x the C3 shape ----
from cornell_moe_amd.api import DeviceGPMCMC  # noqa: E402
w = make_workload("C3", num_restarts=8)
rng = np.random.default_rng(16)
nm = 16
hyp = np.c_[rng.uniform(0.8, 1.2, nm), rng.uniform(0.6, 0.8, size=(nm, w.d))]
noi = np.full((nm, 1), 0.01)
t0 = time.perf_counter()
GM = DeviceGPMCMC(hyp, noi, w.X, w.y, ())
t_build = time.perf_counter() - t0
disc_all = np.stack([w.discrete] * nm)
best_all = np.array([float(g.additional_mean(w.discrete).min()) for g in GM.gps])
mc = {"num_mcmc": nm, "gpu_build_16_gps_s": t_build}
for R_ in (1, 8):
    t = timeit(lambda: GM.kg_batch(w.inner_gd, w.bounds, disc_all, w.Xq_restarts[:R_], None, w.M, best_all, w.kg_normals), reps=2)
    mc["batch_%d_mcmc_evals_per_s" % R_] = R_ / t
    mc["batch_%d_per_gp_evals_per_s" % R_] = R_ * nm / t
out["C3_x16_mcmc"] = mc
del GM

# ---- C5 ----
w = make_workload("C5")
t0 = time.perf_counter()
G = DeviceGP(w.hyperparameters, w.X, w.y, w.noise, w.derivs)
c5 = {"gpu_build_s_N8000_first": time.perf_counter() - t0}
del G
t0 = time.perf_counter()
G = DeviceGP(w.hyperparameters, w.X, w.y, w.noise, w.derivs)
c5["gpu_build_s_N8000"] = time.perf_counter() - t0
best = float(G.additional_mean(w.discrete).min())
t = timeit(lambda: G.kg(w.inner_gd, w.bounds, w.discrete, w.Xq, None, w.M, best, w.kg_normals), reps=2)
c5["gpu_dkg_grad_evals_per_s"] = 1.0 / t
c5["gpu_ms_per_eval"] = 1e3 * t
km = G.last_kernel_ms()
c5["kernel_ms"] = {k: float(v) for k, v in km.items()}
c5["note"] = ("reference CPU path not timed at this size: one evaluation re-factorises the (N+m)^2 = 8032^2 fantasy matrix "
              "(1.7e11 flop at ~2 GFLOP/s) and re-solves it for each of the 20000 samples (2.6e12 flop): hours")
Xq4 = np.random.default_rng(5).uniform(0.05, 0.95, size=(4,) + w.Xq.shape)
t4 = timeit(lambda: G.kg_batch(w.inner_gd, w.bounds, w.discrete, Xq4, None, w.M, best, w.kg_normals), reps=2)
c5["gpu_batch_4_evals_per_s"] = 4.0 / t4
t0 = time.perf_counter()
G.add_points(np.random.default_rng(6).uniform(size=(4, 12)), np.zeros((4, 4)))
c5["gpu_add_4_points_s"] = time.perf_counter() - t0
ms, nbytes = G.cov_build_probe(np.random.default_rng(1).uniform(size=(20000, 12)), repeat=5)
out["cov_build_N8000xM20000_derivative_rows"] = {"ms": ms, "GB_per_s": nbytes / ms / 1e6, "frac_of_8TBs": nbytes / ms / 1e6 / 8000.0}
out["C5"] = c5

# ---- log marginal likelihood and its hyper-parameter gradient at C3's data ----
from cornell_moe_amd.api import LogLikelihood  # noqa: E402
w = make_workload("C3")
LL = LogLikelihood(w.X, w.y, ())
sets = np.tile(np.r_[w.hyperparameters, w.noise], (64, 1)) * np.linspace(0.7, 1.3, 64)[:, None]
LL.evaluate(sets)
ll = {}
for k in (1, 8, 64):
    ll["ms_per_set_in_calls_of_%d" % k] = 1e3 * timeit(lambda: LL.evaluate(sets[:k]), reps=5) / k
ll["grad_ms_per_set"] = 1e3 * timeit(lambda: LL.grad(sets[0]), reps=5)
if HAVE_REF:
    t0 = time.perf_counter()
    ref.log_likelihood(1, w.alpha, w.lengths, w.X, w.y, w.noise, [])
    ll["ref_1core_ms_per_set"] = 1e3 * (time.perf_counter() - t0)
out["log_likelihood_n1000"] = ll
w = make_workload("C5")
LL = LogLikelihood(w.X, w.y, w.derivs)
th = np.r_[w.hyperparameters, w.noise]
LL.evaluate(th[None])
out["log_likelihood_N8000"] = {"ms_per_set": 1e3 * timeit(lambda: LL.evaluate(th[None]), reps=3),
                               "grad_ms_per_set": 1e3 * timeit(lambda: LL.grad(th), reps=2)}
print(json.dumps(out, indent=1))
