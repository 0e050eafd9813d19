"""Small driver for rocprofv3: a few batched q-KG gradient evaluations at a named config (no torch, no CPU baseline).
    python tools/prof_kg.py [config] [restarts] [repeats]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cornell_moe_amd.api import DeviceGP  # noqa: E402
from cornell_moe_amd.workloads import make_workload  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
R = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
over = {}
if ":" in cfg:  # e.g. C3:n=60,M=2000  -- override fields of a named configuration
    cfg, tail = cfg.split(":", 1)
    for kv in tail.split(","):
        k, v = kv.split("=")
        over[k] = int(v)
steps = over.pop("steps", None)   # inner GD steps per restart (instruction-count fits: tools/mc_overhead.sh)
if "g" in over:                   # observe the first g partial derivatives (bench.py --derivs)
    over["derivs"] = tuple(range(over.pop("g")))
w = make_workload(cfg, num_restarts=R, **over)
if steps is not None:
    w.inner_gd = (w.inner_gd[0], steps) + tuple(w.inner_gd[2:])
G = DeviceGP(w.hyperparameters, w.X, w.y, w.noise, w.derivs)
best = float(G.additional_mean(w.discrete).min())
for i in range(reps):
    t0 = time.perf_counter()
    r = G.kg_batch(w.inner_gd, w.bounds, w.discrete, w.Xq_restarts, None, w.M, best, w.kg_normals)
    dt = time.perf_counter() - t0
    km = G.last_kernel_ms()
    print("rep %d: %.3f ms/eval wall; per-eval kernel ms: mc %.4f cov %.4f tail %.4f state %.4f; passes/sample: value %.2f grad %.2f"
          % (i, 1e3 * dt / R, km["mc"], km["cov_build"], km["tail"], km["state"], r["mean_evals"] / (R * w.M),
             r["grad_evals"] / (R * w.M)), flush=True)

# the covariance-assembly kernel at the N x (R M) shape of bench.py's roofline_cov_build (the q-KG tail itself no longer
# materialises this matrix), so that the PMC passes see it
if cfg == "C3":
    import numpy as np
    Rp = min(R, 8)   # (bench.py's roofline_cov_build shape: at most 8 evaluations' worth of columns, 640 MB)
    pts = np.random.default_rng(7).uniform(size=(Rp * w.M, w.d))
    ms, nbytes = G.cov_build_probe(pts, repeat=2)
    print("cov_build probe %d x %d: %.4f ms per launch, %.0f GB/s" % (w.n, Rp * w.M, ms, nbytes / ms / 1e6), flush=True)
