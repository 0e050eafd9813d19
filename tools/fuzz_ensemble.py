"""Randomised check of the ensemble-wide launches (csrc/launch.hpp, mcmc.hip: replay_ensemble): random ensembles -- size, GP shape,
observed derivatives, fidelity dimensions, points being sampled, inner domain, batch size, value-only calls -- evaluated with every
kernel issued once for all members and member by member; KG and grad KG must agree BIT FOR BIT.
    python tools/fuzz_ensemble.py [num_cases = 60] [seed = 2026] [n_max = 400]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cornell_moe_amd import api  # noqa: E402
from cornell_moe_amd.workloads import make_workload  # noqa: E402


def run(num_cases=60, seed=2026, n_max=400):
    rng = np.random.default_rng(seed)
    bad = merged_calls = 0
    for case in range(num_cases):
        d = int(rng.integers(1, 9))
        g = int(rng.integers(0, min(3, d) + 1)) if rng.uniform() < 0.35 else 0
        derivs = tuple(int(v) for v in rng.permutation(d)[:g])
        q = int(rng.integers(1, 5))
        p = int(rng.integers(0, 3))
        n = int(rng.integers(2, n_max if rng.uniform() < 0.3 else 60))
        P = int(rng.integers(1, 13))
        M = int(rng.choice([2, 16, 64, 128, 300]))
        nm = int(rng.integers(2, 9))
        f = int(rng.integers(1, d)) if (d > 1 and rng.uniform() < 0.25) else 0
        E = int(rng.integers(1, 8))
        simplex = 1 if (d - f >= 2 and rng.uniform() < 0.2) else 0
        w = make_workload(seed=20_000 + case, n=n, d=d, q=q, M=M, P=P, derivs=derivs, p=p)
        hypers = np.column_stack([w.alpha * rng.uniform(0.7, 1.4, nm)] + [w.lengths[k] * rng.uniform(0.5, 2.0, nm) for k in range(d)])
        noises = np.tile(np.asarray(w.noise, dtype=np.float64).reshape(1, -1), (nm, 1)) * rng.uniform(0.8, 1.2, (nm, 1))
        Xq_all = rng.uniform(0.05, 0.45 if simplex else 0.95, (E, q, d))
        disc0 = w.discrete[:, :d - f]
        disc = np.tile(disc0.reshape(1, -1), (nm, 1)) + 0.01 * rng.standard_normal((nm, disc0.size))
        best = rng.uniform(-1.0, 0.0, nm)
        gd = tuple(w.inner_gd[:8]) + (simplex,)
        bounds = w.bounds[:2 * (d - f)]
        Xp = w.Xp if p else None
        try:
            G = api.DeviceGPMCMC(hypers, noises, w.X, w.y, derivs)
        except api.OptimalLearningException:
            continue
        want_grad = bool(rng.uniform() < 0.8)
        out = []
        try:
            for on in (0, 1, 1):
                api.set_ensemble_launches(on)
                s0 = api.ensemble_launch_stats()
                try:
                    out.append(G.kg_batch(gd, bounds, disc, Xq_all, Xp, M, best, w.kg_normals, want_grad=want_grad, num_fidelity=f))
                except api.OptimalLearningException as e:
                    out.append(("raised", type(e).__name__))
                s1 = api.ensemble_launch_stats()
                if on:
                    merged_calls += s1[0] - s0[0]
        finally:
            api.set_ensemble_launches(-1)
        ok = True
        for o in out[1:]:
            if isinstance(out[0][0], str) or isinstance(o[0], str):
                ok = ok and out[0] == o
            else:
                ok = ok and np.array_equal(out[0][0], o[0]) and (not want_grad or np.array_equal(out[0][1], o[1]))
        if not ok:
            bad += 1
            print("ENSEMBLE MISMATCH case %d: n=%d d=%d q=%d p=%d g=%s f=%d P=%d M=%d members=%d E=%d simplex=%d grad=%s" % (
                case, n, d, q, p, derivs, f, P, M, nm, E, simplex, want_grad), flush=True)
    print("ensemble fuzz: %d cases, %d mismatches; %d evaluations went down merged" % (num_cases, bad, merged_calls))
    return bad


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    sys.exit(1 if run(*a) else 0)
