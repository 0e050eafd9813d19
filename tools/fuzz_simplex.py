"""Randomised parity sweep of KG with the INNER optimisations over the simplex domain (r4): random shapes / kernels / derivative sets /
fidelity coordinates / optimiser settings, points inside the unit simplex, data that pulls the optimisations towards the diagonal face;
device (the kernel it picks and the workgroup-per-sample kernel) against the unmodified reference's
KnowledgeGradientEvaluator<SimplexIntersectTensorProductDomain> (oracle/_ref) on the same normal tables.
    python tools/fuzz_simplex.py [num_cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cornell_moe_amd import api  # noqa: E402
from helpers import TOL  # noqa: E402
from oracle import ref  # noqa: E402


def run(num_cases=60, seed=4):
    rng = np.random.default_rng(seed)
    bad = on_face_total = 0
    for case in range(num_cases):
        d = int(rng.integers(2, 8))
        f = int(rng.integers(1, d)) if (rng.uniform() < 0.25 and d > 2) else 0
        size = d - f
        g = int(rng.integers(0, min(3, d) + 1)) if rng.uniform() < 0.4 else 0
        derivs = tuple(int(v) for v in rng.permutation(d)[:g])
        q, p = int(rng.integers(1, 4)), int(rng.integers(0, 2))
        n, P, M = int(rng.integers(8, 160)), int(rng.integers(2, 10)), int(rng.integers(4, 41))
        cov = int(rng.integers(0, 2))
        lo = float(rng.choice([0.0, -0.2, 0.05]))
        hi = float(rng.choice([1.0, 0.7, 1.3]))
        gd = (1, int(rng.integers(2, 13)), int(rng.integers(1, 3)), 3, float(rng.choice([0.0, 0.5])), float(rng.choice([1.0, 0.3, 2.0])),
              float(rng.choice([0.1, 0.5, 1.0])), float(rng.choice([1e-10, 1e-6])))

        def simplex_pts(count, top):
            pts = []
            while len(pts) < count:
                x = rng.uniform(max(lo, 0.0) + 0.005, min(hi, 1.0), size=size)
                if x.sum() <= top:
                    pts.append(x)
            return np.array(pts).reshape(count, size)
        if max(lo, 0.0) * size >= 0.9:
            continue
        fid = lambda k: rng.uniform(0.3, 1.0, size=(k, f))  # noqa: E731
        Xs = simplex_pts(n, 0.92)
        X = np.hstack([Xs, fid(n)]) if f else Xs
        y = np.zeros((n, 1 + g))
        pull = float(rng.choice([-1.5, -0.5, 1.0]))   # (negative: the posterior mean falls towards the diagonal face)
        y[:, 0] = pull * Xs.sum(1) + 0.3 * np.sin(5 * Xs).sum(1) + 0.05 * rng.uniform(size=n)
        for a, dd in enumerate(derivs):
            y[:, 1 + a] = (pull if dd < size else 0.0) + 1.5 * np.cos(5 * X[:, dd]) * (dd < size)
        lengths, noise, alpha = rng.uniform(0.25, 0.8, size=d), np.full(1 + g, 0.02), float(rng.uniform(0.7, 1.5))
        bounds = np.tile([lo, hi], size)
        Xq = np.hstack([simplex_pts(q, 0.85), fid(q)]) if f else simplex_pts(q, 0.85)
        Xp = (np.hstack([simplex_pts(p, 0.85), fid(p)]) if f else simplex_pts(p, 0.85)) if p else None
        disc = simplex_pts(P, 0.9)
        m = (q + p) * (1 + g)
        normals = rng.standard_normal(((M + 1) // 2, m))
        try:
            R = ref.RefGP(cov, alpha, lengths, X, y, noise, list(derivs))
        except Exception:
            continue
        full = np.hstack([disc, np.ones((P, f))]) if f else disc
        best = float(R.additional_mean(full).min())
        rc = R.kg(gd, bounds, disc, Xq, Xp, M, best, normals, num_fidelity=f, domain_type=1)
        G = api.DeviceGP(np.r_[alpha, lengths], X, y, noise, derivs, cov_type=cov)
        scale = max(float(np.abs(rc["grad"]).max()), abs(rc["kg"]), 1e-6)
        loose = gd[1] * gd[2] > 8
        on_face = rc["best_point"][:, :size].sum(axis=1) > 0.99
        on_face_total += int(on_face.sum())
        for variant in ("auto", "1"):
            if variant == "auto":
                os.environ.pop("MOE_KG_VARIANT", None)
            else:
                os.environ["MOE_KG_VARIANT"] = variant
            try:
                rg = G.kg(gd + (1,), bounds, disc, Xq, Xp, M, best, normals, num_fidelity=f, want_best_points=True)
            except api.OptimalLearningException as e:
                if "too large" in str(e):
                    continue
                raise
            e_kg = abs(rg["kg"] - rc["kg"]) / max(abs(rc["kg"]), 1e-6)
            e_gr = float(np.abs(rg["grad"] - rc["grad"]).max()) / scale
            off = np.abs(rg["best_point"][:, :size] - rc["best_point"][:, :size]).max(axis=1)
            ptol = 1e-6 if loose else 1e-8
            mism = float((off > ptol).mean())
            gtol = max(1e-6 if loose else TOL["grad_kg"], 2.0e-5 * mism)
            inside = rg["best_point"][:, :size].min() >= -1e-12 and rg["best_point"][:, :size].sum(axis=1).max() <= 1.0 + 1e-12
            if e_kg > TOL["kg"] or e_gr > gtol or mism > max(0.05, 2.5 / M) or not inside or (G.last_kernel_info()["variant"] == 0 and not G.last_kernel_info()["lane"]):   # (r6: the lane-parked kernel carries the simplex update)
                bad += 1
                print("SIMPLEX KG MISMATCH case %d variant %s (kernel %d): n=%d d=%d f=%d g=%s q=%d p=%d P=%d M=%d cov=%d box=[%g, %g] gd=%s: "
                      "rel kg %.2e grad %.2e end points off %.3f (max %.2e; on the face: %d of %d) inside=%s" % (
                          case, variant, G.last_kernel_info()["variant"], n, d, f, derivs, q, p, P, M, cov, lo, hi, gd, e_kg, e_gr, mism,
                          off.max(), int(on_face.sum()), M, inside), flush=True)
        G.close()
    os.environ.pop("MOE_KG_VARIANT", None)
    print("simplex fuzz: %d cases, %d violations; %d reference end points on the diagonal face" % (num_cases, bad, on_face_total))
    return bad


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    sys.exit(1 if run(*a) else 0)
