"""Generate tests/golden/ref_fixtures.npz from the UNMODIFIED reference (oracle/_ref, built from /root/reference).

Run where /root/reference exists:   python tools/make_golden.py
The fixtures pin oracle/moe_oracle.c and the device path on boxes where the reference is absent.  Cases follow the shapes
of the reference's own ping-test fixtures (gpp_knowledge_gradient_optimization_test.cpp:384-441, gpp_math_test.cpp:1446-1702):
dim=3, num_sampled=7, q in {1,2,3}, p in {0,2}, num_pts=5, derivatives {0,1,2} or none, 16 MC iterations, alpha=2.80723,
lengths ~ U(0.5,2.5), data ~ U(-5,5), noise 0.1, best_so_far=7.0 -- plus a few larger synthetic cases.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "ref_fixtures.npz")


def make_case(seed, n, d, q, p, P, derivs, M, cov_type, alpha, noise, best_so_far, box, inner_gd, data_scale=5.0,
              length_range=(0.5, 2.5)):
    rng = np.random.default_rng(seed)
    g = len(derivs)
    c = dict(seed=seed, n=n, d=d, q=q, p=p, P=P, derivs=np.array(derivs, dtype=np.int32), M=M, cov_type=cov_type,
             alpha=alpha, best_so_far=best_so_far, inner_gd=np.array(inner_gd, dtype=np.float64))
    c["lengths"] = rng.uniform(length_range[0], length_range[1], size=d)
    c["X"] = rng.uniform(box[0], box[1], size=(n, d))
    c["y"] = rng.uniform(-data_scale, data_scale, size=(n, 1 + g))
    c["noise"] = np.full(1 + g, noise)
    c["bounds"] = np.tile(np.array(box, dtype=np.float64), d)
    c["Xq"] = rng.uniform(box[0], box[1], size=(q, d))
    c["Xp"] = rng.uniform(box[0], box[1], size=(p, d))
    c["discrete"] = rng.uniform(box[0], box[1], size=(P, d))
    c["query"] = rng.uniform(box[0], box[1], size=(4, d))
    m = (q + p) * (1 + g)
    c["kg_normals"] = rng.standard_normal(size=((M + 1) // 2, m))
    c["ei_normals"] = rng.standard_normal(size=(M, q + p))
    return c


def run_case(c):
    gp = ref.RefGP(int(c["cov_type"]), float(c["alpha"]), c["lengths"], c["X"], c["y"], c["noise"], list(c["derivs"]))
    out = {}
    K, kiy, mean = gp.dump()
    out["K_chol"] = np.tril(K)
    out["K_inv_y"] = kiy
    out["mean"] = np.array(mean)
    pts = c["query"]
    g = len(c["derivs"])
    m4 = 4 * (1 + g)
    out["q_mean"] = gp.mean(pts)
    out["q_grad_mean"] = gp.grad_mean(pts)
    out["q_var"] = gp.var(pts)
    out["q_chol_var"] = np.tril(gp.chol_var(pts).reshape(m4, m4).T)
    out["q_grad_var"] = gp.grad_var(pts, 2)
    out["q_grad_chol_var"] = gp.grad_chol_var(pts, 2)
    out["q_mix_cov"] = gp.mix_cov(pts, list(c["derivs"]))
    pm, pg = gp.posterior_mean(pts[0])
    out["post_mean"] = np.array(pm)
    out["post_grad"] = pg
    Xp = c["Xp"] if c["p"] > 0 else None
    ei, gei, _ = gp.ei(c["Xq"], Xp, int(c["M"]), float(c["ei_best"]), c["ei_normals"])
    out["ei"] = np.array(ei)
    out["grad_ei"] = gei
    # analytic 1,0-EI at each query point (OnePotentialSampleExpectedImprovementEvaluator, gpp_math.cpp:2195-2259)
    a = [gp.ei_analytic(pt, float(c["ei_best"])) for pt in pts]
    out["ei_analytic"] = np.array([v for v, _ in a])
    out["grad_ei_analytic"] = np.array([gr for _, gr in a])
    if "ms_starts" in c:  # 1,0-EI multistart gradient descent from a fixed start set (gpp_math.hpp:1683-1742)
        best, found = gp.ei_multistart_analytic(c["ms_gd"], c["bounds"], c["ms_starts"], float(c["ei_best"]))
        out["ms_best_point"] = best
        out["ms_found"] = np.array(int(found))
        out["ms_best_ei"] = np.array(gp.ei_analytic(best, float(c["ei_best"]))[0])
    # log marginal likelihood at the case's own hyper-parameters and at two perturbed sets (gpp_model_selection.cpp:540-612)
    lls = []
    for scale in (1.0, 0.7, 1.6):
        lls.append(ref.log_likelihood(int(c["cov_type"]), float(c["alpha"]) * scale, c["lengths"] * scale, c["X"], c["y"],
                                      c["noise"] * scale, list(c["derivs"])))
    out["log_likelihood"] = np.array(lls)
    r = gp.kg(c["inner_gd"], c["bounds"], c["discrete"], c["Xq"], Xp, int(c["M"]), float(c["best_so_far"]), c["kg_normals"],
              want_grad=True, details=True)
    out["kg"] = np.array(r["kg"])
    out["grad_kg"] = r["grad"]
    out["kg_best_point"] = r["best_point"]
    out["kg_to_sample_mean"] = r["to_sample_mean"]
    out["kg_chol_var"] = r["chol_var"]
    rv = gp.kg(c["inner_gd"], c["bounds"], c["discrete"], c["Xq"], Xp, int(c["M"]), float(c["best_so_far"]), c["kg_normals"],
               want_grad=False)
    out["kg_value_only"] = np.array(rv["kg"])
    return out


def mcmc_cases():
    """MCMC-averaged evaluators: small ensembles, with and without derivative observations / fidelity
    dimensions, from the reference's GaussianProcessMCMC + KnowledgeGradientMCMCEvaluator / ExpectedImprovementMCMCEvaluator."""
    from oracle import orc
    out = []
    rng = np.random.default_rng(4242)
    n, d, q, p, P, M, nm = 30, 4, 2, 1, 5, 24, 3
    for derivs, f in (((), 0), ((1,), 1), ((0, 2), 2)):
        g = len(derivs)
        c = dict(n=n, d=d, q=q, p=p, P=P, M=M, num_mcmc=nm, derivs=np.array(derivs, dtype=np.int32), num_fidelity=f)
        c["X"] = rng.uniform(0.05, 1.0, size=(n, d))
        c["y"] = rng.uniform(-1.0, 1.0, size=(n, 1 + g))
        c["hypers"] = np.c_[rng.uniform(0.8, 1.5, nm), rng.uniform(0.4, 0.9, size=(nm, d))]
        c["noises"] = rng.uniform(0.01, 0.1, size=(nm, 1 + g))
        c["Xq"] = rng.uniform(0.2, 0.9, size=(q, d))
        c["Xp"] = rng.uniform(0.2, 0.9, size=(p, d))
        c["discrete"] = rng.uniform(0.0, 1.0, size=(nm, P, d - f))
        c["kg_best"] = rng.uniform(-0.5, 0.5, nm)
        c["ei_best"] = np.full(nm, float(np.median(c["y"][:, 0])))
        c["kg_normals"] = rng.standard_normal(((M + 1) // 2, (q + p) * (1 + g)))
        c["ei_normals"] = rng.standard_normal((M, q + p))
        c["inner_gd"] = np.array((1, 6, 1, 3, 0.0, 1.0, 0.1, 1e-10))
        c["bounds"] = np.tile([0.0, 1.0], d)
        c["ms_starts"] = rng.uniform(0.0, 1.0, size=(24, d))
        c["ms_gd"] = np.array((24, 20, 2, 4, 0.7, 0.05, 0.2, 1e-8))
        R = ref.RefGPMCMC(c["hypers"], c["noises"], c["X"], c["y"], derivs)
        o = {}
        kg, gkg = R.kg(c["inner_gd"], c["bounds"][:2 * (d - f)], c["discrete"], c["Xq"], c["Xp"], M, c["kg_best"], c["kg_normals"],
                       num_fidelity=f)
        o["kg"], o["grad_kg"] = np.array(kg), gkg
        o["kg_value_only"] = np.array(R.kg(c["inner_gd"], c["bounds"][:2 * (d - f)], c["discrete"], c["Xq"], c["Xp"], M, c["kg_best"],
                                           c["kg_normals"], want_grad=False, num_fidelity=f)[0])
        ei, gei = R.ei(c["Xq"], c["Xp"], M, c["ei_best"], c["ei_normals"])
        o["ei"], o["grad_ei"] = np.array(ei), gei
        best, found = R.ei_multistart_analytic(c["ms_gd"], c["bounds"], c["ms_starts"], c["ei_best"])
        o["ms_best_point"], o["ms_found"] = best, np.array(int(found))
        O = orc.OrcGPMCMC(c["hypers"], c["noises"], c["X"], c["y"], derivs)
        o["ms_best_ei"] = np.array(O.ei_analytic(best, c["ei_best"], want_grad=False)[0])
        print("mcmc case: g=%d f=%d  KG=%.12g  EI=%.12g  ms_best_ei=%.6g found=%d" % (g, f, kg, ei, float(o["ms_best_ei"]), found))
        out.append((c, o))
    return out


OUT_LL = os.path.join(ROOT, "tests", "golden", "ref_ll_grad.npz")


def ll_grad_fixtures():
    """Hyper-parameter gradients of the log marginal likelihood from the reference's
    LogMarginalLikelihoodEvaluator::ComputeGradLogLikelihood: Matern-5/2 with and without derivative observations (the
    kernel the Python boundary builds) and the squared exponential without, at a few sizes and hyper-parameter sets."""
    rng = np.random.default_rng(777)
    blob, k = {}, 0
    for (n, d, derivs, cov) in ((7, 3, (), 1), (7, 3, (0, 1, 2), 1), (40, 3, (0, 2), 1), (60, 4, (), 1), (60, 4, (), 0),
                                (120, 6, (), 1), (90, 8, (1, 3), 1), (150, 10, (), 1)):
        g = len(derivs)
        X = rng.uniform(size=(n, d))
        y = np.sin(3 * X).sum(1, keepdims=True) + 0.1 * rng.uniform(size=(n, 1))
        if g:
            y = np.hstack([y] + [3 * np.cos(3 * X[:, [j]]) for j in derivs])
        for rep in range(2):
            alpha = float(rng.uniform(0.5, 2.5))
            lengths = rng.uniform(0.3, 1.5, size=d)
            noise = rng.uniform(0.005, 0.2, size=1 + g)
            grad = ref.log_likelihood_grad(cov, alpha, lengths, X, y, noise, list(derivs))
            val = ref.log_likelihood(cov, alpha, lengths, X, y, noise, list(derivs))
            for key, v in (("X", X), ("y", y), ("derivs", np.array(derivs, dtype=np.int32)), ("cov_type", cov), ("alpha", alpha),
                           ("lengths", lengths), ("noise", noise), ("grad", grad), ("value", val)):
                blob["g%d_%s" % (k, key)] = np.asarray(v)
            k += 1
    blob["num"] = np.array(k)
    np.savez_compressed(OUT_LL, **blob)
    print("wrote", OUT_LL, os.path.getsize(OUT_LL), "bytes,", k, "cases")


OUT_SHAPES = os.path.join(ROOT, "tests", "golden", "ref_shapes.npz")


def shape_fixtures():
    """The BENCHMARKED shapes, from the unmodified reference: C2 in full (q-EI, n=500, d=4, q=2, M=1000),
    the C3 shape (q-KG, n=1000, d=8, q=4, P=10) at M=1000, and a C5-like d-KG case (d=12, q=8, g=3, P=50, n=300, M=200).
    Inputs are regenerated by cornell_moe_amd.workloads.make_workload from the stored keyword arguments (numpy's
    default_rng streams are stable across versions); a checksum of X and of the normal table guards against drift."""
    from cornell_moe_amd.workloads import make_workload
    blob = {}

    def checksum(w):
        return np.array([float(w.X.sum()), float(w.kg_normals.sum()), float(w.Xq.sum()), float(w.discrete.sum())])

    # C2 in full
    w = make_workload("C2")
    gp = ref.RefGP(1, w.alpha, w.lengths, w.X, w.y, w.noise, [])
    best = float(w.y[:, 0].min())
    ei, gei, _ = gp.ei(w.Xq, None, w.M, best, w.ei_normals)
    blob["c2_ei"], blob["c2_grad_ei"], blob["c2_best"], blob["c2_check"] = np.array(ei), gei, np.array(best), checksum(w)
    # (with best_so_far = min y the improvement is 0 for every sample at these points: also pin a live case, best = median y)
    best_med = float(np.median(w.y[:, 0]))
    ei_m, gei_m, _ = gp.ei(w.Xq, None, w.M, best_med, w.ei_normals)
    blob["c2_ei_median"], blob["c2_grad_ei_median"], blob["c2_best_median"] = np.array(ei_m), gei_m, np.array(best_med)
    print("C2 (best = median y): EI=%.15g" % ei_m)
    blob["c2_mean"] = gp.mean(w.query)
    blob["c2_var"] = gp.var(w.query[:4])
    print("C2: EI=%.15g" % ei)
    # C3 shape at M = 1000
    for tag, kw in (("c3", dict(name="C3", M=1000)),
                    ("c5", dict(name="C5", n=300, M=200))):
        w = make_workload(**kw)
        gp = ref.RefGP(1, w.alpha, w.lengths, w.X, w.y, w.noise, list(w.derivs))
        best = float(gp.additional_mean(w.discrete).min())
        r = gp.kg(w.inner_gd, w.bounds, w.discrete, w.Xq, None, w.M, best, w.kg_normals, want_grad=True, details=True)
        rv = gp.kg(w.inner_gd, w.bounds, w.discrete, w.Xq, None, w.M, best, w.kg_normals, want_grad=False)
        blob[tag + "_kw_n"], blob[tag + "_kw_M"] = np.array(w.n), np.array(w.M)
        blob[tag + "_best_so_far"] = np.array(best)
        blob[tag + "_kg"], blob[tag + "_grad_kg"], blob[tag + "_best_point"] = np.array(r["kg"]), r["grad"], r["best_point"]
        blob[tag + "_kg_value_only"] = np.array(rv["kg"])
        blob[tag + "_check"] = checksum(w)
        blob[tag + "_seconds"] = np.array(r["seconds"])
        print("%s: n=%d d=%d q=%d g=%d M=%d  KG=%.15g  (reference: %.2f s state + %.2f s evaluation)" % (
            tag, w.n, w.d, w.q, w.g, w.M, r["kg"], r["seconds"][0], r["seconds"][1]))
    np.savez_compressed(OUT_SHAPES, **blob)
    print("wrote", OUT_SHAPES, os.path.getsize(OUT_SHAPES), "bytes")


OUT_SHAPES3 = os.path.join(ROOT, "tests", "golden", "ref_shapes_r3.npz")

from cornell_moe_amd.workloads import R3_PARITY_CASES as R3_CASES  # noqa: E402  (shared with tests/)


def shape_fixtures_r3():
    """Round 3: the EXACT headline configuration (C3 at M = 10 000 -- the reference needs ~20 s for it), C5's d-KG at n = 1000
    (N = 4000: the streamed weight table + workgroup-per-sample kernel are what the device picks there) and the lifted-size set,
    all from the unmodified reference.  Per-sample end points are stored for every case (C3: 10 000 x 8)."""
    from cornell_moe_amd.workloads import make_workload
    blob = {}
    for tag, kw in R3_CASES:
        w = make_workload(**kw)
        gp = ref.RefGP(1, w.alpha, w.lengths, w.X, w.y, w.noise, list(w.derivs))
        best = float(gp.additional_mean(w.discrete).min())
        Xp = w.Xp if w.p else None
        r = gp.kg(w.inner_gd, w.bounds, w.discrete, w.Xq, Xp, w.M, best, w.kg_normals, want_grad=True, details=True)
        blob[tag + "_best_so_far"] = np.array(best)
        blob[tag + "_kg"], blob[tag + "_grad_kg"], blob[tag + "_best_point"] = np.array(r["kg"]), r["grad"], r["best_point"]
        blob[tag + "_check"] = np.array([float(w.X.sum()), float(w.kg_normals.sum()), float(w.Xq.sum()), float(w.discrete.sum())])
        blob[tag + "_seconds"] = np.array(r["seconds"])
        if tag != "c3full":
            rv = gp.kg(w.inner_gd, w.bounds, w.discrete, w.Xq, Xp, w.M, best, w.kg_normals, want_grad=False)
            blob[tag + "_kg_value_only"] = np.array(rv["kg"])
            pts = w.query[:3]
            blob[tag + "_q_mean"], blob[tag + "_q_grad_mean"], blob[tag + "_q_var"] = gp.mean(pts), gp.grad_mean(pts), gp.var(pts)
        print("%s: n=%d d=%d q=%d p=%d g=%d m=%d M=%d  KG=%.15g  (reference: %.2f s state + %.2f s evaluation)" % (
            tag, w.n, w.d, w.q, w.p, w.g, w.m, w.M, r["kg"], r["seconds"][0], r["seconds"][1]), flush=True)
    np.savez_compressed(OUT_SHAPES3, **blob)
    print("wrote", OUT_SHAPES3, os.path.getsize(OUT_SHAPES3), "bytes")


OUT_SHAPES4 = os.path.join(ROOT, "tests", "golden", "ref_shapes_r4.npz")


def shape_fixtures_r4():
    """Round 4: BASELINE.json configs[4] EXACTLY (n = 2000, d = 12, q = 8, g = 3, P = 50; M = 64 samples) and
    the stretch point g = 12 (m = 104) at n = 600 (N = 7800), from the unmodified reference: KG, grad KG, every end point, and
    the value-only evaluation."""
    from cornell_moe_amd.workloads import make_workload, R4_PARITY_CASES
    blob = {}
    for tag, kw in R4_PARITY_CASES:
        w = make_workload(**kw)
        gp = ref.RefGP(1, w.alpha, w.lengths, w.X, w.y, w.noise, list(w.derivs))
        best = float(gp.additional_mean(w.discrete).min())
        r = gp.kg(w.inner_gd, w.bounds, w.discrete, w.Xq, None, w.M, best, w.kg_normals, want_grad=True, details=True)
        blob[tag + "_best_so_far"] = np.array(best)
        blob[tag + "_kg"], blob[tag + "_grad_kg"], blob[tag + "_best_point"] = np.array(r["kg"]), r["grad"], r["best_point"]
        blob[tag + "_check"] = np.array([float(w.X.sum()), float(w.kg_normals.sum()), float(w.Xq.sum()), float(w.discrete.sum())])
        blob[tag + "_seconds"] = np.array(r["seconds"])
        rv = gp.kg(w.inner_gd, w.bounds, w.discrete, w.Xq, None, w.M, best, w.kg_normals, want_grad=False)
        blob[tag + "_kg_value_only"] = np.array(rv["kg"])
        pts = w.query[:3]
        blob[tag + "_q_mean"], blob[tag + "_q_grad_mean"], blob[tag + "_q_var"] = gp.mean(pts), gp.grad_mean(pts), gp.var(pts)
        print("%s: n=%d d=%d q=%d g=%d m=%d M=%d  KG=%.15g  (reference: %.2f s state + %.2f s evaluation)" % (
            tag, w.n, w.d, w.q, w.g, w.m, w.M, r["kg"], r["seconds"][0], r["seconds"][1]), flush=True)
    np.savez_compressed(OUT_SHAPES4, **blob)
    print("wrote", OUT_SHAPES4, os.path.getsize(OUT_SHAPES4), "bytes")


OUT_SHAPES6 = os.path.join(ROOT, "tests", "golden", "ref_shapes_r6.npz")
BUILD16K = dict(name="C5", n=4096, M=2)   # N = 16 384 training entries (d = 12, g = 3): the build beyond the fused-step schedule's limit


def shape_fixtures_r6():
    """Round 6, from the unmodified reference:
    `c5m512` -- BASELINE.json configs[4] EXACTLY at M = 512 samples (N = 8000, 32 tiles): KG, grad KG, all 512 end points, the
    value-only evaluation (the r4 fixture has M = 64);
    `build16k` -- the GP BUILD at N = 16 384 (n = 4096, d = 12, g = 3): the reference's own factor (diagonal, three rows) and
    K^-1 (y - mean) in full, plus the posterior at three points -- the N^3 / 3 of its scalar Cholesky, no Monte Carlo."""
    from cornell_moe_amd.workloads import make_workload
    blob = {}
    w = make_workload("C5", M=512)
    gp = ref.RefGP(1, w.alpha, w.lengths, w.X, w.y, w.noise, list(w.derivs))
    best = float(gp.additional_mean(w.discrete).min())
    r = gp.kg(w.inner_gd, w.bounds, w.discrete, w.Xq, None, w.M, best, w.kg_normals, want_grad=True, details=True)
    tag = "c5m512"
    blob[tag + "_best_so_far"] = np.array(best)
    blob[tag + "_kg"], blob[tag + "_grad_kg"], blob[tag + "_best_point"] = np.array(r["kg"]), r["grad"], r["best_point"]
    blob[tag + "_check"] = np.array([float(w.X.sum()), float(w.kg_normals.sum()), float(w.Xq.sum()), float(w.discrete.sum())])
    blob[tag + "_seconds"] = np.array(r["seconds"])
    rv = gp.kg(w.inner_gd, w.bounds, w.discrete, w.Xq, None, w.M, best, w.kg_normals, want_grad=False)
    blob[tag + "_kg_value_only"] = np.array(rv["kg"])
    print("%s: n=%d d=%d q=%d g=%d m=%d M=%d  KG=%.15g  (reference: %.2f s state + %.2f s evaluation)" % (
        tag, w.n, w.d, w.q, w.g, w.m, w.M, r["kg"], r["seconds"][0], r["seconds"][1]), flush=True)
    del gp
    np.savez_compressed(OUT_SHAPES6, **blob)   # (kept even if the big build below is interrupted)
    import time
    w = make_workload(**BUILD16K)
    t0 = time.time()
    gp = ref.RefGP(1, w.alpha, w.lengths, w.X, w.y, w.noise, list(w.derivs))
    secs = time.time() - t0
    N = gp.N
    K, kiy, mean = gp.dump()
    tag = "build16k"
    rows = np.array([N - 1, N // 2, 4097])
    blob[tag + "_check"] = np.array([float(w.X.sum()), float(w.y.sum()), float(N)])
    blob[tag + "_chol_diag"] = np.diag(K).copy()
    blob[tag + "_chol_rows_idx"] = rows
    blob[tag + "_chol_rows"] = np.stack([np.tril(K)[r_] for r_ in rows])
    blob[tag + "_K_inv_y"], blob[tag + "_mean"] = kiy, np.array(mean)
    blob[tag + "_seconds"] = np.array(secs)
    del K
    pts = w.query[:3]
    blob[tag + "_q_mean"], blob[tag + "_q_grad_mean"], blob[tag + "_q_var"] = gp.mean(pts), gp.grad_mean(pts), gp.var(pts)
    print("%s: N=%d built by the reference in %.1f s; log det K = %.15g" % (tag, N, secs, 2.0 * np.log(blob[tag + "_chol_diag"]).sum()), flush=True)
    np.savez_compressed(OUT_SHAPES6, **blob)
    print("wrote", OUT_SHAPES6, os.path.getsize(OUT_SHAPES6), "bytes")


OUT_BUILD26K = os.path.join(ROOT, "tests", "golden", "ref_build26k.npz")


def build26k_fixture():
    """Round 6: the GP build at the SURVEY 8(d) stretch point itself -- n = 2000, d = 12, all 12 derivatives observed: N = 26 000, K =
    5.4 GB -- from the unmodified reference's scalar Cholesky (~40 minutes on one core, ~20 GB of host memory): the factor's diagonal
    and three rows, K^-1 (y - mean) in full, the posterior at three points.  Build only: no Monte Carlo."""
    import time
    from cornell_moe_amd.workloads import make_workload
    w = make_workload("C5", derivs=tuple(range(12)), M=2)
    t0 = time.time()
    gp = ref.RefGP(1, w.alpha, w.lengths, w.X, w.y, w.noise, list(w.derivs))
    secs = time.time() - t0
    N = gp.N
    print("reference build at N = %d: %.1f s" % (N, secs), flush=True)
    K, kiy, mean = gp.dump()
    blob = {}
    rows = np.array([N - 1, N // 2, 4097])
    blob["check"] = np.array([float(w.X.sum()), float(w.y.sum()), float(N)])
    blob["chol_diag"] = np.diag(K).copy()
    blob["chol_rows_idx"] = rows
    blob["chol_rows"] = np.stack([np.where(np.arange(N) <= r_, K[r_], 0.0) for r_ in rows])
    blob["K_inv_y"], blob["mean"], blob["seconds"] = kiy, np.array(mean), np.array(secs)
    del K
    pts = w.query[:3]
    blob["q_mean"], blob["q_grad_mean"], blob["q_var"] = gp.mean(pts), gp.grad_mean(pts), gp.var(pts)
    np.savez_compressed(OUT_BUILD26K, **blob)
    print("wrote", OUT_BUILD26K, os.path.getsize(OUT_BUILD26K), "bytes; log det K = %.15g" % (2.0 * np.log(blob["chol_diag"]).sum()))


OUT_SIMPLEX = os.path.join(ROOT, "tests", "golden", "ref_simplex.npz")


def simplex_fixtures():
    """Round 4: the EI outer optimiser over SimplexIntersectTensorProductDomain (gpp_domain.hpp:215-349) from the unmodified reference --
    ComputeOptimalPointsToSampleViaMultistartGradientDescent<SimplexIntersectTensorProductDomain> at q = 1, p = 0 (analytic EI), 24 starts
    inside the simplex, data whose good region lies beyond the diagonal face so that the simplex limit of LimitUpdate (half the distance to
    the face along the tensor-limited direction) is what shapes the ascent; max_relative_change = 1.0 in one case (the epsilon tweak)."""
    blob, k = {}, 0
    for (seed, n, d, box, outer) in ((7101, 40, 3, (0.0, 1.0), (24, 40, 3, 4, 0.6, 0.5, 1.0, 1e-9)),
                                     (7102, 30, 2, (0.0, 1.0), (24, 30, 2, 4, 0.7, 0.3, 0.5, 1e-9)),
                                     (7103, 50, 4, (-0.2, 0.8), (24, 25, 2, 4, 0.7, 0.4, 0.8, 1e-9))):
        rng = np.random.default_rng(seed)
        c = dict(n=n, d=d, rng_seed=seed)
        c["X"] = rng.uniform(0.0, 0.6, size=(n, d))
        c["y"] = (-np.sum(c["X"], axis=1) + 0.2 * np.sin(5 * c["X"]).sum(1) + 0.05 * rng.uniform(size=n))[:, None]  # lower = better beyond the face
        c["alpha"], c["lengths"], c["noise"] = 1.0, rng.uniform(0.3, 0.6, size=d), np.array([0.01])
        c["bounds"] = np.tile(np.array(box), d)
        starts = []
        while len(starts) < 24:
            x = rng.uniform(max(box[0], 0.0), min(box[1], 1.0), size=d)
            if x.sum() <= 0.95:
                starts.append(x)
        c["starts"] = np.array(starts)
        c["outer_gd"] = np.array(outer)
        c["best_so_far"] = float(c["y"].min())
        gp = ref.RefGP(1, c["alpha"], c["lengths"], c["X"], c["y"], c["noise"], [])
        best, found = gp.ei_multistart_analytic(c["outer_gd"], c["bounds"], c["starts"], c["best_so_far"], domain_type=1)
        best_tp, found_tp = gp.ei_multistart_analytic(c["outer_gd"], c["bounds"], c["starts"], c["best_so_far"], domain_type=0)
        print("simplex EI multistart case %d: d=%d best=%s (sum %.6f) found=%d; tensor-product best=%s (sum %.4f)" % (
            k, d, np.array2string(best, precision=6), best.sum(), found, np.array2string(best_tp, precision=4), best_tp.sum()))
        for key, val in c.items():
            blob["s%d_in_%s" % (k, key)] = np.asarray(val)
        blob["s%d_out_best_point" % k], blob["s%d_out_found" % k] = best, np.array(int(found))
        blob["s%d_out_best_point_tensor" % k] = best_tp
        k += 1
    blob["num"] = np.array(k)
    np.savez_compressed(OUT_SIMPLEX, **blob)
    print("wrote", OUT_SIMPLEX, os.path.getsize(OUT_SIMPLEX), "bytes")


OUT_SIMPLEX_KG = os.path.join(ROOT, "tests", "golden", "ref_simplex_kg.npz")


def simplex_kg_fixtures():
    """Round 4: KG with the INNER optimisations over SimplexIntersectTensorProductDomain, from the unmodified reference
    (KnowledgeGradientEvaluator<SimplexIntersectTensorProductDomain>, the instantiation behind DomainTypes::kSimplex in
    gpp_python_knowledge_gradient.cpp:288-296), and its multistart driver with outer AND inner simplex domain.  Data whose posterior
    mean falls towards the diagonal face, discretised points inside the simplex, inner optimisers that take enough steps to reach the
    face; one case with max_relative_change = 1 (the epsilon tweak), one with a fidelity coordinate (the simplex then spans the free
    coordinates only), one with an observed derivative (table rows permuted: the update must still walk the coordinates in order)."""
    blob, k = {}, 0
    for (seed, n, d, q, p, P, M, derivs, f, box, inner) in (
            (8101, 40, 3, 2, 0, 6, 40, (), 0, (0.0, 1.0), (1, 20, 2, 3, 0.0, 1.0, 0.5, 1e-10)),
            (8102, 36, 2, 2, 1, 5, 32, (), 0, (0.0, 1.0), (1, 15, 2, 3, 0.0, 2.0, 1.0, 1e-10)),
            (8103, 40, 4, 2, 0, 6, 32, (), 1, (-0.1, 0.9), (1, 12, 2, 3, 0.0, 1.0, 0.3, 1e-10)),
            (8104, 14, 3, 2, 0, 6, 32, (1,), 0, (0.0, 1.0), (1, 12, 1, 3, 0.0, 1.0, 0.4, 1e-10))):
        rng = np.random.default_rng(seed)
        g, size = len(derivs), d - f
        c = dict(n=n, d=d, q=q, p=p, P=P, M=M, derivs=np.array(derivs, dtype=np.int32), num_fidelity=f, rng_seed=seed)

        def in_simplex(count, hi):
            pts = []
            while len(pts) < count:
                x = rng.uniform(max(box[0], 0.0) + 0.01, min(box[1], 1.0), size=size)
                if x.sum() <= hi:
                    pts.append(x)
            return np.array(pts)
        Xs = in_simplex(n, 0.9)
        c["X"] = np.hstack([Xs, rng.uniform(0.3, 1.0, size=(n, f))]) if f else Xs
        c["y"] = np.zeros((n, 1 + g))
        c["y"][:, 0] = -1.5 * Xs.sum(1) + 0.2 * np.sin(5 * Xs).sum(1) + 0.05 * rng.uniform(size=n)   # lower = better beyond the face
        for a, dd in enumerate(derivs):
            c["y"][:, 1 + a] = -1.5 + np.cos(5 * c["X"][:, dd])
        c["alpha"], c["lengths"], c["noise"] = 1.0, rng.uniform(0.3, 0.6, size=d), np.full(1 + g, 0.01)
        c["bounds"] = np.tile(np.array(box), d)
        if f:
            c["bounds"][2 * size:] = np.tile([0.3, 1.0], f)
        xq = in_simplex(q, 0.8)
        c["Xq"] = np.hstack([xq, rng.uniform(0.4, 0.9, size=(q, f))]) if f else xq
        xp = in_simplex(p, 0.8) if p else np.zeros((0, size))
        c["Xp"] = (np.hstack([xp, rng.uniform(0.4, 0.9, size=(p, f))]) if f else xp) if p else np.zeros((0, d))
        c["discrete"] = in_simplex(P, 0.85)
        c["inner_gd"] = np.array(inner)
        gp = ref.RefGP(1, c["alpha"], c["lengths"], c["X"], c["y"], c["noise"], list(derivs))
        disc_full = np.hstack([c["discrete"], np.ones((P, f))]) if f else c["discrete"]
        c["best_so_far"] = float(gp.additional_mean(disc_full).min())
        m = (q + p) * (1 + g)
        c["normals"] = rng.standard_normal((((M + 1) // 2), m))
        Xp = c["Xp"] if p else None
        r = gp.kg(c["inner_gd"], c["bounds"][: 2 * size], c["discrete"], c["Xq"], Xp, M, c["best_so_far"], c["normals"], num_fidelity=f,
                  domain_type=1)
        rt = gp.kg(c["inner_gd"], c["bounds"][: 2 * size], c["discrete"], c["Xq"], Xp, M, c["best_so_far"], c["normals"], num_fidelity=f,
                   domain_type=0)
        sums = r["best_point"][:, :size].sum(1)
        print("simplex KG case %d: d=%d f=%d g=%d q=%d p=%d  KG=%.12g (tensor product: %.12g)  end points with sum > 0.99: %d of %d, max sum %.12f; "
              "end points that differ from the tensor-product run: %d" % (k, d, f, g, q, p, r["kg"], rt["kg"], int((sums > 0.99).sum()), M,
                                                                         sums.max(), int((np.abs(r["best_point"] - rt["best_point"]).max(1) > 1e-9).sum())))
        for key, val in c.items():
            blob["e%d_in_%s" % (k, key)] = np.asarray(val)
        blob["e%d_out_kg" % k], blob["e%d_out_grad" % k], blob["e%d_out_best_point" % k] = np.array(r["kg"]), r["grad"], r["best_point"]
        blob["e%d_out_kg_tensor" % k] = np.array(rt["kg"])
        k += 1
    blob["num_eval"] = np.array(k)
    # the multistart driver, outer and inner domain simplex
    k = 0
    inner = np.array((1, 8, 1, 3, 0.0, 1.0, 0.3, 1e-10))
    for (seed, n, d, q, P, M, outer) in ((8201, 40, 3, 1, 6, 32, (24, 6, 2, 4, 0.7, 0.3, 0.5, 1e-7)),
                                         (8202, 36, 2, 2, 5, 32, (24, 5, 2, 4, 0.7, 0.2, 1.0, 1e-7))):
        rng = np.random.default_rng(seed)
        c = dict(n=n, d=d, q=q, p=0, P=P, M=M, derivs=np.array((), dtype=np.int32), num_fidelity=0, rng_seed=seed % 1000)

        def in_simplex(count, hi):
            pts = []
            while len(pts) < count:
                x = rng.uniform(0.01, 1.0, size=d)
                if x.sum() <= hi:
                    pts.append(x)
            return np.array(pts)
        c["X"] = in_simplex(n, 0.9)
        c["y"] = (-1.5 * c["X"].sum(1) + 0.2 * np.sin(5 * c["X"]).sum(1) + 0.05 * rng.uniform(size=n))[:, None]
        c["alpha"], c["lengths"], c["noise"] = 1.0, rng.uniform(0.3, 0.6, size=d), np.array([0.01])
        c["bounds"] = np.tile([0.0, 1.0], d)
        c["discrete"] = in_simplex(P, 0.85)
        c["starts"] = in_simplex(24 * q, 0.9).reshape(24, q, d)
        c["outer_gd"], c["inner_gd"] = np.array(outer), inner
        gp = ref.RefGP(1, c["alpha"], c["lengths"], c["X"], c["y"], c["noise"], [])
        c["best_so_far"] = float(gp.additional_mean(c["discrete"]).min())
        c["normals"] = ref.normal_draws(c["rng_seed"], ((M + 1) // 2) * q).reshape(-1, q)
        best, found = gp.kg_multistart(c["outer_gd"], inner, c["bounds"], c["discrete"], c["starts"], None, M, c["best_so_far"], c["rng_seed"],
                                       domain_type=1)
        best_tp, _ = gp.kg_multistart(c["outer_gd"], inner, c["bounds"], c["discrete"], c["starts"], None, M, c["best_so_far"], c["rng_seed"],
                                      domain_type=0)
        print("simplex KG multistart case %d: d=%d q=%d best=%s (sums %s) found=%d; tensor-product best=%s" % (
            k, d, q, np.array2string(best, precision=6), np.array2string(best.sum(1), precision=6), found, np.array2string(best_tp, precision=4)))
        for key, val in c.items():
            blob["m%d_in_%s" % (k, key)] = np.asarray(val)
        blob["m%d_out_best_point" % k], blob["m%d_out_found" % k] = best, np.array(int(found))
        blob["m%d_out_best_point_tensor" % k] = best_tp
        k += 1
    blob["num_ms"] = np.array(k)
    np.savez_compressed(OUT_SIMPLEX_KG, **blob)
    print("wrote", OUT_SIMPLEX_KG, os.path.getsize(OUT_SIMPLEX_KG), "bytes")


OUT_MS = os.path.join(ROOT, "tests", "golden", "ref_kg_multistart.npz")


def kg_multistart_fixtures():
    """The KG OUTER optimiser from the unmodified reference:
    ComputeKGOptimalPointsToSampleViaMultistartGradientDescent (gpp_knowledge_gradient_optimization.hpp:860-935) and
    ComputeKGMCMCOptimalPointsToSampleViaMultistartGradientDescent (gpp_knowledge_gradient_mcmc_optimization.hpp:665-760),
    one thread, 24 starts, NormalRNG(seed) -- whose stream is stored as the explicit table the device driver replays."""
    blob, k = {}, 0
    inner = np.array((1, 6, 1, 3, 0.0, 1.0, 0.1, 1e-10))
    for (seed, n, d, q, p, P, M, derivs, f, outer) in (
            (5101, 40, 3, 2, 0, 8, 64, (), 0, (24, 6, 2, 4, 0.7, 0.1, 0.2, 1e-7)),
            (5102, 50, 4, 2, 1, 8, 48, (), 0, (24, 5, 2, 4, 0.7, 0.08, 0.2, 1e-7)),
            (5103, 10, 3, 2, 0, 6, 32, (0, 2), 0, (24, 5, 1, 4, 0.7, 0.1, 0.2, 1e-7)),
            (5104, 36, 4, 1, 1, 6, 40, (), 1, (24, 6, 2, 4, 0.7, 0.1, 0.2, 1e-7))):
        rng = np.random.default_rng(seed)
        g = len(derivs)
        c = dict(n=n, d=d, q=q, p=p, P=P, M=M, derivs=np.array(derivs, dtype=np.int32), num_fidelity=f, rng_seed=seed % 1000)
        c["X"] = rng.uniform(0.05, 0.95, size=(n, d))
        c["y"] = np.zeros((n, 1 + g))
        c["y"][:, 0] = np.sin(3 * c["X"]).sum(1) + 0.1 * rng.uniform(size=n)
        for a, dd in enumerate(derivs):
            c["y"][:, 1 + a] = 3 * np.cos(3 * c["X"][:, dd])
        c["alpha"] = 1.1
        c["lengths"] = rng.uniform(0.5, 0.9, size=d)
        c["noise"] = np.full(1 + g, 0.02)
        c["bounds"] = np.tile([0.0, 1.0], d)
        c["Xp"] = rng.uniform(0.1, 0.9, size=(p, d))
        c["discrete"] = rng.uniform(0.0, 1.0, size=(P, d - f))
        c["starts"] = rng.uniform(0.05, 0.95, size=(24, q, d))
        c["outer_gd"], c["inner_gd"] = np.array(outer), inner
        gp = ref.RefGP(1, c["alpha"], c["lengths"], c["X"], c["y"], c["noise"], list(derivs))
        disc_full = np.hstack([c["discrete"], np.ones((P, f))]) if f else c["discrete"]
        c["best_so_far"] = float(gp.additional_mean(disc_full).min())
        m = (q + p) * (1 + g)
        c["normals"] = ref.normal_draws(c["rng_seed"], ((M + 1) // 2) * m).reshape(-1, m)
        Xp = c["Xp"] if p else None
        best, found = gp.kg_multistart(c["outer_gd"], inner, c["bounds"], c["discrete"], c["starts"], Xp, M, c["best_so_far"],
                                       c["rng_seed"], num_fidelity=f)
        # KG of the returned point on a FRESH state through the table route (for information; the driver's own value is the
        # frozen-head one, see oracle/moe_oracle.c: orc_kg_head)
        kb = gp.kg(inner, c["bounds"][: 2 * (d - f)], c["discrete"], best, Xp, M, c["best_so_far"], c["normals"], want_grad=False,
                   num_fidelity=f)["kg"]
        # the seeded route of the single-evaluation entry points: KG at the first three starts from NormalRNG(seed) itself
        ks = [gp.kg_seeded(inner, c["bounds"][: 2 * (d - f)], c["discrete"], c["starts"][i], Xp, M, c["best_so_far"], c["rng_seed"],
                           num_fidelity=f) for i in range(3)]
        o = dict(best_point=best, found=np.array(int(found)), best_kg_fresh=np.array(kb), seeded_kg=np.array(ks))
        print("kg multistart case %d: n=%d d=%d q=%d p=%d g=%d f=%d  KG(best point, fresh state)=%.12g found=%d" % (
            k, n, d, q, p, g, f, kb, found))
        for key, val in c.items():
            blob["k%d_in_%s" % (k, key)] = np.asarray(val)
        for key, val in o.items():
            blob["k%d_out_%s" % (k, key)] = np.asarray(val)
        k += 1
    blob["num"] = np.array(k)
    # the reference build's NormalRNG stream itself (pins moe_normal_draws)
    blob["stream_seeds"] = np.array([0, 7, 314, 2718, 123456789])
    blob["stream_draws"] = np.array([ref.normal_draws(int(sd), 2001) for sd in blob["stream_seeds"]])
    # MCMC twin: without and with a fidelity dimension (cost and its gradient live), q = 2
    for mi, (seed, n, d, q, p, P, M, nm, f, outer) in enumerate((
            (6201, 30, 3, 2, 0, 6, 32, 3, 0, (24, 5, 2, 4, 0.7, 0.1, 0.2, 1e-7)),
            (6202, 32, 4, 2, 1, 6, 32, 2, 1, (24, 4, 2, 4, 0.7, 0.05, 0.2, 1e-7)))):
        rng = np.random.default_rng(seed)
        c = dict(n=n, d=d, q=q, p=p, P=P, M=M, num_mcmc=nm, derivs=np.array((), dtype=np.int32), num_fidelity=f,
                 rng_seed=77 + mi)
        c["X"] = rng.uniform(0.05, 0.95, size=(n, d))
        c["y"] = (np.sin(3 * c["X"]).sum(1) + 0.1 * rng.uniform(size=n))[:, None]
        c["hypers"] = np.c_[rng.uniform(0.8, 1.5, nm), rng.uniform(0.5, 0.9, size=(nm, d))]
        c["noises"] = rng.uniform(0.01, 0.05, size=(nm, 1))
        c["bounds"] = np.tile([0.0, 1.0], d)
        if f:
            c["bounds"][2 * (d - f):] = np.tile([0.2, 1.0], f)  # fidelity coordinates stay away from 0 (the cost divides by them)
        c["Xp"] = rng.uniform(0.25, 0.9, size=(p, d))
        c["discrete"] = rng.uniform(0.0, 1.0, size=(nm, P, d - f))
        c["starts"] = rng.uniform(0.25, 0.9, size=(24, q, d))
        c["outer_gd"], c["inner_gd"] = np.array(outer), inner
        R = ref.RefGPMCMC(c["hypers"], c["noises"], c["X"], c["y"], ())
        bests = []
        for i in range(nm):
            gi = ref.RefGP(1, c["hypers"][i, 0], c["hypers"][i, 1:], c["X"], c["y"], c["noises"][i], [])
            dfull = np.hstack([c["discrete"][i], np.ones((P, f))]) if f else c["discrete"][i]
            bests.append(float(gi.additional_mean(dfull).min()))
        c["best_so_far"] = np.array(bests)
        m = q + p
        c["normals"] = ref.normal_draws(c["rng_seed"], ((M + 1) // 2) * m).reshape(-1, m)
        Xp = c["Xp"] if p else None
        best, found = R.kg_multistart(c["outer_gd"], inner, c["bounds"], c["discrete"], c["starts"], Xp, M, c["best_so_far"],
                                      c["rng_seed"], num_fidelity=f)
        print("kg mcmc multistart %d: q=%d p=%d f=%d found=%d best=%s" % (mi, q, p, f, found, np.round(best.ravel(), 4)))
        for key, val in c.items():
            blob["mk%d_in_%s" % (mi, key)] = np.asarray(val)
        blob["mk%d_out_best_point" % mi], blob["mk%d_out_found" % mi] = best, np.array(int(found))
    blob["num_mcmc"] = np.array(2)
    np.savez_compressed(OUT_MS, **blob)
    print("wrote", OUT_MS, os.path.getsize(OUT_MS), "bytes")


def ll_multistart_fixtures():
    """r5: the maximum-likelihood hyper-parameter optimisers (SURVEY 8b 'next'; gpp_model_selection.hpp:967-1103) from explicit
    linear-space initial guesses (oracle/ref_harness.cpp: ref_ll_multistart -- the body of
    MultistartGradientDescentHyperparameterOptimization minus its Latin-hypercube draw) -> tests/golden/ref_ll_multistart.npz.
    Cases 0-2 use contractive steps (pre_mult 1e-3 / 2e-4: a relative perturbation of 1e-12 of the gradient moves the end point by
    <= 3e-11) so that the end point can be pinned; case 3 has the GD settings of the reference's own tests
    (gpp_model_selection_test.cpp:700-710, 892-900: gamma 0.5, pre_mult 0.5, max_relative_change 0.02), under which EVERY step is cut to
    2 % of the distance to the nearest wall and only the SIGN of each gradient component matters -- near an optimum that sign flips on
    rounding, and the same 1e-12 perturbation moves the end point by 1.6 % (measured here): pinned to its likelihood only."""
    out = {}
    k = 0
    for seed, n, d, derivs, S, steps, restarts, gamma, pre, mrc in ((11, 30, 2, (), 6, 150, 2, 0.5, 1.0e-3, 0.1),
                                                                    (12, 45, 3, (), 1, 100, 3, 0.5, 2.0e-4, 0.2),
                                                                    (13, 24, 2, (1,), 4, 120, 2, 0.5, 1.0e-3, 0.1),
                                                                    (11, 30, 2, (), 6, 60, 3, 0.5, 0.5, 0.02)):
        rng = np.random.default_rng(seed)
        g = len(derivs)
        X = rng.uniform(0.0, 1.0, size=(n, d))
        true_len = rng.uniform(0.3, 0.6, size=d)
        # a smooth function + a little noise; derivative observations from finite formulas of the same function
        f = lambda x: np.sin(x / true_len).sum(axis=-1)
        y = np.zeros((n, 1 + g))
        y[:, 0] = f(X) + 0.05 * rng.standard_normal(n)
        for a, dd in enumerate(derivs):
            y[:, 1 + a] = np.cos(X[:, dd] / true_len[dd]) / true_len[dd] + 0.05 * rng.standard_normal(n)
        nh = 1 + d + 1 + g
        domain_log10 = np.tile([-2.0, 1.0], (nh, 1))
        domain_log10[1 + d:, :] = [-3.0, 0.0]          # noise variances in [1e-3, 1]
        guesses = 10.0 ** (domain_log10[:, 0] + (domain_log10[:, 1] - domain_log10[:, 0]) * rng.uniform(0.25, 0.75, size=(S, nh)))
        gd = np.array((S, steps, restarts, 0, gamma, pre, mrc, 1.0e-7))
        hyper0 = guesses[0]
        best, val, found = ref.ll_multistart(1, hyper0[0], hyper0[1:1 + d], X, y, hyper0[1 + d:], derivs, gd, domain_log10, guesses)
        v0 = [ref.log_likelihood(1, h[0], h[1:1 + d], X, y, h[1 + d:], derivs) for h in guesses]
        print("ll multistart case %d: n=%d d=%d g=%d S=%d  best LL %.10g (best initial %.10g) found=%s" % (k, n, d, g, S, val, max(v0), found))
        for key, v in (("X", X), ("y", y), ("derivs", np.array(derivs, dtype=np.int64)), ("gd", gd), ("domain_log10", domain_log10),
                       ("guesses", guesses), ("best", best), ("best_value", np.array(val)), ("found", np.array(int(found))),
                       ("initial_values", np.array(v0)), ("contractive", np.array(int(pre < 0.01)))):
            out["s%d_%s" % (k, key)] = v
        k += 1
    out["num"] = np.array(k)
    path = os.path.join(os.path.dirname(OUT), "ref_ll_multistart.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


def main():
    if "--ll-multistart" in sys.argv:
        ll_multistart_fixtures()
        return
    if "--ll-grad" in sys.argv:
        ll_grad_fixtures()
        return
    if "--shapes" in sys.argv:
        shape_fixtures()
        return
    if "--shapes-r3" in sys.argv:
        shape_fixtures_r3()
        return
    if "--shapes-r4" in sys.argv:
        shape_fixtures_r4()
        return
    if "--shapes-r6" in sys.argv:
        shape_fixtures_r6()
        return
    if "--build26k" in sys.argv:
        build26k_fixture()
        return
    if "--simplex-kg" in sys.argv:
        simplex_kg_fixtures()
        return
    if "--simplex" in sys.argv:
        simplex_fixtures()
        return
    if "--kg-multistart" in sys.argv:
        kg_multistart_fixtures()
        return
    ll_grad_fixtures()
    shape_fixtures()
    shape_fixtures_r3()
    kg_multistart_fixtures()
    cases = []
    inner_test = (1, 100, 10, 3, 0.0, 1.0, 0.1, 1e-10)   # inner GD of the reference's KG ping test (100 steps, 10 restarts)
    inner_prod = (1, 6, 1, 3, 0.0, 1.0, 0.1, 1e-10)      # examples/main.py:123-130
    k = 0
    for (q, p) in [(1, 0), (2, 0), (1, 2), (3, 2)]:
        for derivs in [(), (0, 1, 2)]:
            cases.append(make_case(3141 + k, 7, 3, q, p, 5, derivs, 16, 0, 2.80723, 0.1, 7.0, (-5.0, 5.0), inner_test))
            k += 1
    for derivs in [(), (1,)]:
        cases.append(make_case(2718 + k, 7, 3, 2, 1, 5, derivs, 16, 1, 2.80723, 0.1, 7.0, (-5.0, 5.0), inner_prod))
        k += 1
    # larger synthetic cases in the unit box (SURVEY 8d style)
    cases.append(make_case(1001, 60, 4, 2, 0, 10, (), 64, 1, 1.0, 0.01, 0.0, (0.0, 1.0), inner_prod, data_scale=1.0,
                           length_range=(0.6, 0.8)))
    cases.append(make_case(1002, 80, 5, 3, 1, 10, (), 64, 0, 1.3, 0.05, 0.2, (0.0, 1.0), inner_prod, data_scale=1.0,
                           length_range=(0.5, 0.9)))
    cases.append(make_case(1003, 40, 3, 2, 1, 6, (0, 2), 32, 1, 1.0, 0.1, 0.3, (0.0, 1.0), inner_prod, data_scale=1.0,
                           length_range=(0.5, 0.9)))
    blob = {"num_cases": np.array(len(cases))}
    for i, c in enumerate(cases):
        c["ei_best"] = float(np.median(c["y"][:, 0]))
        if c["seed"] in (1001, 1002, 1003):  # EI multistart: 24 starts (>= 20: the reference pops a 20-deep queue), own rng
            rs = np.random.default_rng(int(c["seed"]) + 77)
            c["ms_starts"] = rs.uniform(0.0, 1.0, size=(24, c["d"]))
            c["ms_gd"] = np.array((24, 20, 2, 4, 0.7, 0.05, 0.2, 1e-8))  # contractive steps: FP64 round-off stays ~1e-13
        out = run_case(c)
        for key, val in c.items():
            blob["c%d_in_%s" % (i, key)] = np.asarray(val)
        for key, val in out.items():
            blob["c%d_out_%s" % (i, key)] = np.asarray(val)
        print("case %d: n=%d d=%d q=%d p=%d g=%d cov=%d  KG=%.12g  EI=%.12g" % (
            i, c["n"], c["d"], c["q"], c["p"], len(c["derivs"]), c["cov_type"], float(out["kg"]), float(out["ei"])))
    mc = mcmc_cases()
    blob["num_mcmc_cases"] = np.array(len(mc))
    for i, (c, o) in enumerate(mc):
        for key, val in c.items():
            blob["m%d_in_%s" % (i, key)] = np.asarray(val)
        for key, val in o.items():
            blob["m%d_out_%s" % (i, key)] = np.asarray(val)
    # the reference's own known-answer vectors (gpp_linear_algebra_test.cpp:237-262), restated as data
    blob["la_A"] = np.array([[81.0, 27.0, 0.0, 90.0], [27.0, 13.0, 8.0, 44.0], [0.0, 8.0, 52.0, 40.0], [90.0, 44.0, 40.0, 217.0]])
    blob["la_A_chol"] = np.array([[9.0, 0, 0, 0], [3.0, 2.0, 0, 0], [0.0, 4.0, 6.0, 0], [10.0, 7.0, 2.0, 8.0]])
    blob["la_B"] = np.array([[25.0, 15.0, -5.0], [15.0, 18.0, 0.0], [-5.0, 0.0, 11.0]])
    blob["la_B_chol"] = np.array([[5.0, 0, 0], [3.0, 3.0, 0], [-1.0, 1.0, 3.0]])
    for name in ("la_A", "la_B"):
        rc, Lr = ref.cholesky(blob[name])
        assert rc == 0 and np.array_equal(np.tril(Lr), blob[name + "_chol"]), "reference disagrees with its own golden vector"
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **blob)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
