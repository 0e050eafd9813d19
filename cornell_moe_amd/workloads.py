"""Synthetic GP / acquisition workloads: the named BASELINE.json configurations.

All FP64.  One ``numpy.random.default_rng(seed)`` per configuration; X ~ U[0,1]^d, y = sum_k sin(3 x_k) + 0.1 U[0,1]
(plus the analytic partials 3 cos(3 x_k) for observed derivative dims), Matern-5/2 with alpha = 1, l_k = 0.7,
noise 0.01, domain [0,1]^d, inner GD parameters = examples/main.py:123-130 of the reference.
"""
import numpy as np

INNER_GD = (1, 6, 1, 3, 0.0, 1.0, 0.1, 1.0e-10)  # examples/main.py:123-130

CONFIGS = {
    # name: (seed, n, d, q, M, P, derivs)
    "C1": dict(seed=1001, n=200, d=2, q=1, M=0, P=0, derivs=()),
    "C2": dict(seed=1002, n=500, d=4, q=2, M=1000, P=10, derivs=()),
    "C3": dict(seed=1003, n=1000, d=8, q=4, M=10000, P=10, derivs=()),
    "C4": dict(seed=1004, n=1000, d=8, q=4, M=10000, P=10, derivs=()),
    "C5": dict(seed=1005, n=2000, d=12, q=8, M=20000, P=50, derivs=(0, 1, 2)),
}

# Parity-test cases of round 3 (tools/make_golden.py: shape_fixtures_r3 -> tests/golden/ref_shapes_r3.npz; the other configs
# are parity-test cases, not bench lines): the EXACT headline configuration, C5's d-KG at n = 1000 (big enough for the
# workgroup-per-sample kernel + streamed weight table), and the sizes lifted in round 2 (d > 16, g > 4, m > 64).
R3_PARITY_CASES = (
    ("c3full", dict(name="C3")),                                                                   # M = 10 000
    ("c5n1000", dict(name="C5", n=1000, M=64)),                                                    # N = 4000, m = 32: 16 tiles x 4 weights
    ("d24g2", dict(seed=3301, n=40, d=24, q=2, p=0, M=16, P=8, derivs=(3, 17))),                   # padded dimension 24
    ("d32g0", dict(seed=3302, n=48, d=32, q=4, p=1, M=32, P=6, derivs=())),                        # padded dimension 32
    ("d12g12", dict(seed=3303, n=40, d=12, q=8, p=0, M=16, P=10, derivs=tuple(range(12)))),        # C5 with all 12 derivatives: m = 104
    ("m128", dict(seed=3304, n=40, d=8, q=12, p=4, M=16, P=6, derivs=(0, 1, 2, 3, 4, 5, 6))),      # m = 16 x 8 = 128 (the limit)
    ("d20g8", dict(seed=3305, n=36, d=20, q=3, p=1, M=16, P=6, derivs=(0, 2, 4, 6, 8, 10, 12, 14))),  # 8 derivative slots, d > 16
)

# Round 4 (tools/make_golden.py: shape_fixtures_r4 -> tests/golden/ref_shapes_r4.npz): BASELINE.json configs[4] at its REAL
# size (n = 2000, N = 8000, m = 32; M = 64 keeps the reference to minutes: two N^3/3 factorisations + ~0.07 s per sample),
# and C5's stretch point (all 12 derivatives observed, m = 104) at the largest n the reference finishes in a few minutes.
R4_PARITY_CASES = (
    ("c5full", dict(name="C5", M=64)),                                                             # N = 8000, 32 tiles
    ("c5g12", dict(name="C5", n=600, M=32, derivs=tuple(range(12)))),                              # N = 7800, m = 104
)


class Workload(object):
    pass


def make_workload(name=None, seed=None, n=None, d=None, q=None, M=None, P=None, derivs=None, num_restarts=1, p=0):
    """Build a synthetic workload; keyword arguments override the named configuration's entries."""
    cfg = dict(CONFIGS[name]) if name else dict(seed=0, n=0, d=0, q=1, M=0, P=0, derivs=())
    for k, v in (("seed", seed), ("n", n), ("d", d), ("q", q), ("M", M), ("P", P), ("derivs", derivs)):
        if v is not None:
            cfg[k] = v
    rng = np.random.default_rng(cfg["seed"])
    n, d, q, M, P = cfg["n"], cfg["d"], cfg["q"], cfg["M"], cfg["P"]
    derivs = tuple(int(v) for v in cfg["derivs"])
    g = len(derivs)
    w = Workload()
    w.name, w.n, w.d, w.q, w.p, w.M, w.P, w.derivs, w.g = name, n, d, q, p, M, P, derivs, g
    w.X = rng.uniform(0.0, 1.0, size=(n, d))
    y = np.sin(3.0 * w.X).sum(axis=1) + 0.1 * rng.uniform(0.0, 1.0, size=n)
    w.y = np.zeros((n, 1 + g))
    w.y[:, 0] = y
    for a, dd in enumerate(derivs):
        w.y[:, 1 + a] = 3.0 * np.cos(3.0 * w.X[:, dd])
    w.alpha = 1.0
    w.lengths = np.full(d, 0.7)
    w.hyperparameters = np.concatenate([[w.alpha], w.lengths])
    w.noise = np.full(1 + g, 0.01)
    w.bounds = np.tile(np.array([0.0, 1.0]), d)
    w.Xq = rng.uniform(0.0, 1.0, size=(q, d))
    w.Xp = rng.uniform(0.0, 1.0, size=(p, d))
    w.discrete = rng.uniform(0.0, 1.0, size=(P, d))
    w.query = rng.uniform(0.0, 1.0, size=(100, d))
    m = (q + p) * (1 + g)
    w.m = m
    # KG: antithetic table Z[ceil(M/2)][m]; EI: table Z[M][q+p] (drawn after, so KG tables do not depend on it)
    w.kg_normals = rng.standard_normal(size=((M + 1) // 2, m))
    w.ei_normals = rng.standard_normal(size=(M, q + p))
    # extra restarts (C4): independent points_to_sample sets
    w.Xq_restarts = rng.uniform(0.0, 1.0, size=(num_restarts, q, d))
    w.Xq_restarts[0] = w.Xq
    w.inner_gd = INNER_GD
    return w


def kg_normals_full(w):
    """Expand the antithetic table to the [M][m] array the device consumes: row 2j = Z[j], row 2j+1 = -Z[j]
    (gpp_knowledge_gradient_optimization.cpp:171-180)."""
    Z = w.kg_normals
    full = np.empty((2 * Z.shape[0], Z.shape[1]))
    full[0::2] = Z
    full[1::2] = -Z
    return full[: w.M]
