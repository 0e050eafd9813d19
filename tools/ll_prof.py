"""rocprofv3 driver: the log-likelihood gradient (and one value) at N = 8000 (n = 2000, d = 12, g = 3), three times each."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cornell_moe_amd import api
n, d, g = 2000, 12, 3
rng = np.random.default_rng(3)
X = rng.uniform(size=(n, d)); y = rng.normal(size=(n, 1 + g))
LL = api.LogLikelihood(X, y, tuple(range(g)))
th = np.r_[1.0, np.full(d, 0.7), np.full(1 + g, 0.05)]
what = sys.argv[1] if len(sys.argv) > 1 else "grad"
for _ in range(3):
    print(LL.grad(th)[:3] if what == "grad" else LL.evaluate(th[None, :]))
