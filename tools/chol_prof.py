"""rocprofv3 driver: two GP builds at N = 8000 (n = 2000, d = 12, g = 3)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cornell_moe_amd.api import DeviceGP
g = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rng = np.random.default_rng(1005)
n, d = 2000, 12
X = rng.uniform(size=(n, d))
y = np.zeros((n, 1 + g)); y[:, 0] = np.sin(3 * X).sum(1)
for _ in range(2):
    G = DeviceGP(np.r_[1.0, np.full(d, 0.7)], X, y, np.full(1 + g, 0.01), tuple(range(g)))
    del G
