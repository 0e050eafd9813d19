"""K(X, X) assembly probe at ONE size (for rocprofv3 --kernel-trace --stats: every cov_build_points_kernel launch of the process has
the same shape -- the GP constructor's own build and `repeat` probe launches).  usage: kxx_one.py <num_derivs: 3 -> N = 8000, 12 -> N = 26000>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cornell_moe_amd.api import DeviceGP
from cornell_moe_amd.workloads import make_workload
g = int(sys.argv[1]) if len(sys.argv) > 1 else 3
w = make_workload("C5", M=2, derivs=tuple(range(g)))
G = DeviceGP(w.hyperparameters, w.X, w.y, w.noise, w.derivs)
ms, nbytes = G.kxx_build_probe(repeat=20 if g <= 3 else 8)
print("K(X,X) build n=%d g=%d N=%d: HIP-event average %.4f ms per launch, %.0f algorithmic bytes, %.0f GB/s = %.3f of 8 TB/s"
      % (w.n, w.g, w.n * (1 + w.g), ms, nbytes, nbytes / ms / 1e6, nbytes / ms / 1e6 / 8000.0))
