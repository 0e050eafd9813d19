"""GP build wall time over N with the 128-tile GEMM kernel off / forced / chosen by its cost estimate (MOE_GEMM128 = 0 / 2 / 1): python tools/build_sweep.py [N ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cornell_moe_amd.api import DeviceGP
for N in [int(v) for v in sys.argv[1:]] or [3000, 4000, 6000, 8000, 12000, 16000, 20000]:
    rng = np.random.default_rng(5)
    d = 8
    X = rng.uniform(size=(N, d)); y = np.sin(3 * X).sum(1, keepdims=True)
    out = []
    for mode in ("0", "2", "1"):
        os.environ["MOE_GEMM128"] = mode
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); G = DeviceGP(np.r_[1.0, np.full(d, 0.5)], X, y, [0.01]); ts.append(time.perf_counter() - t0); del G
        out.append(1e3 * min(ts[1:]))
    print("N = %6d: build %8.2f ms (64-tile kernels)  %8.2f ms (128-tile wherever it is built)  %8.2f ms (chosen by the cost estimate)" % (N, out[0], out[1], out[2]), flush=True)
