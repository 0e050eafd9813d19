"""A loop of single q-EI value+gradient evaluations at C2 (the latency path) for a kernel trace.   python tools/ei_loop.py [reps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cornell_moe_amd.api import DeviceGP  # noqa: E402
from cornell_moe_amd.workloads import make_workload  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
w = make_workload("C2")
G = DeviceGP(w.hyperparameters, w.X, w.y, w.noise, ())
best = float(np.min(w.y[:, 0])) + 0.5
for want_grad in (False, True):
    G.ei(w.Xq, None, w.M, best, w.ei_normals, want_grad=want_grad)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        G.ei(w.Xq, None, w.M, best, w.ei_normals, want_grad=want_grad)
        ts.append(time.perf_counter() - t0)
    print("C2 q-EI %s: median %.1f us, min %.1f us" % ("value+grad" if want_grad else "value     ", 1e6 * float(np.median(ts)), 1e6 * min(ts)))
