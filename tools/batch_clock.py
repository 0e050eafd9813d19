"""Does the MC kernel of an 8-evaluation call run slower than that of a 64-evaluation call because it is short (clock ramp), or
because of its shape?  Alternates call sizes back to back and prints the kernel's ms per evaluation."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cornell_moe_amd.api import DeviceGP
from cornell_moe_amd.workloads import make_workload
w64 = make_workload("C3", num_restarts=64)
G = DeviceGP(w64.hyperparameters, w64.X, w64.y, w64.noise, w64.derivs)
best = float(G.additional_mean(w64.discrete).min())
def call(R):
    t0 = time.perf_counter()
    G.kg_batch(w64.inner_gd, w64.bounds, w64.discrete, w64.Xq_restarts[:R], None, w64.M, best, w64.kg_normals)
    dt = time.perf_counter() - t0
    return G.last_kernel_ms()["mc"], 1e3 * dt / R
for R in (64, 64, 8, 8, 8, 8, 64, 8, 16, 16, 32, 32, 8, 8):
    mc, wall = call(R)
    print("R = %2d: mc kernel %.4f ms/eval, wall %.4f ms/eval" % (R, mc, wall), flush=True)
