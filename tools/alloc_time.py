"""What hipMalloc / hipFree of GP-sized buffers cost (N = 8000: 512 MB per matrix), and a GP's construction / destruction wall time."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hip = C.CDLL("libamdhip64.so")
p = C.c_void_p()
for mb in (64, 512):
    hip.hipMalloc(C.byref(p), C.c_size_t(mb << 20)); hip.hipFree(p)
    ta, tf = [], []
    for _ in range(5):
        t0 = time.perf_counter(); hip.hipMalloc(C.byref(p), C.c_size_t(mb << 20)); t1 = time.perf_counter(); hip.hipFree(p); t2 = time.perf_counter()
        ta.append(t1 - t0); tf.append(t2 - t1)
    print("hipMalloc %4d MB: %.3f ms   hipFree: %.3f ms" % (mb, 1e3 * min(ta), 1e3 * min(tf)))
from cornell_moe_amd.api import DeviceGP
rng = np.random.default_rng(1005)
n, d, g = 2000, 12, 3
X = rng.uniform(size=(n, d)); y = np.zeros((n, 1 + g)); y[:, 0] = np.sin(3 * X).sum(1)
tc, td = [], []
for _ in range(4):
    t0 = time.perf_counter(); G = DeviceGP(np.r_[1.0, np.full(d, 0.7)], X, y, np.full(1 + g, 0.01), tuple(range(g))); t1 = time.perf_counter()
    del G; t2 = time.perf_counter()
    tc.append(t1 - t0); td.append(t2 - t1)
print("DeviceGP N=8000: construct %.2f ms, destroy %.2f ms (best of 3 after the first)" % (1e3 * min(tc[1:]), 1e3 * min(td[1:])))
