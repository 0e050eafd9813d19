"""Build libmoe_hip.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc, in tree.

    python -m cornell_moe_amd.build            # incremental
    python -m cornell_moe_amd.build --force

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with gpurun snapshots.
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "lib", "libmoe_hip.so")
SOURCES = ["kernels_cov.hip", "kernels_linalg.hip", "host_math.hip", "gp.hip", "kg.hip", "kg_state.hip", "kg_mc_dp4.hip", "kg_mc_dp8.hip",
           "kg_mc_dp12.hip", "kg_mc_dp16.hip", "kg_mc_dp4b.hip", "kg_mc_dp8b.hip", "kg_mc_dp12b.hip", "kg_mc_dp16b.hip", "kg_mc_dp24.hip", "kg_mc_dp32.hip", "multistart.hip", "mcmc.hip", "ei.hip", "api.hip", "rccl_comm.hip", "query_grad.hip"]
HEADERS = ["common.hpp", "launch.hpp", "kernels.hpp", "gemm128.hpp", "device_cov.hpp", "fastmath.hpp", "host_math.hpp", "gp.hpp", "kg.hpp", "kg_mc.hpp", "kg_mc_lane.hpp", "kg_state.hpp",
           os.path.join("..", "..", "include", "moe_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         # MFMA accumulators stay in VGPRs: left to its heuristics the compiler parks them in AGPRs and copies all of them
         # out and back around every MFMA group (64 v_accvgpr moves per 16 MFMAs in mfma_gemm_kernel, each waiting for
         # its MFMA to retire)
         "-mllvm", "-amdgpu-mfma-vgpr-form"]
FLAGS += os.environ.get("MOE_EXTRA_FLAGS", "").split()  # e.g. -DMOE_BLOCK_PROF=1 (tools; rebuild with force)


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _newest_header():
    return max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)


def _compile(src, force):
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    srcp = os.path.join(CSRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(srcp), _newest_header()):
        return obj, False
    cmd = [_hipcc()] + FLAGS + ["-c", srcp, "-o", obj]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s" % (src, res.stdout))
    return obj, True


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(os.cpu_count() or 4, len(SOURCES))) as ex:
        results = list(ex.map(lambda s: _compile(s, force), SOURCES))
    objs = [r[0] for r in results]
    rebuilt = any(r[1] for r in results)
    stale = not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs)  # (an object compiled by hand)
    rebuilt = rebuilt or stale
    if rebuilt:
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
        if res.returncode != 0:
            raise RuntimeError("link failed:\n%s" % res.stdout)
    if verbose:
        print("libmoe_hip.so:", LIB, "(rebuilt)" if rebuilt else "(up to date)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
