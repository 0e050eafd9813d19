#!/usr/bin/env python
"""bench.py -- q-KG gradient evaluations / s on BASELINE.json's headline configuration (C3: n=1000, d=8, q=4, 10k MC).

    python bench.py --gpus N --steps K --warmup W

N > 1 runs one rank per GPU over RCCL: either the driver launches this file under `python -m torch.distributed.run
--nproc-per-node N ...` (RANK / LOCAL_RANK / WORLD_SIZE in the environment), or -- when it is called plainly -- bench.py
re-launches ITSELF that way (127.0.0.1 rendezvous on a free port) and relays rank 0's JSON line.

A "step" is one pass of the hot path over one batch: every rank evaluates `--restarts` (default 8) independent q-KG
value+gradient evaluations (different points_to_sample, same GP / discrete set / normal table = the multistart axis of
ComputeKGOptimalPointsToSampleViaMultistartGradientDescent, gpp_knowledge_gradient_optimization.hpp:860-935; C4 is 64
restarts over 8 GPUs = 8 per GPU), then -- when N > 1 -- all ranks exchange their (KG, grad KG) with ONE RCCL all_gather.
Per-GPU work is fixed as N grows ("scaling": "weak").  `--shard mc` instead splits the 10k MC samples of each evaluation
across ranks with one all_reduce per evaluation (strong scaling of a single evaluation); at N > 1 the default run times that
mode too, AFTER the K timed steps of the headline measurement, and reports it as the extra object "mc_shard".
`value` = evaluations all ranks completed / max-over-ranks wall time of the K timed steps; the GP (K factor, K^-1 y) is
resident in HBM before the timed region; per-call host inputs are the q x d query points, the P discrete points and the
normal table (PCIe-inclusive by construction -- see DESIGN.md).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_TBS = 8.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6   # MI355X FP64: vector == matrix (MFMA) dense peak, 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz


def cpu_baseline(w, best, sample_mc, log, budget_s=45.0):
    """The reference CPU path (oracle/_ref, the unmodified C++) timed on this box's host cores on bounded samples of the
    same workload: (1) ONE core, one ComputeGradKnowledgeGradient call with >= 2000 of the 10 000 MC samples
    (a single evaluation is inherently one thread in the reference: its MC loop is serial); (2) a THREAD SWEEP the way the
    reference itself parallelises -- T independent evaluations under OpenMP, one State + RNG per thread
    (gpp_optimization.hpp:1472-1546) -- at T in {8, 32, 64, 128, 256} (up to the core count), keeping the best throughput.
    Every figure is scaled linearly in M to the full 10 000 samples.  `value` is the BEST CPU
    throughput found (what the >= 10x target is judged against); `cores` the threads it used."""
    try:
        from oracle import ref
        if ref.available():
            ncores = ref.num_procs()
            gp = ref.RefGP(1, w.alpha, w.lengths, w.X, w.y, w.noise, ())
            t_begin = time.time()
            one_mc = max(2000, sample_mc)
            r1 = gp.kg(w.inner_gd, w.bounds, w.discrete, w.Xq, None, one_mc, best, w.kg_normals[: (one_mc + 1) // 2])
            one_wall = r1["seconds"][0] + r1["seconds"][1]
            one_core = 1.0 / (one_wall * w.M / float(one_mc))
            sweep = [{"threads": 1, "evals_per_s": one_core, "sample_mc": one_mc, "wall_s": one_wall}]
            for T in (8, 32, 64, 128, 256):
                if T > ncores or time.time() - t_begin > budget_s:
                    break
                mc = sample_mc if T <= 64 else max(sample_mc // 2, 100)
                Xq_all = np.ascontiguousarray(w.Xq_restarts[np.arange(T) % len(w.Xq_restarts)])
                _, _, wall = gp.kg_grad_batch(w.inner_gd, w.bounds, w.discrete, Xq_all, mc, best, w.kg_normals[: (mc + 1) // 2], T)
                sweep.append({"threads": T, "evals_per_s": T / (wall * w.M / float(mc)), "sample_mc": mc, "wall_s": wall})
            top = max(sweep, key=lambda e: e["evals_per_s"])
            return {"value": top["evals_per_s"], "unit": "evals/s", "cores": top["threads"], "kind": "reference",
                    "host_cores": ncores, "one_core_evals_per_s": one_core, "thread_sweep": sweep,
                    "sample": "1 core: one ComputeGradKnowledgeGradient at n=%d d=%d q=%d with %d of the %d MC samples "
                              "(%.1f s); sweep: T independent evaluations under OpenMP (one per thread) with 400 (T <= 64) / "
                              "200 MC samples; all scaled linearly in M; value = best throughput of the sweep (at %d threads)"
                              % (w.n, w.d, w.q, one_mc, w.M, one_wall, top["threads"])}
    except Exception as e:  # pragma: no cover
        log("cpu_baseline: reference unavailable (%s); using the C port" % e)
    from oracle import orc
    gp = orc.OrcGP(1, w.alpha, w.lengths, w.X, w.y, w.noise, ())
    t0 = time.time()
    gp.kg(w.inner_gd, w.bounds, w.discrete, w.Xq, None, sample_mc, best, w.kg_normals[: (sample_mc + 1) // 2])
    wall = time.time() - t0
    return {"value": 1.0 / (wall * w.M / float(sample_mc)), "unit": "evals/s", "cores": 1, "kind": "port",
            "sample": "one orc_kg value+gradient at %d of %d MC samples, wall %.2f s, scaled linearly in M" % (sample_mc, w.M, wall)}


def cpu_baseline_c5(w, log):
    """C5 (d-KG, n=2000, g=3: N=8000, M=20 000) takes the reference hours per evaluation on one core, so its time is
    EXTRAPOLATED: the unmodified reference (oracle/_ref) is timed at n in {250, 500, 1000} with two small sample
    counts each, which separates the per-sample cost t_s(N) (two (N+m)^2 triangular sweeps + the inner optimisation's passes
    over N entries: fitted as a N + b N^2) from the per-evaluation cost T_0(N) (state set-up and the (N+m)^3/3 refactorisation
    of the fantasy GP: fitted as c N^2 + e N^3); then T(C5) = T_0(8000) + 20 000 t_s(8000)."""
    from cornell_moe_amd.workloads import make_workload
    from oracle import ref
    if not ref.available():
        return {"value": None, "unit": "evals/s", "cores": 1, "kind": "reference", "sample": "oracle/_ref not built"}
    M1, M2 = 8, 24
    Ns, ts, T0 = [], [], []
    rows = []
    for n in (250, 500, 1000):
        ww = make_workload("C5", n=n, M=M2)
        gp = ref.RefGP(1, ww.alpha, ww.lengths, ww.X, ww.y, ww.noise, list(ww.derivs))
        best = float(gp.additional_mean(ww.discrete).min())
        tt = []
        for M in (M1, M2):
            r = gp.kg(ww.inner_gd, ww.bounds, ww.discrete, ww.Xq, None, M, best, ww.kg_normals[: (M + 1) // 2])
            tt.append(r["seconds"][0] + r["seconds"][1])
        per_sample = (tt[1] - tt[0]) / float(M2 - M1)
        fixed = tt[0] - M1 * per_sample
        N = n * (1 + ww.g)
        Ns.append(float(N))
        ts.append(per_sample)
        T0.append(fixed)
        rows.append({"n": n, "N": N, "seconds_M%d" % M1: tt[0], "seconds_M%d" % M2: tt[1], "per_sample_s": per_sample, "fixed_s": fixed})
        log("cpu_baseline C5: n=%d N=%d: %.2f s at M=%d, %.2f s at M=%d" % (n, N, tt[0], M1, tt[1], M2))
    Ns = np.array(Ns)
    # non-negative least squares: with three noisy timings an unconstrained fit can return a negative coefficient, which at
    # N = 8000 (twice the largest measured size) may even turn the extrapolated time negative
    try:
        from scipy.optimize import nnls
        ab = nnls(np.c_[Ns, Ns ** 2], np.maximum(np.array(ts), 0.0))[0]
        ce = nnls(np.c_[Ns ** 2, Ns ** 3], np.maximum(np.array(T0), 0.0))[0]
    except ImportError:  # leading terms through the largest size
        ab = np.array([0.0, max(ts[-1], 0.0) / Ns[-1] ** 2])
        ce = np.array([0.0, max(T0[-1], 0.0) / Ns[-1] ** 3])
    Nc = float(w.n * (1 + w.g))
    t_s = float(ab[0] * Nc + ab[1] * Nc ** 2)
    t_0 = float(ce[0] * Nc ** 2 + ce[1] * Nc ** 3)
    total = t_0 + w.M * t_s
    return {"value": 1.0 / total, "unit": "evals/s", "cores": 1, "kind": "reference", "extrapolated_seconds_per_eval": total,
            "model": "T = T0(N) + M t_s(N); t_s = a N + b N^2, T0 = c N^2 + e N^3 (non-negative least squares over the three sizes)",
            "fit": {"a": float(ab[0]), "b": float(ab[1]), "c": float(ce[0]), "e": float(ce[1]), "t_s_at_C5": t_s, "T0_at_C5": t_0},
            "measurements": rows,
            "sample": "reference ComputeGradKnowledgeGradient (1 core) at n = 250, 500, 1000 (N = 1000, 2000, 4000) with %d and %d MC "
                      "samples, extrapolated to N = %d, M = %d" % (M1, M2, int(Nc), w.M)}


def committed_traffic():
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/hbm_traffic.json)."""
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as fh:
            return json.load(fh)
    except Exception:
        return {}


def measure_traffic(restarts, log, timeout_s=150):
    """HBM bytes per launch of the two reported kernels, measured NOW: two rocprofv3 passes (FETCH_SIZE, then WRITE_SIZE, each in
    its own --pmc run with --kernel-trace only, as MI355X_MICROARCH.md prescribes; FETCH_SIZE doubled for gfx950) over
    tools/prof_kg.py -- the same batched evaluation this file times -- reduced by tools/hbm_traffic.py.  Returns
    (dict kernel -> bytes per launch, source string); falls back to the committed measurement when the profiler is missing."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="moe_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", os.path.join(tmp, ctr.lower()), "-o", "p",
                   "--", sys.executable, os.path.join(ROOT, "tools", "prof_kg.py"), "C3", str(restarts), "2"]
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=True)
        res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "hbm_traffic.py"), tmp], stdout=subprocess.PIPE,
                             universal_newlines=True, timeout=60, check=True)
        data = json.loads(res.stdout.strip().splitlines()[-1])
        out = {k: float(v["hbm_bytes_per_launch"]) for k, v in data.items()}
        if "kg_mc_kernel" not in out:
            return None, "rocprofv3 ran but reported no kg_mc_kernel rows"
        return out, "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes run by this bench.py invocation"
    except Exception as e:  # pragma: no cover
        log("measure_traffic failed: %s" % e)
        return None, "in-run PMC passes failed (%s)" % type(e).__name__
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def self_launch(args_list, n, script=None):
    """`python bench.py --gpus N` called plainly: re-launch under torch.distributed.run, one rank per GPU, relay the output.
    (`script` is a test hook: tests/test_host_logic.py drives this launcher with a gloo stand-in.)"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MOE_BENCH_SELF_LAUNCHED="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), script or os.path.abspath(__file__)] + args_list
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--restarts", type=int, default=None, help="independent KG evaluations per GPU per step (default: 8 at C3, 2 at C5)")
    ap.add_argument("--shard", choices=["restarts", "mc"], default="restarts")
    ap.add_argument("--config", default="C3")
    ap.add_argument("--cpu-sample-mc", type=int, default=400)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the in-run rocprofv3 PMC passes (HBM traffic)")
    ap.add_argument("--no-mc-shard", action="store_true", help="N > 1: skip the extra MC-sharded measurement")
    ap.add_argument("--no-batch1", action="store_true", help="skip the extra one-at-a-time (batch-1 latency) measurement")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(sys.argv[1:], args.gpus))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    def log(msg):
        if rank == 0:
            print("[bench] " + msg, file=sys.stderr, flush=True)

    import torch
    from cornell_moe_amd import _lib, dist as mdist
    from cornell_moe_amd.api import DeviceGP
    from cornell_moe_amd.workloads import make_workload

    _lib.load()
    _lib.require_gpu()
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # MOE_BENCH_BACKEND=gloo is a test hook: it lets the N > 1 code path run with several ranks sharing ONE GPU (collectives
    # on host tensors); the measured configuration is always nccl (= RCCL), one rank per GPU.
    backend = os.environ.get("MOE_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev if backend == "nccl" else None   # where collective buffers live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    R = args.restarts if args.restarts is not None else (2 if args.config == "C5" else 8)
    w = make_workload(args.config, num_restarts=R * world)
    G = DeviceGP(w.hyperparameters, w.X, w.y, w.noise, w.derivs, device=local_rank)
    best = float(G.additional_mean(w.discrete).min())  # knowledge_gradient.py:366-368
    my_restarts = w.Xq_restarts[rank * R:(rank + 1) * R]

    def step(mode=None, restarts=None):
        mode = mode or args.shard
        if mode == "restarts":
            mine = my_restarts if restarts is None else restarts
            r = G.kg_batch(w.inner_gd, w.bounds, w.discrete, mine, None, w.M, best, w.kg_normals)
            kg = r["kg_sum"] / w.M
            grad = r["grad_sum"] / w.M
            if world > 1 and restarts is None:
                idx = list(range(rank * R, (rank + 1) * R))
                kg, grad = mdist.gather_restarts(idx, kg, grad, R * world, device=cdev)
            return kg, grad, r
        first, count = mdist.shard_samples(w.M, rank, world)
        Rm = R if restarts is None else len(restarts)
        r = G.kg_batch(w.inner_gd, w.bounds, w.discrete, w.Xq_restarts[:Rm], None, w.M, best, w.kg_normals,
                       first_sample=first, num_local=count)
        kg, grad = r["kg_sum"], r["grad_sum"]
        if world > 1:
            buf = torch.from_numpy(np.concatenate([kg[:, None], grad.reshape(Rm, -1)], axis=1))
            buf = buf.to(cdev) if cdev is not None else buf
            dist.all_reduce(buf)
            out = buf.cpu().numpy()
            kg, grad = out[:, 0], out[:, 1:].reshape(grad.shape)
        return kg / w.M, grad / w.M, r

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=cdev if cdev is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    ms_mc = ms_cov = ms_tail = ms_state = 0.0
    val_passes = grad_passes = 0
    for _ in range(args.steps):
        kg, grad, r = step()
        km = G.last_kernel_ms()
        ms_mc += km["mc"]
        ms_cov += km["cov_build"]
        ms_tail += km["tail"]
        ms_state += km["state"]
        val_passes += r["mean_evals"]
        grad_passes += r["grad_evals"]
    local_elapsed = time.perf_counter() - t0     # this rank's own time for its K steps (before the closing barrier)
    fence()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    assert np.all(np.isfinite(kg)) and np.all(np.isfinite(grad))

    evals_per_step = R * world if args.shard == "restarts" else R
    total_evals = evals_per_step * args.steps
    value = total_evals / elapsed
    per_rank = [R * args.steps / local_elapsed if args.shard == "restarts" else args.steps * R / local_elapsed]
    if world > 1:
        t = torch.tensor(per_rank, dtype=torch.float64, device=cdev if cdev is not None else "cpu")
        gathered = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        per_rank = [float(g.item()) for g in gathered]

    # ---- extras, OUTSIDE the timed region of `value` ----
    extras = {}
    if world > 1 and args.shard == "restarts" and not args.no_mc_shard:
        # MC-sample sharding of ONE evaluation at a time (strong scaling of the single-evaluation latency): every rank takes an
        # even-aligned slice of the 10k samples, ONE all_reduce of 1 + q d doubles per evaluation
        one = w.Xq_restarts[:1]
        ksteps = max(10, args.steps)
        for _ in range(3):
            step("mc", one)
        fence()
        t1 = time.perf_counter()
        for _ in range(ksteps):
            step("mc", one)
        fence()
        dt = max_over_ranks(time.perf_counter() - t1)
        extras["mc_shard"] = {"value": ksteps / dt, "unit": "evals/s", "ms_per_eval": 1e3 * dt / ksteps, "evals_per_step": 1,
                              "samples_per_rank": mdist.shard_samples(w.M, 0, world)[1],
                              "collective": "one all_reduce(SUM) of %d doubles per evaluation" % (1 + w.q * w.d)}
    if not args.no_batch1 and args.shard == "restarts":
        # one evaluation per call (what compute_grad_knowledge_gradient does): the batch-1 latency next to the batch-R rate
        one = my_restarts[:1]
        ksteps = max(20, args.steps)
        for _ in range(5):
            step("restarts", one)
        fence()
        t1 = time.perf_counter()
        for _ in range(ksteps):
            step("restarts", one)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        extras["batch1"] = {"value": ksteps / dt, "unit": "evals/s per GPU", "ms_per_eval": 1e3 * dt / ksteps,
                            "note": "one evaluation per moe_kg_batch call, one at a time (rank 0)"}
        fence()

    if rank == 0:
        # ---- roofline of the dominant kernel (MC inner optimisation: FP64 vector-ALU bound) ----
        local_evals = R * args.steps                      # evaluations whose kernels this rank launched
        mc_ms = ms_mc / args.steps                        # avg MC-kernel ms per evaluation (HIP events, library stream)
        g1 = 1 + w.g
        npts = (w.n + w.q) * g1                           # N + m covariance entries each pass walks
        n_local = w.M if args.shard == "restarts" else mdist.shard_samples(w.M, 0, world)[1]
        S = val_passes / float(local_evals * n_local)     # counted value passes per sample
        Gp = grad_passes / float(local_evals * n_local)   # counted value+gradient passes per sample
        flops = n_local * npts * (S * (3 * w.d + 32) + Gp * (5 * w.d + 34))   # SURVEY 8(d) per-entry figures
        ach_tflops = flops / (mc_ms * 1e-3) / 1e12
        # ---- roofline of the covariance-assembly kernel (HBM write-bound) ----
        # The q-KG gradient tail no longer materialises T = K(X, x*) (kg.hip: launch_fused_tail), so the assembly kernel
        # (a3/a4: GP build, posterior queries, the d-KG tail) is measured live on the SAME N x M shape the tail used to
        # build -- N = n training rows x (R * M) columns, 640 MB at C3 -- through moe_cov_build_probe (HIP events on the
        # library's stream around `repeat` launches).
        probe_pts = np.random.default_rng(7).uniform(size=(R * n_local, w.d))
        cov_launch_ms, cov_bytes_launch = G.cov_build_probe(probe_pts, repeat=10)
        cov_bytes = cov_bytes_launch / R                                    # 8[nA d + nB d + nA nB]
        cov_tbs = cov_bytes_launch / (cov_launch_ms * 1e-3) / 1e12 if cov_launch_ms > 0 else 0.0
        traffic, traffic_src = (None, "skipped (--no-traffic)")
        if world == 1 and not args.no_traffic and args.config == "C3":
            traffic, traffic_src = measure_traffic(R, log)
        if traffic is None:
            committed = committed_traffic()
            traffic = {k: float(v["hbm_bytes_per_launch"]) for k, v in committed.items() if "hbm_bytes_per_launch" in v}
            traffic_src = "profiles/hbm_traffic.json (committed rocprofv3 PMC passes of an earlier run; %s)" % traffic_src
        mc_kernel = "kg_mc_kernel" if (w.g == 0 and w.n + w.q <= 1600) else "kg_mc_block_kernel"
        out = {
            "metric": ("q-KG gradient evals/s (n=1000,d=8,q=4,10k MC)" if args.config in ("C3", "C4") else
                       "%s gradient evals/s (n=%d,d=%d,q=%d,g=%d,%d MC) [secondary configuration %s]"
                       % ("d-KG" if w.g else "q-KG", w.n, w.d, w.q, w.g, w.M, args.config)),
            "value": value, "unit": "evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak" if args.shard == "restarts" else "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %s value+gradient, n=%d d=%d q=%d g=%d M=%d MC, P=%d discrete pts, Matern-5/2, inner GD "
                                   "(1,6,1,3,0,1,0.1,1e-10); %d evaluations per GPU per step"
                                   % (args.config, "d-KG" if w.g else "q-KG", w.n, w.d, w.q, w.g, w.M, w.P, R),
                       "shard": args.shard, "evals_per_step": evals_per_step},
            "rccl_ranks": world if (world > 1 and backend == "nccl") else 0,
            "per_rank_evals_per_s": per_rank,
            "roofline": {"bound": "fp64_valu", "achieved": ach_tflops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach_tflops / FP64_PEAK_TFLOPS,
                         "traffic": traffic.get("kg_mc_kernel") if mc_kernel == "kg_mc_kernel" else None,
                         "traffic_source": traffic_src,
                         "kernel": mc_kernel, "avg_launch_ms": mc_ms * R, "avg_ms_per_eval": mc_ms,
                         "evals_per_launch": R, "value_passes_per_sample": S, "grad_passes_per_sample": Gp,
                         "entries_per_pass": npts,
                         "note": "FP64 vector-ALU bound (sqrt + exp per covariance entry).  The kernel issues no MFMA: on gfx950 "
                                 "v_mfma_f64 and FP64 VALU instructions do not overlap (measured: profiles/"
                                 "r02_coissue_mfma_vs_valu.txt) and both peak at 78.6 TFLOP/s, the peak used here.  achieved = "
                                 "SURVEY 8(d) algorithmic flops -- (n+u)(1+g) covariance entries per pass x [S (3d+32) + "
                                 "G (5d+34)] with the DEVICE-COUNTED passes S, G -- / HIP-event kernel time on the library's "
                                 "stream; one launch covers all evaluations of a step; traffic = HBM bytes per launch (PMC), "
                                 "tiny next to the compute time"},
            "roofline_cov_build": {"bound": "hbm", "achieved": cov_tbs * 1e3, "peak": HBM_PEAK_TBS * 1e3, "unit": "GB/s",
                                   "frac": cov_tbs / HBM_PEAK_TBS, "traffic": traffic.get("cov_build_kernel"),
                                   "traffic_source": traffic_src,
                                   "kernel": "cov_build_kernel, N x (R M) = %d x %d, measured by moe_cov_build_probe (the q-KG "
                                             "tail itself no longer writes this matrix: it recomputes the entries where "
                                             "they are consumed)" % (w.n, R * n_local),
                                   "avg_launch_ms": cov_launch_ms, "bytes_per_launch": cov_bytes_launch,
                                   "bytes_per_eval_equiv": cov_bytes},
            "kernel_ms_per_eval": {"mc": mc_ms, "cov_build": ms_cov / args.steps, "tail": ms_tail / args.steps,
                                   "state_host": ms_state / args.steps},
        }
        out.update(extras)
        if not args.no_cpu_baseline and world == 1 and args.config == "C5":
            out["cpu_baseline"] = cpu_baseline_c5(w, log)
            out["speedup_vs_cpu_one_core"] = value / out["cpu_baseline"]["value"]
        elif not args.no_cpu_baseline and world == 1:  # reported baseline: rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(w, best, args.cpu_sample_mc, log)
            out["speedup_vs_cpu_best"] = value / out["cpu_baseline"]["value"]
            if "one_core_evals_per_s" in out["cpu_baseline"]:
                out["speedup_vs_cpu_one_core"] = value / out["cpu_baseline"]["one_core_evals_per_s"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
