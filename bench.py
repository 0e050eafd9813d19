#!/usr/bin/env python
"""bench.py -- q-KG gradient evaluations / s on BASELINE.json's headline configuration (C3: n=1000, d=8, q=4, 10k MC).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the hot path over one batch: every rank evaluates `--restarts` (default 8) independent q-KG
value+gradient evaluations (different points_to_sample, same GP / discrete set / normal table = the multistart axis of
ComputeKGOptimalPointsToSampleViaMultistartGradientDescent, gpp_knowledge_gradient_optimization.hpp:860-935; C4 is 64
restarts over 8 GPUs = 8 per GPU), then -- when N > 1 -- all ranks exchange their (KG, grad KG) with ONE RCCL all_gather.
Per-GPU work is fixed as N grows ("scaling": "weak").  `--shard mc` instead splits the 10k MC samples of each evaluation
across ranks with one all_reduce per evaluation (strong scaling of a single evaluation).
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


def cpu_baseline(w, best, sample_mc, log):
    """The reference CPU path (oracle/_ref, unmodified C++) on a bounded sample of the same workload, all host cores,
    parallelised the way the reference itself is (independent evaluations under OpenMP, one State+RNG per thread)."""
    try:
        from oracle import ref
        if ref.available():
            cores = ref.num_procs()
            gp = ref.RefGP(1, w.alpha, w.lengths, w.X, w.y, w.noise, ())
            Xq_all = np.ascontiguousarray(w.Xq_restarts[np.arange(cores) % len(w.Xq_restarts)])
            nm = w.kg_normals[: (sample_mc + 1) // 2]
            _, _, wall = gp.kg_grad_batch(w.inner_gd, w.bounds, w.discrete, Xq_all, sample_mc, best, nm, cores)
            per_eval_full = wall * (w.M / float(sample_mc))  # linear in M
            return {"value": cores / per_eval_full, "unit": "evals/s", "cores": cores, "kind": "reference",
                    "sample": "%d independent ComputeGradKnowledgeGradient calls (one per core, OpenMP) at n=%d d=%d q=%d with "
                              "%d of the %d MC samples, wall %.2f s, scaled linearly in M" % (cores, w.n, w.d, w.q, sample_mc,
                                                                                             w.M, wall)}
    except Exception as e:  # pragma: no cover
        log("cpu_baseline: reference unavailable (%s); using the C port" % e)
    from oracle import orc
    gp = orc.OrcGP(1, w.alpha, w.lengths, w.X, w.y, w.noise, ())
    t0 = time.time()
    gp.kg(w.inner_gd, w.bounds, w.discrete, w.Xq, None, sample_mc, best, w.kg_normals[: (sample_mc + 1) // 2])
    wall = time.time() - t0
    return {"value": 1.0 / (wall * w.M / float(sample_mc)), "unit": "evals/s", "cores": 1, "kind": "port",
            "sample": "one orc_kg value+gradient at %d of %d MC samples, wall %.2f s, scaled linearly in M" % (sample_mc, w.M, wall)}


def measured_traffic(kernel):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (tools/gpu_round.sh -> tools/hbm_traffic.py:
    FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
    gfx950).  bench.py cannot run the profiler on itself, so it reports the latest committed measurement or null."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as fh:
            return float(json.load(fh)[kernel]["hbm_bytes_per_launch"])
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--restarts", type=int, default=8, help="independent KG evaluations per GPU per step")
    ap.add_argument("--shard", choices=["restarts", "mc"], default="restarts")
    ap.add_argument("--config", default="C3")
    ap.add_argument("--cpu-sample-mc", type=int, default=400)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))

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

    R = args.restarts
    w = make_workload(args.config, num_restarts=R * world)
    G = DeviceGP(w.hyperparameters, w.X, w.y, w.noise, w.derivs, device=local_rank)
    best = float(G.additional_mean(w.discrete).min())  # knowledge_gradient.py:366-368
    my_restarts = w.Xq_restarts[rank * R:(rank + 1) * R]

    def step():
        if args.shard == "restarts":
            r = G.kg_batch(w.inner_gd, w.bounds, w.discrete, my_restarts, None, w.M, best, w.kg_normals)
            kg = r["kg_sum"] / w.M
            grad = r["grad_sum"] / w.M
            if world > 1:
                idx = list(range(rank * R, (rank + 1) * R))
                kg, grad = mdist.gather_restarts(idx, kg, grad, R * world, device=cdev)
            return kg, grad, r
        first, count = mdist.shard_samples(w.M, rank, world)
        r = G.kg_batch(w.inner_gd, w.bounds, w.discrete, w.Xq_restarts[:R], None, w.M, best, w.kg_normals,
                       first_sample=first, num_local=count)
        kg, grad = r["kg_sum"], r["grad_sum"]
        if world > 1:
            buf = torch.from_numpy(np.concatenate([kg[:, None], grad.reshape(R, -1)], axis=1))
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
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev if cdev is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert np.all(np.isfinite(kg)) and np.all(np.isfinite(grad))

    evals_per_step = R * world if args.shard == "restarts" else R
    total_evals = evals_per_step * args.steps
    value = total_evals / elapsed

    if rank == 0:
        # ---- roofline of the dominant kernel (MC inner optimisation: FP64 vector-ALU bound) ----
        local_evals = R * args.steps                      # evaluations whose kernels this rank launched
        mc_ms = ms_mc / args.steps                        # avg MC-kernel ms per evaluation (HIP events, library stream)
        npts = w.n + w.q                                  # N + m rows each pass walks
        n_local = w.M if args.shard == "restarts" else mdist.shard_samples(w.M, 0, world)[1]
        S = val_passes / float(local_evals * n_local)     # counted value passes per sample
        Gp = grad_passes / float(local_evals * n_local)   # counted value+gradient passes per sample
        flops = n_local * npts * (S * (3 * w.d + 32) + Gp * (5 * w.d + 34))   # SURVEY 8(d) per-point figures
        ach_tflops = flops / (mc_ms * 1e-3) / 1e12
        # ---- roofline of the covariance-assembly kernel (HBM write-bound) ----
        # The q-KG gradient tail no longer materialises T = K(X, x*) (kg.hip: launch_fused_tail), so the assembly kernel
        # (a3/a4: GP build, posterior queries, the d-KG tail) is measured live on the SAME N x M shape the tail used to
        # build -- N = n training rows x (R * M) columns, 640 MB at C3 -- through moe_cov_build_probe (HIP events on the
        # library's stream around `repeat` launches).
        probe_pts = np.random.default_rng(7).uniform(size=(R * n_local, w.d))
        cov_launch_ms, cov_bytes_launch = G.cov_build_probe(probe_pts, repeat=10)
        cov_ms = cov_launch_ms / R
        cov_bytes = cov_bytes_launch / R                                    # 8[nA d + nB d + nA nB]
        cov_tbs = cov_bytes_launch / (cov_launch_ms * 1e-3) / 1e12 if cov_launch_ms > 0 else 0.0
        out = {
            "metric": "q-KG gradient evals/s (n=1000,d=8,q=4,10k MC)", "value": value, "unit": "evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak" if args.shard == "restarts" else "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: q-KG value+gradient, n=%d d=%d q=%d M=%d MC, P=%d discrete pts, Matern-5/2, inner GD "
                                   "(1,6,1,3,0,1,0.1,1e-10); %d evaluations per GPU per step" % (args.config, w.n, w.d, w.q, w.M,
                                                                                                 w.P, R),
                       "shard": args.shard, "evals_per_step": evals_per_step},
            "roofline": {"bound": "mfma", "achieved": ach_tflops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach_tflops / FP64_PEAK_TFLOPS, "traffic": measured_traffic("kg_mc_kernel"),
                         "kernel": "kg_mc_kernel", "avg_launch_ms": mc_ms * R, "avg_ms_per_eval": mc_ms,
                         "evals_per_launch": R, "value_passes_per_sample": S, "grad_passes_per_sample": Gp,
                         "note": "dominant kernel is FP64 vector-ALU bound (sqrt + exp per covariance entry); it does not use "
                                 "MFMA -- on gfx950 the dense FP64 MFMA peak equals the FP64 vector peak (78.6 TFLOP/s), "
                                 "which is the peak used here; achieved = SURVEY 8(d) algorithmic flops with the device-counted "
                                 "passes / HIP-event kernel time; one launch covers all evaluations of a step; traffic = "
                                 "measured HBM bytes per launch (bytes, PMC), tiny next to the compute time"},
            "roofline_cov_build": {"bound": "hbm", "achieved": cov_tbs * 1e3, "peak": HBM_PEAK_TBS * 1e3, "unit": "GB/s",
                                   "frac": cov_tbs / HBM_PEAK_TBS, "traffic": measured_traffic("cov_build_kernel"),
                                   "kernel": "cov_build_kernel, N x (R M) = %d x %d, measured by moe_cov_build_probe (the q-KG "
                                             "tail itself no longer writes this matrix: it recomputes the entries where "
                                             "they are consumed)" % (w.n, R * n_local),
                                   "avg_launch_ms": cov_launch_ms, "bytes_per_launch": cov_bytes_launch,
                                   "bytes_per_eval_equiv": cov_bytes},
            "kernel_ms_per_eval": {"mc": mc_ms, "cov_build": ms_cov / args.steps, "tail": ms_tail / args.steps,
                                   "state_host": ms_state / args.steps},
        }
        if not args.no_cpu_baseline and world == 1:  # reported baseline: rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(w, best, args.cpu_sample_mc, log)
            out["speedup_vs_cpu_all_cores"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
