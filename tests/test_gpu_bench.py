"""GPU test of bench.py's own N > 1 path: `python bench.py --gpus 2` called PLAINLY must launch its two ranks itself
(torch.distributed.run, 127.0.0.1), run the restart-sharded steps with the all_gather, the MC-sharded extra with its
all_reduce, and print ONE JSON line from rank 0.  The GPU box has one GPU, so the ranks share it and the collectives run on
gloo (MOE_BENCH_BACKEND, a test hook); the measured configuration is always RCCL, one rank per GPU."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *flags, gpus=2):
    env = dict(os.environ, **extra_env)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--no-traffic", "--no-extras"] + list(flags), env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, universal_newlines=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0]), res.stderr


def test_bench_gpus2_self_launch():
    out, _ = _run({"MOE_BENCH_BACKEND": "gloo"}, "--restarts", "2")
    assert out["n_gpus"] == 2 and out["config"]["evals_per_step"] == 4 and out["scaling"] == "weak"
    assert len(out["per_rank_evals_per_s"]) == 2 and all(v > 0 for v in out["per_rank_evals_per_s"])
    assert out["value"] > 0 and out["mc_shard"]["value"] > 0 and out["mc_shard"]["samples_per_rank"] == 5000
    assert out["roofline"]["bound"] == "fp64_valu" and 0.05 < out["roofline"]["frac"] < 1.0
    assert "cpu_baseline" not in out  # rank 0 at N = 1 only
    # cross-rank determinism: every restart recomputed by rank 0 alone is bit-identical; an MC-sharded evaluation agrees to 1e-12
    assert out["determinism"]["ok"] and out["determinism"]["max_rel_diff_vs_one_rank"] == 0.0 and out["determinism"]["restarts"] == 4
    assert out["mc_shard"]["max_rel_diff_vs_unsharded"] <= 1e-12
    assert out["collective_backend"] == "gloo" and out["rccl_ranks"] == 0 and out["fallback"] is None


def test_bench_rccl_preflight_failure_falls_back_to_gloo():
    """RCCL preferred, but its pre-flight cannot pass (two ranks, one GPU): the run must go on with gloo collectives and say so."""
    out, err = _run({"MOE_BENCH_SHARE_GPU": "1"}, "--restarts", "1", "--no-mc-shard", "--no-batch1")
    assert out["rccl_ranks"] == 0 and out["collective_backend"] == "gloo" and "pre-flight" in out["fallback"]
    assert "RCCL NOT USED" in err
    assert out["value"] > 0 and out["determinism"]["ok"] and out["n_gpus"] == 2


def test_bench_no_process_group_falls_back_to_c_abi_driver():
    """Not even the rendezvous works: rank 0 drives the visible devices in-process through moe_kg_batch_multi, rank 1 leaves."""
    out, err = _run({"MOE_BENCH_SHARE_GPU": "1", "MOE_DIST_FAIL": "rendezvous"}, "--restarts", "1")
    assert out["fallback"].startswith("moe_kg_batch_multi") and out["rccl_ranks"] == 0 and out["devices_used"] == 1
    assert out["value"] > 0 and out["config"]["evals_per_step"] == 2 and "NO PROCESS GROUP" in err


def test_bench_default_step_is_the_c4_job():
    """Without --restarts a step is 64 restarts shared by the ranks (C4), strong scaling."""
    out, _ = _run({"MOE_BENCH_BACKEND": "gloo"}, "--no-mc-shard", "--no-batch1")
    assert out["config"]["evals_per_step"] == 64 and out["config"]["evals_per_gpu_per_step"] == 32 and out["scaling"] == "strong"
    assert out["determinism"]["ok"] and out["determinism"]["restarts"] == 64


def test_bench_world8_c4_job_on_one_gpu():
    """The C4 job at its REAL world size: eight ranks under torch.distributed.run -- rendezvous, gloo control
    plane, the data-plane group, gather_restarts with 8 restarts per rank, the determinism check over all 64 -- sharing the test
    box's one GPU (the collectives on gloo: the test hook); what the driver's 8-GPU run does differently is RCCL and one GPU each."""
    out, _ = _run({"MOE_BENCH_BACKEND": "gloo"}, "--no-mc-shard", "--no-batch1", gpus=8)
    assert out["n_gpus"] == 8 and out["scaling"] == "strong"
    assert out["config"]["evals_per_step"] == 64 and out["config"]["evals_per_gpu_per_step"] == 8
    assert len(out["per_rank_evals_per_s"]) == 8 and all(v > 0 for v in out["per_rank_evals_per_s"])
    assert out["determinism"]["ok"] and out["determinism"]["restarts"] == 64 and out["determinism"]["max_rel_diff_vs_one_rank"] == 0.0
    assert out["collective_us_per_step"]["rank0"] > 0 and out["collective_backend"] == "gloo"


def test_bench_c5_mc_sharded_on_two_ranks():
    """BASELINE configs[4] names d-KG at C5 "x 8 MI355X": its launcher -- `bench.py --gpus N --config C5 --shard mc`, every rank an
    even-aligned slice of the 20 000 samples of each evaluation, ONE all_reduce of 1 + q d doubles per evaluation -- on two ranks
    sharing the test box's GPU (gloo hook).  (r6)"""
    out, _ = _run({"MOE_BENCH_BACKEND": "gloo"}, "--config", "C5", "--shard", "mc", "--restarts", "1", "--no-batch1", "--no-determinism")
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["shard"] == "mc"
    assert "n=2000" in out["config"]["workload"] and "g=3" in out["config"]["workload"] and "M=20000" in out["config"]["workload"]
    assert out["value"] > 0 and out["roofline"]["kernel"] == "kg_mc_stream_kernel" and 0.05 < out["roofline"]["frac"] < 1.0
    assert out["collective_backend"] == "gloo" and out["rccl_ranks"] == 0


def test_rccl_preflight_child_passes_under_a_launcher_environment():
    """The pre-flight child of dist.bring_up as a rank would start it UNDER torch.distributed.run -- TORCHELASTIC_* variables set,
    its own rendezvous port -- with the one GPU of the test box as a world of 1: RCCL bring-up, one checked all_reduce, exit 0.
    (With TORCHELASTIC_USE_AGENT_STORE inherited the child would look for the launcher's store on a port nobody serves.)"""
    from cornell_moe_amd import dist as mdist
    saved = dict(os.environ)
    try:
        os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="1",
                          TORCHELASTIC_USE_AGENT_STORE="True", TORCHELASTIC_RUN_ID="none", TORCHELASTIC_MAX_RESTARTS="0")
        env = mdist.preflight_env(mdist._free_port())
    finally:
        os.environ.clear()
        os.environ.update(saved)
    assert not any(k.startswith("TORCHELASTIC_") for k in env) and env["MASTER_PORT"] != "1"
    res = subprocess.run([sys.executable, "-m", "cornell_moe_amd.dist", "--preflight"], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, universal_newlines=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]


def test_single_buffer_exchange_over_rccl_world1():
    """r6: dist._allgather_flat / gather_restarts / Exchange with their buffers ON THE DEVICE over the nccl (= RCCL) backend -- what the
    driver's multi-GPU run does first -- as a world of one on the test box's GPU: all_gather_into_tensor into the preallocated tensor,
    one copy back, the values intact."""
    code = r'''
import os, numpy as np, torch, torch.distributed as dist
from cornell_moe_amd import dist as mdist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(mdist._free_port()))
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
bufs = {}
send = np.linspace(0.0, 1.0, 33)
a = mdist._allgather_flat(send, 1, None, dev, bufs)
b = mdist._allgather_flat(send * 2.0, 1, None, dev, bufs)
assert np.array_equal(a, send) and np.array_equal(b, 2.0 * send) and len(bufs) == 1 and bufs[(33, str(dev))][1].is_cuda
kg, grad = mdist.gather_restarts([0, 1, 2], [1.0, 2.0, 3.0], np.arange(3 * 4 * 2, dtype=float).reshape(3, 4, 2), 3, group=None, device=dev)
assert np.array_equal(kg, [1.0, 2.0, 3.0]) and np.array_equal(grad.ravel(), np.arange(24.0))
ex = mdist.Exchange(0, 1, None, dev)
out = ex._torch_allgather(send)
assert np.array_equal(out, send)
dist.destroy_process_group()
print("rccl world-1 exchange ok")
'''
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         universal_newlines=True, timeout=300)
    assert res.returncode == 0 and "rccl world-1 exchange ok" in res.stdout, res.stderr[-2000:]
