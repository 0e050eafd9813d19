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


def test_bench_gpus2_self_launch():
    env = dict(os.environ, MOE_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--restarts", "2", "--no-cpu-baseline", "--no-traffic"], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, universal_newlines=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["evals_per_step"] == 4 and out["scaling"] == "weak"
    assert len(out["per_rank_evals_per_s"]) == 2 and all(v > 0 for v in out["per_rank_evals_per_s"])
    assert out["value"] > 0 and out["mc_shard"]["value"] > 0 and out["mc_shard"]["samples_per_rank"] == 5000
    assert out["roofline"]["bound"] == "fp64_valu" and 0.05 < out["roofline"]["frac"] < 1.0
    assert "cpu_baseline" not in out  # rank 0 at N = 1 only
