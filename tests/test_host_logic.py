"""CPU tests (-m "not gpu"): host-side logic -- workloads, MC/restart sharding, and the N>1 collectives on gloo
(world_size 2, multi-process) with a closed-form per-sample contribution standing in for the device evaluation of a shard."""
import os
import socket

import numpy as np
import pytest

from cornell_moe_amd import dist as mdist
from cornell_moe_amd.workloads import kg_normals_full, make_workload


def test_workload_shapes_and_antithetic_table():
    w = make_workload("C3", M=101)
    assert w.X.shape == (1000, 8) and w.Xq.shape == (4, 8) and w.discrete.shape == (10, 8)
    assert w.kg_normals.shape == (51, 4)
    full = kg_normals_full(w)
    assert full.shape == (101, 4)
    assert np.array_equal(full[0::2], w.kg_normals) and np.array_equal(full[1::2], -w.kg_normals[:50])
    w5 = make_workload("C5", n=20, M=8)
    assert w5.y.shape == (20, 4) and w5.m == 8 * 4 and w5.derivs == (0, 1, 2)


@pytest.mark.parametrize("num_mc,world", [(10000, 8), (10000, 3), (7, 2), (1, 4), (2, 8), (101, 5)])
def test_shard_samples_partition(num_mc, world):
    cover = []
    for r in range(world):
        first, count = mdist.shard_samples(num_mc, r, world)
        assert first % 2 == 0 and count >= 0
        if count:
            assert first + count <= num_mc
            assert count % 2 == 0 or first + count == num_mc  # only the globally last slice may end on an odd count
        cover.extend(range(first, first + count))
    assert cover == list(range(num_mc))


def test_shard_restarts_partition():
    allr = sorted(sum((mdist.shard_restarts(64, r, 8) for r in range(8)), []))
    assert allr == list(range(64))
    assert mdist.shard_restarts(3, 2, 8) == [2] and mdist.shard_restarts(3, 5, 8) == []


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, num_mc, q, d, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # per-sample contributions known in closed form: sample i contributes (i + 1) to KG and i * ones to grad
        def eval_shard(first, count):
            idx = np.arange(first, first + count, dtype=np.float64)
            return float(np.sum(idx + 1.0)), np.sum(idx) * np.ones((q, d))
        kg, grad = mdist.kg_grad_mc_sharded(eval_shard, num_mc, rank, world)
        mine = mdist.shard_restarts(5, rank, world)
        lkg = [10.0 + i for i in mine]
        lgrad = np.array([np.full((q, d), float(i)) for i in mine]).reshape(len(mine), q, d)
        akg, agrad = mdist.gather_restarts(mine, lkg, lgrad, 5)
        ret[rank] = (kg, grad, akg, agrad)
    finally:
        dist.destroy_process_group()


def test_gloo_world2_allreduce_and_gather():
    import torch.multiprocessing as mp
    world, num_mc, q, d = 2, 1001, 2, 3
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, num_mc, q, d, ret), nprocs=world, join=True)
    total = num_mc * (num_mc + 1) / 2.0
    for r in range(world):
        kg, grad, akg, agrad = ret[r]
        assert kg == pytest.approx(total / num_mc, rel=1e-15)
        assert np.allclose(grad, (num_mc * (num_mc - 1) / 2.0) / num_mc, rtol=1e-15)
        assert np.array_equal(akg, 10.0 + np.arange(5))
        assert all(np.all(agrad[i] == i) for i in range(5))


def _mcmc_worker(rank, world, port, ret):
    """GP-index shard of an MCMC-averaged KG evaluation: per-GP values from the oracle (test infrastructure standing in for
    the device evaluation, which needs a GPU), the sum over ranks by all_reduce, the mean / cost step by the C ABI's
    host-side moe_kg_mcmc_finalize."""
    import ctypes as C

    import torch.distributed as dist
    from cornell_moe_amd import _lib
    from helpers import load_golden_mcmc
    from oracle import orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        c = load_golden_mcmc()[1]
        i = c.inp
        d, f, q = int(i["d"]), int(i["num_fidelity"]), int(i["q"])
        nm = int(i["num_mcmc"])
        mine = mdist.shard_members(nm, rank, world)

        def local_sums():
            kg, grad = np.zeros(1), np.zeros((1, q, d))
            for k in mine:
                gp = orc.OrcGP(1, float(i["hypers"][k, 0]), i["hypers"][k, 1:], i["X"], i["y"], i["noises"][k], list(i["derivs"]))
                r = gp.kg(i["inner_gd"], i["bounds"][:2 * (d - f)], i["discrete"][k], i["Xq"], i["Xp"], int(i["M"]),
                          float(i["kg_best"][k]), i["kg_normals"], num_fidelity=f)
                kg[0] += r["kg"]
                grad[0] += r["grad"]
            return kg, grad

        def finalize(kg_sum, grad_sum):
            dp = C.POINTER(C.c_double)
            kg = np.array(kg_sum, dtype=np.float64, copy=True)
            grad = np.array(grad_sum, dtype=np.float64, copy=True)
            xq = np.ascontiguousarray(i["Xq"], dtype=np.float64)
            rc = _lib.load().moe_kg_mcmc_finalize(kg.ctypes.data_as(dp), grad.ctypes.data_as(dp), xq.ctypes.data_as(dp), 1, q, d, f, nm)
            assert rc == 0
            return kg, grad

        kg, grad = mdist.kg_mcmc_sharded(local_sums, finalize)
        ret[rank] = (mine, float(kg[0]), grad[0], float(c.out["kg"]), c.out["grad_kg"])
    finally:
        dist.destroy_process_group()


def test_gloo_world2_mcmc_member_shards():
    import torch.multiprocessing as mp
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_mcmc_worker, args=(world, port, ret), nprocs=world, join=True)
    assert sorted(ret[0][0] + ret[1][0]) == [0, 1, 2]
    for r in range(world):
        _, kg, grad, ref_kg, ref_grad = ret[r]
        assert abs(kg - ref_kg) <= 1e-8 * abs(ref_kg)
        assert np.abs(grad - ref_grad).max() <= 1e-8 * max(np.abs(ref_grad).max(), abs(ref_kg))
    assert ret[0][1] == ret[1][1] and np.array_equal(ret[0][2], ret[1][2])  # every rank holds the same reduced result


def _sharded_items(ex, n, width, seed, fail_item=-1):
    import ctypes as C
    from cornell_moe_amd import _lib
    out = np.full(n * width, -1.0)
    err = _lib.MoeError()
    rc = _lib.load().moe_debug_sharded_items(C.byref(ex.c_struct), n, width, seed, fail_item, out.ctypes.data_as(_lib.dp), C.byref(err))
    ex.reraise()
    return rc, out.reshape(n, width), err


def _exchange_worker(rank, world, port, ret):
    """r5: the deal-and-exchange step of the multi-rank outer optimisers (moe_kg_multistart_comm / moe_kg_mcmc_multistart_comm) with
    its all-gather carried by torch.distributed over gloo: dist.Exchange -> moe_comm_t -> csrc sharded_items."""
    import torch.distributed as dist
    from cornell_moe_amd import _lib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ex = mdist.Exchange(rank, world)
        res = []
        for n, width in ((7, 3), (1, 1), (2, 5), (20, 33)):   # fewer items than ranks, ragged last round, a GD step's shape
            rc, out, _ = _sharded_items(ex, n, width, 100.0 * width)
            res.append((rc, out))
        rc_fail, _, err = _sharded_items(ex, 5, 2, 0.0, fail_item=3)   # item 3 belongs to rank 3 % world
        ret[rank] = (res, rc_fail, err.message.decode(), list(err.payload), ex.calls, ex.doubles)
    finally:
        dist.destroy_process_group()


def test_gloo_world2_exchange_of_the_multi_rank_optimisers():
    import torch.multiprocessing as mp
    from cornell_moe_amd import _lib
    world = 2
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_exchange_worker, args=(world, port, ret), nprocs=world, join=True)
    for r in range(world):
        res, rc_fail, msg, payload, calls, doubles = ret[r]
        for (rc, out), (n, width) in zip(res, ((7, 3), (1, 1), (2, 5), (20, 33))):
            assert rc == 0
            want = 100.0 * width + np.arange(n)[:, None] + np.arange(width)[None, :] / 1000.0
            assert np.array_equal(out, want)   # complete, in item order, on every rank
        # the failure of ONE rank comes back from EVERY rank (nobody is left in a collective), with its code and payload
        assert rc_fail == _lib.MOE_ERR_SINGULAR and payload == [3.0, 1.0, 2.0]
        assert ("synthetic failure" in msg) == (r == 3 % world) and (("another rank" in msg) == (r != 3 % world))
        assert calls == 5 and doubles == sum(4 + -(-n // world) * w for n, w in ((7, 3), (1, 1), (2, 5), (20, 33), (5, 2)))


def _flat_worker(rank, world, port, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bufs = {}
        send = np.arange(5, dtype=np.float64) + 100.0 * rank
        a = mdist._allgather_flat(send, world, None, None, bufs)          # one collective into one tensor
        b = mdist._allgather_flat(send + 1.0, world, None, None, bufs)    # the same size again: the tensors are reused
        n_bufs = len(bufs)
        mdist._NO_INTO_TENSOR.add(dist.get_backend(None))                  # a backend without all_gather_into_tensor: list form on views
        c = mdist._allgather_flat(send, world, None, None, bufs)
        ret[rank] = (a, b, c, n_bufs)
    finally:
        mdist._NO_INTO_TENSOR.clear()
        dist.destroy_process_group()


def test_gloo_world2_single_buffer_allgather():
    """r6: the exchange of dist.Exchange / gather_restarts is ONE collective into ONE preallocated tensor and one
    copy back, whichever form of all_gather the backend offers."""
    import torch.multiprocessing as mp
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_flat_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    want = np.concatenate([np.arange(5.0), np.arange(5.0) + 100.0])
    for r in range(world):
        a, b, c, n_bufs = ret[r]
        assert np.array_equal(a, want) and np.array_equal(b, want + 1.0) and np.array_equal(c, want) and n_bufs == 1


def test_native_rccl_exchange_fails_loudly_without_a_device():
    """r6: the library's own RCCL communicator (moe_rccl_*, csrc/rccl_comm.hip).  Without a GPU nothing of it may pretend to work --
    and make_exchange must hand back the torch.distributed exchange instead (world 1: never called)."""
    from cornell_moe_amd import _lib, api
    if _lib.device_count() > 0:
        pytest.skip("a device is visible: tests/test_gpu_multistart.py covers the native exchange")
    assert mdist.NativeExchange.available() is False
    with pytest.raises(api.OptimalLearningException):
        mdist.NativeExchange(0, 1, 0)
    ex = mdist.make_exchange(mdist.Comm(0, 1, "none", None, None, None))
    assert isinstance(ex, mdist.Exchange) and ex.world == 1
    L = _lib.load()
    assert L.moe_rccl_comm(None, None) == _lib.MOE_ERR_INVALID_VALUE and L.moe_rccl_stats(None, None, None, None) == _lib.MOE_ERR_INVALID_VALUE
    L.moe_rccl_destroy(None)


def test_exchange_three_ranks_in_threads_and_callback_errors():
    """The same with three ranks as threads of this process (ctypes releases the GIL around the library call, the callback takes it
    back): an in-memory all-gather handed to dist.Exchange; and an exception INSIDE the callback comes back as the library's failure
    code and is re-raised by Exchange.reraise -- it never unwinds through the C frames."""
    import threading
    from cornell_moe_amd import _lib
    world = 3
    barrier = threading.Barrier(world)
    slots = [None] * world

    def make(rank):
        def allgather(send):
            slots[rank] = send
            barrier.wait()
            out = np.concatenate(slots)
            barrier.wait()
            return out
        return mdist.Exchange(rank, world, allgather=allgather)

    results = [None] * world

    def run(rank):
        results[rank] = _sharded_items(make(rank), 10, 4, 7.0)

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    want = 7.0 + np.arange(10)[:, None] + np.arange(4)[None, :] / 1000.0
    for rc, out, _ in results:
        assert rc == 0 and np.array_equal(out, want)

    def broken(send):
        raise ValueError("transport down")

    ex = mdist.Exchange(0, 2, allgather=broken)
    with pytest.raises(ValueError, match="transport down"):
        _sharded_items(ex, 4, 1, 0.0)
    one = mdist.Exchange(0, 1)   # world 1: no exchange at all
    rc, out, _ = _sharded_items(one, 3, 2, 1.0)
    assert rc == 0 and np.array_equal(out, 1.0 + np.arange(3)[:, None] + np.arange(2)[None, :] / 1000.0) and one.calls == 0


def test_bench_self_launch_world2(tmp_path):
    """`python bench.py --gpus N` called plainly re-launches itself under torch.distributed.run (bench.self_launch): the
    launcher -- free port on 127.0.0.1, one process per rank, RANK / WORLD_SIZE in the environment, exit code relayed -- driven
    here with a gloo stand-in script that does what bench.py's ranks do around the device work (rendezvous, the all_gather of
    per-restart results, the MAX all_reduce of the elapsed time)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "standin.py"
    out = tmp_path / "out.txt"
    script.write_text('''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from cornell_moe_amd import dist as mdist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ.get("MOE_BENCH_SELF_LAUNCHED") == "1" and os.environ["MASTER_ADDR"] == "127.0.0.1"
dist.init_process_group("gloo", rank=rank, world_size=world)
R = 3
idx = list(range(rank * R, (rank + 1) * R))
kg, grad = mdist.gather_restarts(idx, np.array(idx, dtype=float) + 0.5, np.ones((R, 2, 2)) * (rank + 1), R * world)
t = torch.tensor([float(rank + 1)], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
if rank == 0:
    open(sys.argv[1], "w").write("%%d %%g %%s %%g" %% (world, t.item(), kg.tolist(), grad[R * world - 1, 0, 0]))
dist.destroy_process_group()
''' % root)
    code = subprocess.call([sys.executable, "-c",
                            "import sys; sys.path.insert(0, %r); import bench; sys.exit(bench.self_launch([%r], 2, script=%r))"
                            % (root, str(out), str(script))], timeout=300)
    assert code == 0
    assert out.read_text() == "2 2 [0.5, 1.5, 2.5, 3.5, 4.5, 5.5] 2"


def _bringup_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = str(rank), str(world), str(rank)
    msgs = []
    comm = mdist.bring_up(rank, world, rank, prefer="nccl", timeout_s=120.0, log=msgs.append)
    try:
        # the measurement's collectives on whatever data plane came up
        idx = mdist.shard_restarts(4, rank, world)
        kg, grad = mdist.gather_restarts(idx, [1.0 + i for i in idx], np.ones((len(idx), 1, 2)) * (rank + 1), 4, group=comm.group,
                                         device=comm.device)
        ret[rank] = (comm.backend, comm.rccl_ranks, comm.fallback, kg.tolist(), comm.max_over_ranks(rank + 1.0),
                     comm.gather_floats([rank * 10.0]), msgs)
    finally:
        comm.close()


def test_bring_up_falls_back_to_gloo_when_rccl_preflight_fails():
    """dist.bring_up on a box without GPUs: the RCCL pre-flight children exit non-zero, every rank agrees over the gloo control
    plane, and the collectives of the measurement run on gloo -- loudly (Comm.fallback, the log line)."""
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_bringup_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    for r in range(world):
        backend, rccl, fallback, kg, mx, gathered, msgs = ret[r]
        assert backend == "gloo" and rccl == 0 and "pre-flight" in fallback
        assert kg == [1.0, 2.0, 3.0, 4.0] and mx == 2.0 and gathered == [[0.0], [10.0]]
        assert any("RCCL NOT USED" in m for m in msgs)


def test_bring_up_success_path_under_torch_distributed_run(tmp_path):
    """The SUCCESS path of dist.bring_up the way the driver runs it -- two ranks started by torch.distributed.run (which sets
    TORCHELASTIC_USE_AGENT_STORE=True: the pre-flight children must not look for the launcher's store on their own port), the
    pre-flight children rendezvous among themselves, every rank agrees, the data group is created next to the gloo control group
    and carries the measurement's collectives.  On this box the data plane is gloo (MOE_DIST_DATA_BACKEND, a test hook); the
    steps are the ones RCCL takes."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "standin.py"
    out = tmp_path / "out.txt"
    script.write_text('''
import os, sys
sys.path.insert(0, %r)
import numpy as np
from cornell_moe_amd import dist as mdist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True"
msgs = []
comm = mdist.bring_up(rank, world, int(os.environ["LOCAL_RANK"]), prefer="nccl", timeout_s=120.0, log=msgs.append)
idx = mdist.shard_restarts(4, rank, world)
kg, grad = mdist.gather_restarts(idx, [1.0 + i for i in idx], np.ones((len(idx), 1, 2)) * (rank + 1), 4, group=comm.group, device=comm.device)
mx = comm.max_over_ranks(rank + 1.0)
if rank == 0:
    open(sys.argv[1], "w").write("%%s|%%s|%%s|%%g|%%d" %% (comm.backend, comm.fallback, kg.tolist(), mx, len(msgs)))
comm.close()
''' % root)
    env = dict(os.environ, MOE_DIST_DATA_BACKEND="gloo")
    code = subprocess.call([sys.executable, "-c",
                            "import sys; sys.path.insert(0, %r); import bench; sys.exit(bench.self_launch([%r], 2, script=%r))"
                            % (root, str(out), str(script))], env=env, timeout=600)
    assert code == 0
    assert out.read_text() == "gloo (test hook)|None|[1.0, 2.0, 3.0, 4.0]|2|0"


def test_simplex_domain_start_generation_and_dispatch():
    """r4: the host side of DomainTypes.simplex -- start sets by rejection from a Latin hypercube in the clipped box
    (SimplexIntersectTensorProductDomain::GenerateUniformPointsInDomain, gpp_domain.cpp:179-232, RepeatedDomain's transposition and
    cut to the shortest repeat), the status-key name, and the dispatch's refusal of an unknown domain type."""
    from cornell_moe_amd import GPP, multistart

    class Rng(object):
        def __init__(self):
            self.seed = 100

        def _next_uniform_seed(self):
            self.seed += 1
            return self.seed

    import cornell_moe_amd.api as api_mod
    real = api_mod.latin_hypercube
    api_mod.latin_hypercube = lambda seed, bounds, count: np.random.default_rng(seed).uniform(
        np.asarray(bounds)[0::2], np.asarray(bounds)[1::2], size=(count, len(bounds) // 2))   # (no GPU library on this box)
    try:
        st = multistart._starts(Rng(), np.tile([-0.5, 2.0], 3), 40, 2, 3, domain_type=1)
    finally:
        api_mod.latin_hypercube = real
    assert st.shape[1:] == (2, 3) and 10 <= st.shape[0] <= 40
    flat = st.reshape(-1, 3)
    assert flat.min() >= 0.0 and flat.max() <= 1.0 and np.all(flat.sum(axis=1) <= 1.0 + 1e-12)
    assert np.array_equal(multistart._simplex_box(np.tile([-0.5, 2.0], 3), 3), np.tile([0.0, 1.0], 3))
    # r5: an EMPTY intersection is the reference's BoundsException before any point is drawn (gpp_domain.cpp:107-141) --
    # 'lower left' corner sum >= 1, or an interval emptied by the clip -- not ten rounds of 5 x larger rejection draws
    calls = []
    api_mod.latin_hypercube = lambda seed, bounds, count: calls.append(count)
    try:
        for empty in (np.tile([0.4, 0.9], 3), np.array([0.1, 0.2, 1.5, 2.0, 0.0, 0.3]), np.array([0.5, 0.6, 0.5, 0.6])):
            with pytest.raises(api_mod.BoundsException):
                multistart._starts(Rng(), empty, 200, 2, len(empty) // 2, domain_type=1)
    finally:
        api_mod.latin_hypercube = real
    assert calls == []
    # a sliver that rejects every draw: the loop stops within its memory bound and returns no start (the multistart entry points then
    # refuse num_starts = 0 with "num_multistarts must be > 1", as the reference's drivers do)
    api_mod.latin_hypercube = lambda seed, bounds, count: (calls.append(count), np.full((count, 2), 0.75))[1]
    try:
        none = multistart._starts(Rng(), np.tile([0.0, 1.0], 2), 200, 1, 2, domain_type=1)
    finally:
        api_mod.latin_hypercube = real
    assert none.shape == (0, 1, 2) and max(calls) * 2 <= multistart._MAX_DRAW_DOUBLES

    class P(object):
        domain_type = GPP.DomainTypes.simplex

    assert GPP._domain_name(P()) == "simplex_tensor_product"
    GPP._check_domain_type(P())
    GPP._check_domain_type(P(), kg=True)   # (r4: KG takes the simplex too -- the MC kernels' line search carries the update)

    class Bad(object):
        domain_type = 7

    with pytest.raises(GPP.OptimalLearningException):
        GPP._check_domain_type(Bad(), kg=True)
    assert multistart._gd(type("O", (), {"optimizer_parameters": type("Q", (), dict(
        num_multistarts=3, max_num_steps=4, max_num_restarts=1, num_steps_averaged=0, gamma=0.5, pre_mult=1.0,
        max_relative_change=0.3, tolerance=1e-6))()})(), 1)[8] == 1
