"""Multi-GPU sharding of the q-KG hot path: one process per GPU, `torch.distributed` (backend "nccl" == RCCL over xGMI
on ROCm; "gloo" in the CPU tests).

The path shards on two independent axes:
  * multistart restarts (independent points_to_sample sets; what the reference runs as OpenMP iterations,
    gpp_optimization.hpp:1472-1546): rank r evaluates restarts r, r+W, ...; ONE all_gather of R x (1 + q*d) doubles makes
    every rank see every (KG, grad KG) -- the merge the reference does under `omp critical` (:1537-1545);
  * MC samples of one evaluation: contiguous EVEN-ALIGNED slices (antithetic pairs 2j/2j+1 stay together,
    gpp_knowledge_gradient_optimization.cpp:171-180); ONE all_reduce(SUM) of 1 + q*d doubles (264 B at the headline
    shape) -- latency-bound, so it is a single fused collective per evaluation.
  * GP index of an MCMC-averaged evaluation (num_mcmc GPs over the same data): rank r builds and
    evaluates members r, r+W, ...; ONE all_reduce(SUM) of E x (1 + q*d) doubles of plain per-GP sums, then the mean /
    fidelity-cost step (moe_kg_mcmc_finalize) on every rank.
torch is used for the process group and the collective only; compute goes through the C ABI.
"""
import numpy as np


def shard_samples(num_mc, rank, world):
    """Contiguous even-aligned slice [first, first+count) of the MC samples for `rank` (count may be 0 for tiny num_mc)."""
    pairs = (num_mc + 1) // 2
    base, rem = divmod(pairs, world)
    p0 = rank * base + min(rank, rem)
    p1 = p0 + base + (1 if rank < rem else 0)
    first = 2 * p0
    last = min(2 * p1, num_mc)
    return first, max(last - first, 0)


def shard_restarts(num_restarts, rank, world):
    """Indices of the restarts `rank` owns (round-robin, like omp schedule(static,1))."""
    return list(range(rank, num_restarts, world))


def _tensor(buf, device):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(buf, dtype=np.float64))
    return t.to(device) if device is not None else t


def allreduce_kg(kg_sum, grad_sum, num_mc, group=None, device=None):
    """Sum the un-normalised shard results over ranks and normalise: returns (KG, grad KG[q,d])."""
    import torch.distributed as dist
    grad_sum = np.asarray(grad_sum, dtype=np.float64)
    buf = np.concatenate([[float(kg_sum)], grad_sum.ravel()])
    t = _tensor(buf, device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    out = t.cpu().numpy()
    return out[0] / num_mc, out[1:].reshape(grad_sum.shape) / num_mc


def kg_grad_mc_sharded(eval_shard, num_mc, rank, world, group=None, device=None):
    """One KG-gradient evaluation with its MC samples sharded over `world` ranks.

    eval_shard(first_sample, num_local) -> (kg_sum, grad_sum[q,d]) is the C-ABI call (DeviceGP.kg(...first_sample=,
    num_local=...)) returning UN-normalised sums; ranks with an empty slice contribute zeros (shape from `grad_shape`)."""
    first, count = shard_samples(num_mc, rank, world)
    kg_sum, grad_sum = eval_shard(first, count)
    return allreduce_kg(kg_sum, grad_sum, num_mc, group=group, device=device)


def gather_restarts(local_idx, local_kg, local_grad, num_restarts, group=None, device=None):
    """all_gather the per-restart (KG, grad KG) of every rank; returns arrays ordered by global restart index.

    local_idx: global indices this rank evaluated (from shard_restarts); local_kg[len], local_grad[len, q, d]."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    local_grad = np.asarray(local_grad, dtype=np.float64)
    qd = int(np.prod(local_grad.shape[1:])) if local_grad.ndim > 1 else 0
    per = (num_restarts + world - 1) // world  # padded rows per rank so the gather is a single fixed-size collective
    buf = np.zeros((per, 2 + qd))
    buf[:, 0] = -1.0
    for row, (gi, kgv) in enumerate(zip(local_idx, local_kg)):
        buf[row, 0] = gi
        buf[row, 1] = kgv
        buf[row, 2:] = local_grad[row].ravel()
    rows = _allgather_flat(buf.ravel(), world, group, device).reshape(world * per, 2 + qd)   # ONE collective, ONE copy back
    kg = np.zeros(num_restarts)
    grad = np.zeros((num_restarts,) + tuple(local_grad.shape[1:]))
    for row in rows:
        gi = int(row[0])
        if gi >= 0:
            kg[gi] = row[1]
            grad[gi] = row[2:].reshape(grad.shape[1:])
    return kg, grad


_NO_INTO_TENSOR = set()   # (backend names whose all_gather_into_tensor raised: they take the list form from then on)


def _allgather_flat(send, world, group, device, bufs=None):
    """All-gather of a flat float64 array over `group`: recv[world * n] in rank order.  One collective into ONE tensor
    (all_gather_into_tensor) and -- with the buffers on a device -- ONE device->host copy (r6; until r5 a list all_gather followed by
    a .cpu() per rank: `world` synchronising copies per exchange).  `bufs`: a dict the caller keeps, so that repeated exchanges of
    one size reuse their tensors."""
    import torch
    import torch.distributed as dist
    n = int(send.size)
    key = (n, str(device))
    if bufs is not None and key in bufs:
        tin, tout = bufs[key]
    else:
        tin = torch.empty(n, dtype=torch.float64, device=device if device is not None else "cpu")
        tout = torch.empty(world * n, dtype=torch.float64, device=tin.device)
        if bufs is not None:
            if len(bufs) > 16:
                bufs.clear()
            bufs[key] = (tin, tout)
    tin.copy_(torch.from_numpy(np.ascontiguousarray(send, dtype=np.float64)))
    backend = dist.get_backend(group)
    if backend not in _NO_INTO_TENSOR:
        try:
            dist.all_gather_into_tensor(tout, tin, group=group)
            return tout.cpu().numpy() if tout.is_cuda else tout.numpy().copy()
        except (RuntimeError, NotImplementedError):
            _NO_INTO_TENSOR.add(backend)
    outs = list(tout.view(world, n).unbind(0))   # (views of the one receive tensor: still a single copy back)
    dist.all_gather(outs, tin, group=group)
    return tout.cpu().numpy() if tout.is_cuda else tout.numpy().copy()


class Exchange(object):
    """moe_comm_t for the multi-rank outer optimisers (r5; include/moe_hip.h: moe_kg_multistart_comm, moe_kg_mcmc_multistart_comm):
    the library's all-gather of `count` doubles per rank, carried by torch.distributed -- `group` / `device` as Comm hands them out
    (backend nccl = RCCL over xGMI with the buffers on `device`, gloo with host tensors).  Counts exchanges and bytes for the
    benchmark lines.  An exception raised inside the callback cannot unwind through the C frames: it is stored, the library gets a
    failure code, and the caller re-raises it (reraise)."""

    def __init__(self, rank, world, group=None, device=None, allgather=None):
        from . import _lib
        import ctypes as C
        self.rank, self.world, self.group, self.device = int(rank), int(world), group, device
        self.calls, self.doubles, self.seconds = 0, 0, 0.0
        self._exc = None
        self._bufs = {}
        self._impl = allgather or self._torch_allgather
        self._cb = _lib.ALLGATHER_FN(self._callback)  # (kept alive with the object)
        self.c_struct = _lib.Comm(self.rank, self.world, self._cb, None)
        self._C = C

    @classmethod
    def from_comm(cls, comm):
        """From dist.bring_up()'s Comm (world 1: an Exchange that is never called)."""
        return cls(comm.rank, comm.world, comm.group, comm.device)

    def _torch_allgather(self, send):
        # one collective into one preallocated tensor, one copy back (r6; `world` .cpu() calls per exchange before)
        return _allgather_flat(send, self.world, self.group, self.device, self._bufs)

    def _callback(self, ctx, send, recv, count):
        import time
        try:
            t0 = time.perf_counter()
            a = np.ctypeslib.as_array(send, shape=(count,)).copy()
            out = np.asarray(self._impl(a), dtype=np.float64).reshape(self.world * count)
            np.ctypeslib.as_array(recv, shape=(self.world * count,))[:] = out
            self.calls += 1
            self.doubles += count
            self.seconds += time.perf_counter() - t0
            return 0
        except BaseException as e:  # noqa: BLE001 -- nothing may propagate into the C caller
            self._exc = e
            return 1

    def reraise(self):
        if self._exc is not None:
            e, self._exc = self._exc, None
            raise e


class NativeExchange(object):
    """moe_comm_t carried by the library's OWN RCCL communicator (r6; include/moe_hip.h: moe_rccl_*, csrc/rccl_comm.hip): the
    all-gather of the multi-rank optimisers is ncclAllGather on a stream of the library -- no Python callback inside the optimiser
    loop.  The 128-byte unique id is made on rank 0 and broadcast over the DEFAULT process group (gloo: the control plane of
    bring_up); construction is collective over the ranks.  Same attributes as Exchange (c_struct, calls, doubles, seconds, reraise)."""

    def __init__(self, rank, world, device_index, broadcast=None):
        import ctypes as C
        from . import _lib
        L = _lib.load()
        self.rank, self.world = int(rank), int(world)
        err = _lib.MoeError()
        ident = C.create_string_buffer(128)
        if self.rank == 0:
            _check_rc(L.moe_rccl_unique_id(ident, C.byref(err)), err)
        raw = bytes(ident.raw)
        if self.world > 1:
            if broadcast is None:
                import torch.distributed as dist
                box = [raw if self.rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                raw = box[0]
            else:
                raw = broadcast(raw)
        self._h = C.c_void_p(None)
        _check_rc(L.moe_rccl_create(raw, self.rank, self.world, int(device_index), C.byref(self._h), C.byref(err)), err)
        self.c_struct = _lib.Comm()
        L.moe_rccl_comm(self._h, C.byref(self.c_struct))
        self._L, self._C = L, C
        self._base = (0, 0, 0.0)

    @staticmethod
    def available():
        """True where a device is visible, librccl can be loaded and hands out a unique id (no communicator is created)."""
        import ctypes as C
        from . import _lib
        if _lib.device_count() <= 0:
            return False
        err = _lib.MoeError()
        return _lib.load().moe_rccl_unique_id(C.create_string_buffer(128), C.byref(err)) == 0

    def _stats(self):
        C = self._C
        calls, nbytes, sec = C.c_longlong(0), C.c_longlong(0), C.c_double(0.0)
        self._L.moe_rccl_stats(self._h, C.byref(calls), C.byref(nbytes), C.byref(sec))
        return calls.value, nbytes.value, sec.value

    calls = property(lambda self: self._stats()[0] - self._base[0], lambda self, v: self._reset())
    doubles = property(lambda self: (self._stats()[1] - self._base[1]) // (8 * max(self.world, 1)), lambda self, v: None)
    seconds = property(lambda self: self._stats()[2] - self._base[2], lambda self, v: None)

    def _reset(self):
        self._base = self._stats()

    def allreduce_sum(self, values):
        """In-place sum over the ranks of a float64 array (the MC-sharded evaluation's collective)."""
        from . import _lib
        a = np.ascontiguousarray(values, dtype=np.float64)
        err = _lib.MoeError()
        _check_rc(self._L.moe_rccl_allreduce_sum(self._h, a.ctypes.data_as(_lib.dp), int(a.size), self._C.byref(err)), err)
        return a

    def allgather(self, send):
        """The exchange itself, callable from Python (tests): recv[world][count]."""
        from . import _lib
        a = np.ascontiguousarray(send, dtype=np.float64)
        out = np.zeros(self.world * a.size)
        rc = self.c_struct.allgather(self.c_struct.ctx, a.ctypes.data_as(_lib.dp), out.ctypes.data_as(_lib.dp), int(a.size))
        if rc != 0:
            raise RuntimeError("native RCCL all-gather failed")
        return out.reshape(self.world, a.size)

    def reraise(self):
        pass

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.moe_rccl_destroy(self._h)
            self._h = self._C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:  # pragma: no cover
            pass


def _check_rc(rc, err):
    if rc != 0:
        from .api import OptimalLearningException
        raise OptimalLearningException(err.message.decode("utf-8", "replace") or "native RCCL exchange failed (code %d)" % rc)


def make_exchange(comm, local_rank=0, native=None):
    """The exchange of the multi-rank optimisers for dist.bring_up()'s Comm: the library's native RCCL communicator where the data
    plane is RCCL (MOE_NATIVE_RCCL=0, or native=False: torch.distributed's), torch.distributed otherwise (gloo)."""
    import os
    if native is None:
        native = os.environ.get("MOE_NATIVE_RCCL", "1") != "0"
    if native and comm.world > 1 and comm.backend == "nccl":
        # construction is collective: the ranks first agree (control plane) that every one of them can load librccl
        import torch
        import torch.distributed as dist
        flag = torch.tensor([1.0 if NativeExchange.available() else 0.0], dtype=torch.float64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if float(flag.item()) >= 0.5:
            return NativeExchange(comm.rank, comm.world, local_rank)
    return Exchange.from_comm(comm)


def shard_members(num_mcmc, rank, world):
    """GP indices of an MCMC ensemble that `rank` builds and evaluates (round-robin)."""
    return list(range(rank, num_mcmc, world))


def kg_mcmc_sharded(local_sums, finalize, group=None, device=None):
    """One batch of MCMC-averaged KG evaluations with the GP index sharded over the ranks.

    local_sums() -> (kg_sum [E], grad_sum [E][q][d]): DeviceGPMCMC(members=shard_members(...)).kg_batch(..., finalize=False)
    (zeros when this rank owns no member); finalize(kg_sum, grad_sum) -> (kg [E], grad [E][q][d]) is
    DeviceGPMCMC.kg_finalize (mean over ALL members, fidelity cost and its gradient term)."""
    import torch.distributed as dist
    kg_sum, grad_sum = local_sums()
    kg_sum = np.asarray(kg_sum, dtype=np.float64)
    grad_sum = np.asarray(grad_sum, dtype=np.float64)
    t = _tensor(np.concatenate([kg_sum.ravel(), grad_sum.ravel()]), device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    out = t.cpu().numpy()
    return finalize(out[:kg_sum.size].reshape(kg_sum.shape), out[kg_sum.size:].reshape(grad_sum.shape))


# ---------------------------------------------------------------------------------------------------------------------
# Process-group bring-up with a pre-flight (round 3).  The data plane is RCCL (`nccl`), one rank per GPU; but a first contact
# with a multi-GPU node must not be able to hang or kill the job, so:
#   1. the DEFAULT group is gloo (TCP on 127.0.0.1): the control plane -- agreement between ranks, timing reductions on host
#      tensors -- never depends on RCCL;
#   2. RCCL is tried first in a throw-away CHILD process per rank (`python -m cornell_moe_amd.dist --preflight`, its own
#      rendezvous port): init_process_group("nccl", device_id=...), one all_reduce of a double, checked, exit 0.  The parent
#      waits at most `timeout_s` and kills exactly the child it started; a hang or crash inside RCCL stays in the child;
#   3. the ranks agree (gloo all_reduce MIN) on whether every child passed; only then the parent creates its own nccl group.
#      Otherwise the collectives of the measurement run on the gloo group and the caller is told why (`Comm.fallback`).
# ---------------------------------------------------------------------------------------------------------------------
class Comm(object):
    """What bring_up() hands back: `group` carries the data-plane collectives, `device` is where their buffers live (a cuda
    device for nccl, None = host tensors for gloo), `backend` in {"nccl", "gloo", "none"}, `fallback` = None or the reason the
    preferred backend was not used."""

    def __init__(self, rank, world, backend, group, device, fallback, preflight_s=0.0):
        self.rank, self.world, self.backend, self.group, self.device, self.fallback = rank, world, backend, group, device, fallback
        self.preflight_s = preflight_s

    @property
    def rccl_ranks(self):
        return self.world if (self.world > 1 and self.backend == "nccl") else 0

    def barrier(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()  # control plane (gloo): never an RCCL call outside the measured collectives

    def max_over_ranks(self, x):
        if self.world == 1:
            return float(x)
        import torch
        import torch.distributed as dist
        t = torch.tensor([float(x)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def gather_floats(self, values):
        """all_gather of a few host doubles per rank over the control plane: returns a [world][len] list."""
        if self.world == 1:
            return [list(map(float, values))]
        import torch
        import torch.distributed as dist
        t = torch.tensor(list(map(float, values)), dtype=torch.float64)
        outs = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(outs, t)
        return [o.tolist() for o in outs]

    def close(self):
        if self.world > 1:
            import torch.distributed as dist
            try:
                dist.barrier()
                dist.destroy_process_group()
            except Exception:  # pragma: no cover
                pass


def _preflight_child():
    """The body of the throw-away child: RCCL bring-up + one checked all_reduce.  Exit code 0 = RCCL works for this rank.
    (MOE_DIST_DATA_BACKEND=gloo -- a test hook -- runs the same steps on gloo / host tensors, so that the whole success path of
    bring_up can be exercised on a box without GPUs.)"""
    import datetime
    import os
    import sys
    import torch
    import torch.distributed as dist
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("MOE_DIST_DATA_BACKEND", "nccl")
    if backend == "nccl":
        if not torch.cuda.is_available():
            sys.exit(3)
        if torch.cuda.device_count() == 1 and os.environ.get("MOE_BENCH_SHARE_GPU") != "1":
            local = 0  # (a launcher that narrows every rank's visibility to its own GPU: see bench.py)
        if torch.cuda.device_count() <= local:
            sys.exit(3)
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=60))
    else:
        dev = torch.device("cpu")
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    t = torch.full((1,), float(rank + 1), dtype=torch.float64, device=dev)
    dist.all_reduce(t)
    if backend == "nccl":
        torch.cuda.synchronize()
    ok = abs(float(t.item()) - world * (world + 1) / 2.0) < 1e-12
    dist.destroy_process_group()
    sys.exit(0 if ok else 4)


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def preflight_env(port):
    """Environment of the pre-flight child: this rank's RANK / LOCAL_RANK / WORLD_SIZE, its OWN rendezvous port, and none of the
    launcher's TORCHELASTIC_* variables -- with TORCHELASTIC_USE_AGENT_STORE=True (what torch.distributed.run sets) env://
    initialisation connects to the launcher's store at MASTER_PORT as a client instead of starting one, and the children, whose
    port nobody serves, would wait for it until they time out."""
    import os
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    env["MASTER_ADDR"] = os.environ.get("MASTER_ADDR", "127.0.0.1")
    env["MASTER_PORT"] = str(port)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def bring_up(rank, world, local_rank, prefer="nccl", timeout_s=150.0, log=None):
    """Default (control) group on gloo, data group on `prefer` when its pre-flight passes on EVERY rank; see the block comment."""
    import datetime
    import os
    import subprocess
    import sys
    import time
    log = log or (lambda m: None)
    if world <= 1:
        return Comm(rank, 1, "none", None, None, None)
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if os.environ.get("MOE_DIST_FAIL") == "rendezvous":  # test hook: what the caller does when no process group comes up
        raise RuntimeError("MOE_DIST_FAIL=rendezvous")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=600))
    if prefer != "nccl":
        return Comm(rank, world, "gloo", None, None, None)
    t0 = time.time()
    port = [_free_port() if rank == 0 else 0]
    dist.broadcast_object_list(port, src=0)
    child = subprocess.Popen([sys.executable, "-m", "cornell_moe_amd.dist", "--preflight"], env=preflight_env(port[0]),
                             cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), stdout=subprocess.DEVNULL,
                             stderr=subprocess.PIPE, universal_newlines=True)
    why = None
    try:
        _, err = child.communicate(timeout=timeout_s)
        if child.returncode != 0:
            tail = (err or "").strip().splitlines()
            why = "rccl pre-flight failed on rank %d (exit %d): %s" % (rank, child.returncode, tail[-1] if tail else "no message")
    except subprocess.TimeoutExpired:
        child.kill()  # exactly the process started above
        child.communicate()
        why = "rccl pre-flight timed out after %.0f s on rank %d" % (timeout_s, rank)
    flag = torch.tensor([0.0 if why else 1.0], dtype=torch.float64)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    whys = [None] * world
    dist.all_gather_object(whys, why)
    took = time.time() - t0
    if float(flag.item()) < 0.5:
        reason = next((w for w in whys if w), "rccl pre-flight failed")
        log("RCCL NOT USED -- %s; collectives run on gloo (host tensors)" % reason)
        return Comm(rank, world, "gloo", None, None, reason, took)
    data_backend = os.environ.get("MOE_DIST_DATA_BACKEND", "nccl")  # (test hook, see _preflight_child)
    if data_backend != "nccl":
        grp = dist.new_group(backend=data_backend, timeout=datetime.timedelta(seconds=120))
        t = torch.ones(1, dtype=torch.float64)
        dist.all_reduce(t, group=grp)
        assert abs(float(t.item()) - world) < 1e-12
        return Comm(rank, world, data_backend + " (test hook)", grp, None, None, took)
    dev = torch.device("cuda", local_rank)
    # The pre-flight child isolated the FIRST contact with RCCL; the group created here, in the parent, is protected by its 120 s
    # timeout only.  new_group is collective over the default group, so the ranks must make the same sequence of calls: after
    # each attempt they agree over gloo (MIN) whether it succeeded EVERYWHERE, and only then does anyone move on to the attempt
    # without device_id -- a rank whose attempt raised alone must not retry alone.
    grp, err, all_ok = None, None, False
    for kwargs in (dict(device_id=dev), dict()):
        mine = None
        try:
            mine = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=120), **kwargs)
        except Exception as e:  # noqa: BLE001
            err = "%s: %s" % (type(e).__name__, e)
        flag = torch.tensor([1.0 if mine is not None else 0.0], dtype=torch.float64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if float(flag.item()) >= 0.5:
            grp, all_ok = mine, True
            break
        if mine is not None:  # it came up here but not everywhere: drop it, everybody takes the next attempt
            try:
                dist.destroy_process_group(mine)
            except Exception:  # noqa: BLE001  # pragma: no cover
                pass
    if not all_ok:
        reason = "rccl group creation failed after a passing pre-flight (%s)" % err
        log("RCCL NOT USED -- %s; collectives run on gloo (host tensors)" % reason)
        return Comm(rank, world, "gloo", None, None, reason, took)
    t = torch.ones(1, dtype=torch.float64, device=dev)
    dist.all_reduce(t, group=grp)
    torch.cuda.synchronize()
    assert abs(float(t.item()) - world) < 1e-12
    return Comm(rank, world, "nccl", grp, dev, None, took)


if __name__ == "__main__":
    import sys
    if "--preflight" in sys.argv:
        _preflight_child()
