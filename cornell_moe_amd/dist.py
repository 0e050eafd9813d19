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
    import torch
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
    t = _tensor(buf, device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t, group=group)
    kg = np.zeros(num_restarts)
    grad = np.zeros((num_restarts,) + tuple(local_grad.shape[1:]))
    for o in outs:
        o = o.cpu().numpy()
        for row in o:
            gi = int(row[0])
            if gi >= 0:
                kg[gi] = row[1]
                grad[gi] = row[2:].reshape(grad.shape[1:])
    return kg, grad


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
