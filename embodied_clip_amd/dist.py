"""Data-parallel plumbing: actors shard over GPUs, ONE flat gradient bucket is summed.

Replaces [U] AllenAct ``OnPolicyTrainer.backprop_step`` (engine.py):
``p.grad *= local_bsize/global_bsize; dist.all_reduce(p.grad, async_op=True)`` issued PER PARAMETER
(17 calls, 4 B ... 9.6 MB) and Habitat's DDP buckets.  Here the policy's
parameters/gradients live in one flat fp32 buffer, so each optimiser step is a single 13.9 MB SUM
all-reduce -- on MI355X that is one RCCL ring pass over xGMI (latency-bound, ~0.2 ms at 8 ranks).
The frozen encoder, rollout storage, GRU state and advantage normalisation stay rank-local, exactly
as in the reference.  ``torch.distributed`` (backend "nccl" == RCCL on ROCm, "gloo" in CPU tests)
is the transport; this module holds no kernels.
"""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.distributed as dist


def shard_actors(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard of the global actor (sampler) list owned by ``rank``: (start, count)."""
    base, rem = divmod(n_total, world)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def grad_scale(local_bsize: int, global_bsize: int) -> float:
    """The reference pre-scales local mean-gradients so that a SUM all-reduce yields the global mean."""
    return float(local_bsize) / float(global_bsize)


def _coll_device(group=None, device=None):
    """Where the tensors of a small bookkeeping collective must live: this rank's GPU under RCCL, the host under gloo."""
    if dist.get_backend(group) == "nccl":
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        return dev
    return torch.device("cpu")


def gather_actor_counts(n_local: int, world: int, group=None, device=None) -> List[int]:
    """Every rank's actor (sampler) count, in rank order -- ONE all_gather at worker construction.  The reference scales a
    rank's gradient by ``local_bsize / global_bsize`` ([U] ``OnPolicyTrainer.backprop_step``); with shards of different sizes
    (``shard_actors(10, r, 3)`` = 4, 3, 3) the global size is the SUM of the ranks' sizes, not ``world x local``."""
    if world <= 1 or not (dist.is_available() and dist.is_initialized()):
        return [int(n_local)] * max(int(world), 1)      # (no process group: a simulated shard of `world` equal ones)
    dev = _coll_device(group, device)
    mine = torch.tensor([int(n_local)], dtype=torch.int64, device=dev)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, mine, group=group)
    return [int(t.item()) for t in out]


def minibatch_bounds(n: int, num_mini_batch: int) -> List[int]:
    """[U] allenact ``RolloutStorage.recurrent_generator``: sampler ranges are cut at ``round(linspace(0, n, M + 1))``."""
    return [int(round(i * n / num_mini_batch)) for i in range(num_mini_batch + 1)]


def global_minibatch_sizes(counts: List[int], num_mini_batch: int) -> List[int]:
    """Samplers in minibatch range i summed over the ranks (every rank cuts ITS samplers into the same number of ranges and
    all ranks visit range i at the same optimiser step: one shuffle stream, ``check_job_seed``)."""
    tot = [0] * num_mini_batch
    for n in counts:
        b = minibatch_bounds(n, num_mini_batch)
        for i in range(num_mini_batch):
            tot[i] += b[i + 1] - b[i]
    return tot


def collective_active(world: int, force: bool = False) -> bool:
    """Will ``allreduce_flat`` issue a collective?  (a process group is up, and there is more than one rank or ``force``)"""
    return (world > 1 or force) and dist.is_available() and dist.is_initialized()


def allreduce_flat(flat_grads: torch.Tensor, group=None, force: bool = False) -> torch.Tensor:
    """In-place SUM over ranks of the flat gradient bucket (no-op when not initialised / world 1).
    ``force``: issue the collective also at world size 1 (an identity, but RCCL's communicator set-up and its kernel
    run -- the first-contact check of the N > 1 path on a 1-GPU box)."""
    if dist.is_available() and dist.is_initialized() and (force or dist.get_world_size(group) > 1):
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group)
    return flat_grads


def check_job_seed(seed: int, world: int, group=None, device=None) -> None:
    """The minibatch shuffle stream must be IDENTICAL on every rank (every rank visits the same sampler range at the same
    optimiser step; only then is the SUM all-reduce with the fixed 1/world scale the global minibatch mean when
    N % num_mini_batch != 0).  With a process group up, the ranks compare their seeds (MIN == MAX) and a launcher that
    passed per-rank seeds fails here instead of training on silently inconsistent gradients."""
    if world <= 1 or not (dist.is_available() and dist.is_initialized()):
        return
    # (the tensors of an RCCL collective must live on THIS rank's GPU -- a launcher may pass device="cuda:k" without
    #  torch.cuda.set_device -- and on the host under gloo)
    dev = _coll_device(group, device)
    lo = torch.tensor([float(seed)], dtype=torch.float64, device=dev)
    hi = lo.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    if float(lo.item()) != float(hi.item()):
        raise ValueError(f"Worker(seed=...) must be the JOB seed, identical on every rank (got {int(lo.item())}..{int(hi.item())}); "
                         "per-rank randomness is derived from it internally (seed + 7919 * rank)")
