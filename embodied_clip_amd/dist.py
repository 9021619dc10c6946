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

from typing import Tuple

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


def allreduce_flat(flat_grads: torch.Tensor, group=None) -> torch.Tensor:
    """In-place SUM over ranks of the flat gradient bucket (no-op when not initialised / world 1)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group)
    return flat_grads
