"""world_size=2 (gloo, CPU): actors sharded over ranks + one flat-bucket SUM all-reduce of
(local/global)-scaled gradients == the unsharded gradient."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from embodied_clip_amd import synthetic as syn
from embodied_clip_amd.dist import allreduce_flat, gather_actor_counts, global_minibatch_sizes, grad_scale, shard_actors
from oracle import policy as opol
from oracle import ppo as oppo

CFG = dict(in_channels=32, spatial=2, hidden=16)
T, N = 4, 6


def _rendezvous():
    """File-store rendezvous: no TCP port to race for (a freshly probed 'free' port can be taken before bind)."""
    import tempfile
    fd, path = tempfile.mkstemp(prefix="ec_gloo_"); os.close(fd); os.unlink(path)
    return path


def _case(N=N):
    sd = syn.policy_state_dict(3, **CFG)
    g = torch.Generator().manual_seed(0)
    feat = torch.randn(T, N, 32, 2, 2, generator=g).abs()
    goal = syn.synthetic_goals(1, (T, N)); h0 = torch.randn(1, N, 16, generator=g) * 0.3
    masks = syn.synthetic_masks(2, T, N, 0.2)
    actions = torch.randint(0, 6, (T, N), generator=g)
    old_lp = -torch.rand(T, N, 1, generator=g); old_v = torch.randn(T, N, 1, generator=g)
    ret = torch.randn(T, N, 1, generator=g); nadv = torch.randn(T, N, 1, generator=g)
    return sd, feat, goal, h0, masks, actions, old_lp, old_v, ret, nadv


def _grads(sd, feat, goal, h0, masks, actions, old_lp, old_v, ret, nadv, sl):
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    lg, vv, _ = opol.actor_critic_forward(feat[:, sl], goal[:, sl], h0[:, sl], masks[:, sl], leaves)
    total, _ = oppo.ppo_loss(lg, vv, actions[:, sl], old_lp[:, sl], old_v[:, sl], ret[:, sl], nadv[:, sl])
    gs = torch.autograd.grad(total, list(leaves.values()))
    return dict(zip(leaves.keys(), gs))


def _worker(rank, world, port, q, sectioned=False, N=N):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["GLOO_SOCKET_IFNAME"] = "lo"
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    try:
        from embodied_clip_amd.policy import PolicyHandle   # host-only use: flat bucket layout
        torch.set_num_threads(1)
        case = _case(N)
        start, cnt = shard_actors(N, rank, world)
        g = _grads(*case, slice(start, start + cnt))
        h = PolicyHandle(**CFG)
        # the GLOBAL batch size as engine.Worker obtains it: every rank's actor count through one all_gather -- shards need
        # not be equal (N = 5 over 2 ranks: 3 + 2), where a fixed 1 / world scale would weight the two means equally
        counts = gather_actor_counts(cnt, world)
        assert counts == [shard_actors(N, r, world)[1] for r in range(world)] and global_minibatch_sizes(counts, 1) == [N]
        flat = h.flatten(g, "cpu") * grad_scale(T * cnt, T * sum(counts))
        if sectioned:    # engine.Worker(overlap_allreduce=True): GRU + heads section first, then what is left of the bucket
            rec = h.recurrent_section()
            assert 0 < rec.start < rec.stop == h.flat_size
            allreduce_flat(flat[rec])
            allreduce_flat(flat[:rec.start])
        else:
            allreduce_flat(flat)
        if rank == 0:
            q.put(flat)
    finally:
        dist.destroy_process_group()


def test_global_minibatch_sizes_of_uneven_shards():
    """engine.Worker's gradient scale m / nmb_global: the samplers of minibatch range i summed over the ranks."""
    assert global_minibatch_sizes([3, 2], 1) == [5]
    assert global_minibatch_sizes([4, 3, 3], 2) == [2 + 2 + 2, 2 + 1 + 1]      # round(linspace(0, n, 3)) per rank: 4 -> 2|2, 3 -> 2|1
    assert sum(global_minibatch_sizes([64] * 8, 4)) == 512
    assert gather_actor_counts(7, 1) == [7] and gather_actor_counts(7, 3) == [7, 7, 7]   # no process group: equal simulated shards


def test_shard_actors_partition():
    for n, w in [(256, 8), (10, 3), (512, 8), (5, 8)]:
        parts = [shard_actors(n, r, w) for r in range(w)]
        assert sum(c for _, c in parts) == n
        assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(w - 1))


@pytest.mark.timeout(180)
@pytest.mark.parametrize("sectioned,n_actors", [(False, N), (True, N), (False, 5)])
def test_two_rank_flat_allreduce_equals_unsharded(sectioned, n_actors):
    """``sectioned``: the bucket reduced as the overlapped worker does it -- the contiguous GRU + heads section
    (``PolicyHandle.recurrent_section()``) in one call, the goal encoder's section in another (in-place on views of the bucket)."""
    world, port = 2, _rendezvous()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, sectioned, n_actors)) for r in range(world)]
    for p in procs:
        p.start()
    flat = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    from embodied_clip_amd.policy import PolicyHandle
    h = PolicyHandle(**CFG)
    ref = h.flatten(_grads(*_case(n_actors), slice(0, n_actors)), "cpu")
    assert flat.shape == ref.shape == (h.flat_size,)
    assert torch.allclose(flat, ref, rtol=1e-4, atol=1e-7), (flat - ref).abs().max()
