"""Shared case of the a18 (gradient all-reduce) tests: a seeded PPO batch and the ORACLE's gradient of the whole batch."""
import torch

from embodied_clip_amd import synthetic as syn


def make(T: int = 8, N: int = 12):
    g = torch.Generator().manual_seed(1)
    case = dict(
        feat=(torch.randn(T, N, 49, 2048, generator=g).abs() * 0.5).to(torch.bfloat16),
        goal=syn.synthetic_goals(3, (T, N)),
        h0=torch.randn(N, 512, generator=g) * 0.3,
        masks=syn.synthetic_masks(4, T, N, 0.2).reshape(T, N),
        actions=torch.randint(0, 6, (T, N), generator=g),
        old_lp=-torch.rand(T, N, generator=g) - 0.5)
    case["old_v"], case["ret"], case["nadv"] = (torch.randn(T, N, generator=g) for _ in range(3))
    return T, N, case


def oracle_gradient(sd, T: int = 8, N: int = 12):
    """d(PPO total loss)/d(params) of the UNSHARDED batch: torch-CPU autograd through oracle/policy.py + oracle/ppo.py."""
    from oracle import policy as opol
    from oracle import ppo as oppo
    T, N, c = make(T, N)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    feat_nchw = c["feat"].float().view(T, N, 7, 7, 2048).permute(0, 1, 4, 2, 3).contiguous()
    lg, vv, _ = opol.actor_critic_forward(feat_nchw, c["goal"], c["h0"].unsqueeze(0), c["masks"].unsqueeze(-1), leaves)
    u = lambda t: t.unsqueeze(-1)
    total, _ = oppo.ppo_loss(lg, vv, c["actions"], u(c["old_lp"]), u(c["old_v"]), u(c["ret"]), u(c["nadv"]))
    names = list(sd.keys())
    return dict(zip(names, torch.autograd.grad(total, [leaves[k] for k in names])))
