"""Policy / PPO leg of ``__graft_entry__.smoke()``: one act step and one optimiser step of the HIP path on the
features the encoder just produced, checked against the CPU oracle (the oracle is imported here ONLY because
smoke() is one of the three places allowed to use it as a checker)."""
from __future__ import annotations

import torch


def run(dev, feat_bf16_nhwc: torch.Tensor) -> None:
    from . import ppo, synthetic as syn
    from .policy import PolicyHandle
    from oracle import policy as opol
    from oracle import ppo as oppo

    N = feat_bf16_nhwc.shape[0]
    T = 1
    sd = syn.policy_state_dict(0)
    h = PolicyHandle()
    flat = h.flatten(sd, dev)
    rows = feat_bf16_nhwc.reshape(N, 49, 2048).contiguous()
    goal = syn.synthetic_goals(5, (T, N))
    h0 = torch.zeros(1, N, 512)
    masks = torch.ones(T, N, 1)
    ws = torch.empty(h.workspace_bytes(T, N, True), dtype=torch.uint8, device=dev)
    hv, hf = h.forward(flat, rows, goal.reshape(-1).to(dev), h0[0].to(dev), masks.reshape(-1).to(dev), T, N, ws)
    feat_ref = rows.float().cpu().view(T, N, 7, 7, 2048).permute(0, 1, 4, 2, 3).contiguous()
    lg, vv, hT = opol.actor_critic_forward(feat_ref, goal, h0, masks, sd)
    err = (hv.view(T, N, 7)[..., :6].cpu() - lg).abs().max().item()
    assert err < 1e-3 * max(1.0, lg.abs().max().item()), f"policy act parity: {err}"
    actions = torch.zeros(T, N, dtype=torch.int64)
    old_lp = opol.categorical_log_prob(lg, actions).unsqueeze(-1).detach()
    ret, adv = vv.detach() + 0.5, torch.ones(T, N, 1)
    total, info = oppo.ppo_loss(lg, vv, actions, old_lp, vv.detach(), ret, adv)
    f = lambda t: t.reshape(-1).contiguous().to(dev)
    dhv, sums = ppo.ppo_loss_raw(hv, f(actions), f(old_lp), f(vv.detach()), f(ret), f(adv), 6)
    grads = torch.zeros_like(flat)
    h.backward(flat, rows, masks.reshape(-1).to(dev), T, N, ws, dhv, None, grads)
    ppo.FlatAdam(flat).step(grads)
    torch.cuda.synchronize()
    got = float((sums[0] + 0.5 * sums[1] + 0.01 * sums[2]) / (T * N))
    assert abs(got - info["ppo_total"]) < 1e-4 * max(1.0, abs(info["ppo_total"])), (got, info["ppo_total"])
    assert torch.isfinite(flat).all()
    print(f"smoke: policy act max|dlogit| = {err:.2e}, PPO loss {got:.6f} (oracle {info['ppo_total']:.6f}), Adam step ok")
