"""Run as a subprocess by tests/test_gpu_policy.py (the library reads EC_POLICY_FAST once per process): one learn pass (forward, PPO
loss, backward) of the HIP policy at a size where the compressor conv over the stored features takes the 8-wave ping-pong kernel
(T*N*49 >= 32,768 rows) against the ORACLE (torch-CPU autograd); prints the forward / per-tensor gradient errors as JSON."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from embodied_clip_amd import synthetic as syn
    from embodied_clip_amd.policy import PolicyHandle
    from embodied_clip_amd.ppo import ppo_loss_raw
    from oracle import policy as opol
    from oracle import ppo as oppo
    T, N = 16, 48
    dev = torch.device("cuda:0")
    h = PolicyHandle()
    sd = syn.policy_state_dict(0)
    flat = h.flatten(sd, dev)
    g = torch.Generator().manual_seed(7)
    feat = (torch.randn(T, N, 49, 2048, generator=g).abs() * 0.5).to(torch.bfloat16)
    goal = syn.synthetic_goals(3, (T, N))
    h0 = torch.randn(N, 512, generator=g) * 0.3
    masks = syn.synthetic_masks(4, T, N, 0.1).reshape(T, N)
    actions = torch.randint(0, 6, (T, N), generator=g)
    old_lp = -torch.rand(T, N, generator=g) - 0.5
    old_v, ret, nadv = (torch.randn(T, N, generator=g) for _ in range(3))
    f = lambda t: t.reshape(-1).contiguous().to(dev)
    rows = feat.reshape(T * N, 49, 2048).contiguous().to(dev)
    ws = torch.empty(h.workspace_bytes(T, N, True), dtype=torch.uint8, device=dev)
    hv, _ = h.forward(flat, rows, f(goal), h0.to(dev), f(masks), T, N, ws)
    dhv, sums = ppo_loss_raw(hv, f(actions), f(old_lp), f(old_v), f(ret), f(nadv), 6)
    gr = torch.zeros_like(flat)
    h.backward(flat, rows, f(masks), T, N, ws, dhv, None, gr)
    torch.cuda.synchronize()
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    feat_nchw = feat.float().view(T, N, 7, 7, 2048).permute(0, 1, 4, 2, 3).contiguous()
    lg, vv, _ = opol.actor_critic_forward(feat_nchw, goal, h0.unsqueeze(0), masks.unsqueeze(-1), leaves)
    u = lambda t: t.unsqueeze(-1)
    total, _ = oppo.ppo_loss(lg, vv, actions, u(old_lp), u(old_v), u(ret), u(nadv))
    names = list(h.offsets.keys())
    ref = dict(zip(names, torch.autograd.grad(total, [leaves[k] for k in names])))
    rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
    hvc = hv.view(T, N, 7).cpu()
    out = {"mode": os.environ.get("EC_POLICY_FAST", "1"), "rows": T * N * 49,
           "logits": rel(hvc[..., :6], lg.detach()), "values": rel(hvc[..., 6:], vv.detach()),
           "grads": {n: rel(gr[o:o + k].cpu(), ref[n].reshape(-1)) for n, (o, k) in h.offsets.items() if k > 0}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
