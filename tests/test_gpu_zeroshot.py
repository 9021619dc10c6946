"""GPU parity of the zero-shot dual-encoder path (BASELINE config 5; readme_files/zeroshot_objectnav.md:3-8).

The fusion op is builder-defined and PARITY-UNPINNED (the model lives on the unmounted ``zeroshot-objectnav`` branch):
x = normalize(CLIP image embedding) (*) normalize(CLIP text embedding of the goal) -> GRU -> heads
(``oracle/policy.py::zeroshot_actor_critic_forward``).  Tolerances: fp32 policy path forward <= 2e-5, gradients
<= 2e-4 rel-L2 per tensor; image embedding (bf16 trunk + attnpool) vs the fp32 oracle rel-L2 <= 3e-2; text-tower
goal table vs the oracle rel-L2 <= 2e-2.
"""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import make_golden as mg  # noqa: E402
from embodied_clip_amd import synthetic as syn  # noqa: E402
from oracle import clip_resnet as ocr, clip_text as otxt, policy as opol, ppo as oppo  # noqa: E402

pytestmark = pytest.mark.gpu
GZ = torch.load(os.path.join(os.path.dirname(__file__), "golden", "zeroshot_golden.pt"))


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()


@pytest.mark.parametrize("bf16", [False, True])
def test_zeroshot_policy_forward_backward_match_oracle_and_golden(bf16):
    from embodied_clip_amd import ppo
    from embodied_clip_amd.policy import PolicyHandle
    dev = torch.device("cuda:0")
    sd, emb, table, goal, h0, masks, actions, a, b, c, d = mg.zeroshot_case()
    T, N, E = emb.shape
    if bf16:
        emb = emb.to(torch.bfloat16).float()
    h = PolicyHandle(in_channels=E, spatial=1, fusion=1)
    assert h.flat_size >= sum(v.numel() for v in sd.values())
    flat = h.flatten(sd, dev)
    h.set_goal_table(table.to(dev).contiguous())
    rows = emb.reshape(T * N, 1, E).to(dev).to(torch.bfloat16 if bf16 else torch.float32).contiguous()
    ws = torch.empty(h.workspace_bytes(T, N, True), dtype=torch.uint8, device=dev)
    f = lambda t: t.reshape(-1).contiguous().to(dev)
    hv, hf = h.forward(flat, rows, f(goal), h0[0].to(dev).contiguous(), f(masks), T, N, ws)
    lg_ref, vv_ref, h_ref = opol.zeroshot_actor_critic_forward(emb, goal, h0, masks, sd, table)
    hv3 = hv.view(T, N, 7)
    assert _rel(hv3[..., :6], lg_ref) < 2e-5 and _rel(hv3[..., 6:], vv_ref) < 2e-5 and _rel(hf, h_ref[0]) < 2e-5
    if not bf16:   # committed golden (generated from fp32 inputs)
        assert _rel(hv3[..., :6], GZ["logits"]) < 2e-5 and _rel(hf, GZ["h"][0]) < 2e-5
    with torch.no_grad():
        old_lp = opol.categorical_log_prob(lg_ref, actions).unsqueeze(-1) + 0.2 * a
        old_v = vv_ref + 0.2 * b
    names = [k for k, v in sd.items() if v.numel()]
    leaves = {k: (v.clone().requires_grad_(True) if v.numel() else v) for k, v in sd.items()}
    lg2, vv2, _ = opol.zeroshot_actor_critic_forward(emb, goal, h0, masks, leaves, table)
    total, info = oppo.ppo_loss(lg2, vv2, actions, old_lp, old_v, c, d)
    total.backward()
    dhv, sums = ppo.ppo_loss_raw(hv, f(actions), f(old_lp), f(old_v), f(c), f(d), 6)
    grads = torch.zeros_like(flat)
    h.backward(flat, rows, f(masks), T, N, ws, dhv, None, grads)
    torch.cuda.synchronize()
    got = float((sums[0] + 0.5 * sums[1] + 0.01 * sums[2]) / (T * N))
    assert abs(got - info["ppo_total"]) < 1e-5 * max(1.0, abs(info["ppo_total"]))
    gv = h.views(grads)
    for k in names:
        assert _rel(gv[k], leaves[k].grad) < 2e-4, (k, _rel(gv[k], leaves[k].grad))
        if not bf16:
            assert abs(float(gv[k].norm()) - GZ["grad_norms"][k]) < 2e-4 * GZ["grad_norms"][k] + 1e-9, k
    # out-of-range goal ids are clamped, not read out of bounds
    bad = goal.clone(); bad[0, 0] = 99
    h.forward(flat, rows, f(bad), h0[0].to(dev).contiguous(), f(masks), T, N, ws)
    torch.cuda.synchronize()


def test_zeroshot_requires_goal_table_and_unit_spatial():
    from embodied_clip_amd import _lib
    from embodied_clip_amd.policy import PolicyHandle
    dev = torch.device("cuda:0")
    with pytest.raises(_lib.EcError):
        PolicyHandle(in_channels=64, spatial=7, fusion=1)              # spatial must be 1
    h = PolicyHandle(in_channels=64, spatial=1, hidden=32, fusion=1)
    sd = syn.policy_state_dict(0, in_channels=64, spatial=1, hidden=32, fusion=1)
    flat = h.flatten(sd, dev)
    ws = torch.empty(h.workspace_bytes(1, 2, False), dtype=torch.uint8, device=dev)
    z = torch.zeros(2, device=dev)
    with pytest.raises(_lib.EcError):                                   # no goal table set
        h.forward(flat, torch.zeros(2, 1, 64, device=dev), z.long(), torch.zeros(2, 32, device=dev), z + 1, 1, 2, ws, for_backward=False)


def test_config5_zeroshot_worker_iteration_matches_oracle():
    """Worker(zeroshot=True): trunk -> AttentionPool2d embeddings in the rollout buffer, text-tower goal table,
    fused act steps, GAE and optimiser steps vs the CPU oracle (reduced text tower: 2 blocks, vocab 1000)."""
    from embodied_clip_amd.engine import Worker
    T, N, R = 3, 2, 2
    enc_sd = syn.rn50_visual_state_dict(0)
    tsd, tok = mg.text_case()
    pol_sd = syn.policy_state_dict(0, in_channels=1024, spatial=1, fusion=1)
    w = Worker(N, T=T, device="cuda:0", seed=5, update_repeats=R, encoder_sd=enc_sd, policy_sd=pol_sd, zeroshot=True,
               text_sd=tsd, goal_tokens=tok)
    assert (w.S, w.C) == (1, 1024) and w.feat.dtype == torch.float32
    w.collect_rollout()
    w.compute_returns()
    torch.cuda.synchronize()
    # goal table vs the oracle text tower
    tref = otxt.encode_text(tok, tsd, heads=8)
    tref = tref / tref.norm(dim=-1, keepdim=True)
    assert _rel(w.goal_table, tref) < 2e-2
    # image embeddings vs the oracle trunk + attnpool
    frames = w.env.frames.cpu()
    emb_gpu = w.feat.cpu().view(T + 1, N, 1024)
    for t in range(T + 1):
        ref = ocr.attnpool(ocr.clip_resnet_preprocessor(frames[t % frames.shape[0]], enc_sd), enc_sd)
        assert _rel(emb_gpu[t], ref) < 3e-2, (t, _rel(emb_gpu[t], ref))
    # act steps replayed by the oracle policy on the GPU's own embeddings and goal table
    table = w.goal_table.cpu()
    masks, goals, actions = w.env.masks.cpu().unsqueeze(-1), w.env.goals.cpu(), w.actions.cpu()
    h = torch.zeros(1, N, w.H)
    vals, lps = [], []
    with torch.no_grad():
        for t in range(T + 1):
            lg, v, h2 = opol.zeroshot_actor_critic_forward(emb_gpu[t][None], goals[t][None], h, masks[t][None], pol_sd, table)
            vals.append(v[0])
            if t < T:
                lps.append(opol.categorical_log_prob(lg, actions[t][None])[0])
                h = h2
    vals, lps = torch.stack(vals), torch.stack(lps)
    assert _rel(w.values.unsqueeze(-1), vals) < 1e-4
    assert (w.logp.cpu() - lps).abs().max() < 1e-4
    Rr = oppo.compute_returns(w.env.rewards.cpu().unsqueeze(-1), vals, masks)
    _, nadv = oppo.normalized_advantages(Rr, vals)
    assert _rel(w.returns.unsqueeze(-1), Rr) < 1e-4
    # R optimiser steps: oracle autograd + clip + Adam on the trainable tensors
    names = [k for k, v in pol_sd.items() if v.numel()]
    ref_sd = {k: pol_sd[k].clone() for k in names}
    st = {}
    for _ in range(R):
        leaves = {k: ref_sd[k].clone().requires_grad_(True) for k in names}
        lg, vv, _ = opol.zeroshot_actor_critic_forward(emb_gpu[:T], goals[:T], torch.zeros(1, N, w.H), masks[:T], leaves, table)
        total, info = oppo.ppo_loss(lg, vv, actions, w.logp.cpu().unsqueeze(-1), w.values[:T].cpu().unsqueeze(-1),
                                    w.returns[:T].cpu().unsqueeze(-1), w.nadv.cpu().unsqueeze(-1))
        grads = list(torch.autograd.grad(total, [leaves[k] for k in names]))
        info["grad_norm"] = oppo.clip_grad_norm_(grads, 0.5)
        if "step" not in st:
            st.update(step=0, m=[torch.zeros_like(g) for g in grads], v=[torch.zeros_like(g) for g in grads])
        st["step"] += 1
        with torch.no_grad():
            oppo.adam_step([ref_sd[k] for k in names], grads, st["m"], st["v"], st["step"], lr=3e-4)
    w.update()
    torch.cuda.synchronize()
    got = w.loss_info()
    assert abs(got["ppo_total"] - info["ppo_total"]) < 2e-4 * max(1.0, abs(info["ppo_total"]))
    assert abs(got["grad_norm"] - info["grad_norm"]) < 2e-3 * info["grad_norm"]
    pv = w.policy.views(w.params)
    for k in names:
        upd, upd_ref = pv[k].cpu() - pol_sd[k], ref_sd[k] - pol_sd[k]
        assert (upd - upd_ref).abs().max() < 0.15 * R * 3e-4 + 1e-7, k
