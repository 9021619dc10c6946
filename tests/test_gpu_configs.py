"""BASELINE.json configs 2 and 3 exercised as WORKLOADS on the GPU (VERDICT r01 "configs_untested").

  * config 3: ViT-B/32 encoder tokens -> drop CLS -> 7x7x768 channels-last map -> the same GRU actor-critic
    (``Worker(encoder="vit")``; the fusion is builder-defined and parity-unpinned, SURVEY.md §8d note):
    small-size vs the oracle (``oracle/clip_vit.py`` -> ``oracle/policy.py`` with ``in_channels=768``) and a
    256 x 128 full-size property test;
  * config 2: 64 actors x rollout 128 (two 32-frame encoder slices: a launch geometry no other test uses).

Tolerances: encoder bf16 path vs the fp32 oracle rel-L2 <= 3e-2 (ViT) / 2e-2 (RN50); policy / PPO (fp32) as
in tests/test_gpu_engine.py.
"""
import pytest
import torch

from embodied_clip_amd import synthetic as syn
from oracle import clip_vit as ovit
from oracle import policy as opol
from oracle import ppo as oppo

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()


def test_config3_vit_worker_iteration_matches_oracle():
    """Worker(encoder='vit'): encoder tokens, act steps, GAE and R optimiser steps vs the CPU oracle."""
    from embodied_clip_amd.engine import Worker
    T, N, R = 3, 2, 2
    enc_sd = syn.vit_visual_state_dict(0)
    pol_sd = syn.policy_state_dict(0, in_channels=768, spatial=7)
    w = Worker(N, T=T, device="cuda:0", seed=5, update_repeats=R, encoder="vit", encoder_sd=enc_sd, policy_sd=pol_sd)
    assert (w.S, w.C) == (7, 768)
    w.collect_rollout()
    w.compute_returns()
    torch.cuda.synchronize()
    frames = w.env.frames.cpu()
    # stored rows are channels-last patch tokens [T+1, N, 49, 768]; the oracle's NCHW view of the same map
    feat_gpu = w.feat.float().cpu().view(T + 1, N, 7, 7, 768).permute(0, 1, 4, 2, 3).contiguous()
    for t in range(T + 1):
        tok = ovit.clip_vit_preprocessor(frames[t % frames.shape[0]], enc_sd)          # [N, 50, 768]
        ref = tok[:, 1:, :].reshape(N, 7, 7, 768).permute(0, 3, 1, 2)                   # drop CLS
        assert _rel(feat_gpu[t], ref) < 3e-2, (t, _rel(feat_gpu[t], ref))
    masks, goals, actions = w.env.masks.cpu().unsqueeze(-1), w.env.goals.cpu(), w.actions.cpu()
    h = torch.zeros(1, N, w.H)
    vals, lps = [], []
    with torch.no_grad():
        for t in range(T + 1):
            lg, v, h2 = opol.actor_critic_forward(feat_gpu[t][None], goals[t][None], h, masks[t][None], pol_sd)
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
    assert _rel(w.nadv.unsqueeze(-1), nadv) < 1e-3
    sd_ref = {k: v.clone() for k, v in pol_sd.items()}
    batch = dict(feat=feat_gpu[:T], goal=goals[:T], h0=torch.zeros(1, N, w.H), masks=masks[:T], actions=actions,
                 old_log_probs=w.logp.cpu().unsqueeze(-1), old_values=w.values[:T].cpu().unsqueeze(-1),
                 returns=w.returns[:T].cpu().unsqueeze(-1), norm_adv=w.nadv.cpu().unsqueeze(-1))
    st = {}
    for _ in range(R):
        info, _ = oppo.ppo_update_step(sd_ref, batch, st)
    w.update()
    torch.cuda.synchronize()
    got = w.loss_info()
    assert abs(got["ppo_total"] - info["ppo_total"]) < 2e-4 * max(1.0, abs(info["ppo_total"]))
    assert abs(got["grad_norm"] - info["grad_norm"]) < 2e-3 * info["grad_norm"]
    pv = w.policy.views(w.params)
    for name, pref in sd_ref.items():
        upd, upd_ref = pv[name].cpu() - pol_sd[name], pref - pol_sd[name]
        assert (upd - upd_ref).abs().max() < 0.15 * R * 3e-4 + 1e-7, (name, (upd - upd_ref).abs().max())


def _check_properties(w):
    """Size-independent properties of one rollout + GAE + a single-epoch update (see tests/test_gpu_fullsize.py)."""
    T, N = w.T, w.N
    for sl in w.slices:
        assert torch.equal(sl.feat[1], sl.feat[5]) and torch.equal(sl.feat[2], sl.feat[T - 2])   # pool of 4 frame batches
        assert not torch.equal(sl.feat[1], sl.feat[2])
    assert torch.isfinite(w.feat.float()).all()
    assert w.actions.min() >= 0 and w.actions.max() < w.A
    assert torch.isfinite(w.logp).all() and (w.logp <= 0).all()
    assert torch.bincount(w.actions.reshape(-1), minlength=w.A).min().item() > T * N // 20
    adv = w.returns[:T] - w.values[:T]
    assert torch.allclose(adv, w.adv, atol=1e-5)
    assert abs(w.nadv.mean().item()) < 1e-4 and abs(w.nadv.std(unbiased=True).item() - 1.0) < 1e-3
    idx = (w.env.masks[1:T + 1] == 0)
    assert torch.allclose(w.returns[:T][idx], w.env.rewards[idx], atol=1e-5)
    p0 = w.params.clone()
    w.update_repeats, saved = 1, w.update_repeats
    try:
        w.update()
        torch.cuda.synchronize()
    finally:
        w.update_repeats = saved
    info = w.loss_info()
    assert abs(info["ratio"] - 1.0) < 1e-4, info
    assert 0.0 < info["grad_norm"] < 1e3
    step = (w.params - p0).abs().max().item()
    assert 0.0 < step <= 3e-4 * 1.001
    for sl in w.slices:
        g = sl.dhv[:, :w.A]
        assert g.sum(dim=1).abs().max().item() < 1e-6 + 1e-4 * g.abs().max().item()


def test_config3_vit_fullsize_properties():
    """256 actors x rollout 128 with the ViT-B/32 encoder (BASELINE config 3's size)."""
    from embodied_clip_amd.engine import Worker
    w = Worker(256, T=128, device="cuda:0", seed=0, encoder="vit")
    w.collect_rollout()
    w.compute_returns()
    torch.cuda.synchronize()
    assert w.feat.shape == (129, 256, 49, 768)
    # a frame encoded inside the 128-frame launch equals the same frame encoded alone (best pool match)
    sl = w.slices[1]
    b = sl.feat[1][11].float().reshape(-1)
    rels = []
    for pidx in range(w.env.pool_steps):
        alone = sl.enc.forward(w.env.frames[pidx][sl.o + 11:sl.o + 12].contiguous())
        torch.cuda.synchronize()
        a = alone[0, 1:].float().reshape(-1)
        rels.append(((a - b).norm() / b.norm()).item())
    assert min(rels) < 2e-3, rels
    _check_properties(w)
    del w
    torch.cuda.empty_cache()


def test_config2_rn50_64_actors_fullsize_properties():
    """BASELINE config 2: 64 actors x rollout 128, CLIP-RN50 (two 32-frame encoder launches per env step)."""
    from embodied_clip_amd.engine import Worker
    w = Worker(64, T=128, device="cuda:0", seed=0)
    assert w.ns == 2 and w.encode_frames == 32
    w.collect_rollout()
    w.compute_returns()
    torch.cuda.synchronize()
    assert w.feat.shape == (129, 64, 49, 2048)
    sl = w.slices[0]
    b = sl.feat[1][7].float().reshape(-1)
    rels = []
    for pidx in range(w.env.pool_steps):
        alone = sl.enc.forward(w.env.frames[pidx][sl.o + 7:sl.o + 8].contiguous())
        torch.cuda.synchronize()
        a = alone[0].float().reshape(-1)
        rels.append(((a - b).norm() / b.norm()).item())
    assert min(rels) < 2e-3, rels
    assert (w.feat >= 0).all()
    _check_properties(w)
    del w
    torch.cuda.empty_cache()


def test_rn50x16_worker_properties():
    """[U] ClipResNetPreprocessor's second model type (readme_files/imagenet_vs_objectnav.md:10-11): the RN50x16 tower
    (width 96, layers (6, 8, 18, 8)) on 224 x 224 frames -> a 3072 x 7 x 7 map per frame, compressor 3072 -> 128 -> 32.
    Functional path (bench.py --encoder rn50x16): one rollout + GAE + update at 64 actors x rollout 8, the size-independent
    property set of the other configs, and a frame's features independent of the batch it was encoded in."""
    from embodied_clip_amd.engine import Worker
    w = Worker(64, T=8, device="cuda:0", seed=0, encoder="rn50x16")
    assert (w.S, w.C) == (7, 3072) and w.ns == 2
    w.collect_rollout()
    w.compute_returns()
    torch.cuda.synchronize()
    assert w.feat.shape == (9, 64, 49, 3072)
    sl = w.slices[0]
    b = sl.feat[1][5].float().reshape(-1)
    rels = []
    for pidx in range(w.env.pool_steps):
        alone = sl.enc.forward(w.env.frames[pidx][sl.o + 5:sl.o + 6].contiguous())
        torch.cuda.synchronize()
        a = alone[0].float().reshape(-1)
        rels.append(((a - b).norm() / b.norm()).item())
    assert min(rels) < 2e-3, rels
    assert (w.feat >= 0).all()
    _check_properties(w)
    del w
    torch.cuda.empty_cache()


def test_sharded_hip_gradients_sum_to_unsharded():
    """Row a18 on the HIP backward: two actor shards (what two ranks own), each pre-scaled by local/global batch,
    summed into one flat bucket == the unsharded HIP gradient (advantages given, so the loss is separable)."""
    from embodied_clip_amd.dist import grad_scale, shard_actors
    from embodied_clip_amd.policy import PolicyHandle
    from embodied_clip_amd.ppo import ppo_loss_raw
    dev = torch.device("cuda:0")
    T, N = 8, 12
    h = PolicyHandle()
    flat = h.flatten(syn.policy_state_dict(0), dev)
    g = torch.Generator().manual_seed(1)
    feat = (torch.randn(T, N, 49, 2048, generator=g).abs() * 0.5).to(torch.bfloat16).to(dev)
    goal = syn.synthetic_goals(3, (T, N)).to(dev)
    h0 = (torch.randn(N, 512, generator=g) * 0.3).to(dev)
    masks = syn.synthetic_masks(4, T, N, 0.2).reshape(T, N).to(dev)
    actions = torch.randint(0, 6, (T, N), generator=g).to(dev)
    old_lp = (-torch.rand(T, N, generator=g) - 0.5).to(dev)
    old_v, ret, nadv = (torch.randn(T, N, generator=g).to(dev) for _ in range(3))

    def grads(lo, cnt, scale):
        sl = slice(lo, lo + cnt)
        c = lambda t: t[:, sl].reshape(-1).contiguous()
        rows = feat[:, sl].reshape(T * cnt, 49, 2048).contiguous()
        ws = torch.empty(h.workspace_bytes(T, cnt, True), dtype=torch.uint8, device=dev)
        hv, _ = h.forward(flat, rows, c(goal), h0[sl].contiguous(), c(masks), T, cnt, ws)
        dhv, _ = ppo_loss_raw(hv, c(actions), c(old_lp), c(old_v), c(ret), c(nadv), 6, grad_scale=scale)
        gr = torch.zeros_like(flat)
        h.backward(flat, rows, c(masks), T, cnt, ws, dhv, None, gr)
        torch.cuda.synchronize()
        return gr

    full = grads(0, N, 1.0)
    bucket = torch.zeros_like(flat)
    for r in range(2):
        lo, cnt = shard_actors(N, r, 2)
        bucket += grads(lo, cnt, grad_scale(T * cnt, T * N))      # what allreduce_flat(SUM) computes over 2 ranks
    for name, (o, k) in h.offsets.items():
        a, b = bucket[o:o + k], full[o:o + k]
        assert _rel(a, b) < 2e-4, (name, _rel(a, b))
    # ... and, so that this is not the HIP path agreeing with itself: the ORACLE's unsharded gradient (torch-CPU
    # autograd through oracle/policy.py + oracle/ppo.py on the same batch) is what the summed bucket must equal
    sd = syn.policy_state_dict(0)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    feat_nchw = feat.float().cpu().view(T, N, 7, 7, 2048).permute(0, 1, 4, 2, 3).contiguous()
    lg, vv, _ = opol.actor_critic_forward(feat_nchw, goal.cpu(), h0.cpu().unsqueeze(0), masks.cpu().unsqueeze(-1), leaves)
    u = lambda t: t.cpu().unsqueeze(-1)
    total, _ = oppo.ppo_loss(lg, vv, actions.cpu(), u(old_lp), u(old_v), u(ret), u(nadv))
    names = list(h.offsets.keys())
    ref = dict(zip(names, torch.autograd.grad(total, [leaves[k] for k in names])))
    for name, (o, k) in h.offsets.items():
        a, b = bucket[o:o + k].cpu(), ref[name].reshape(-1)
        assert _rel(a, b) < 2e-4, (name, _rel(a, b))


def test_config5_zeroshot_fullsize_properties():
    """BASELINE config 5 at its full single-GPU size: 256 actors x rollout 128, CLIP-RN50 trunk + AttentionPool2d image
    embeddings in the rollout buffer, goal = CLIP text-tower table (same property set as configs 2 / 3)."""
    from embodied_clip_amd.engine import Worker
    w = Worker(256, T=128, device="cuda:0", seed=0, zeroshot=True)
    assert (w.S, w.C) == (1, 1024) and w.ns == 2 and w.encode_frames == 128
    w.collect_rollout()
    w.compute_returns()
    torch.cuda.synchronize()
    assert w.feat.shape == (129, 256, 1, 1024) and w.feat.dtype == torch.float32
    # goal table: unit rows (the policy multiplies normalised image embeddings with it)
    assert torch.allclose(w.goal_table.norm(dim=-1), torch.ones(12, device=w.goal_table.device), atol=1e-3)
    # an embedding produced inside the 128-frame launch equals the same frame embedded alone (best pool match)
    sl = w.slices[1]
    b = sl.feat[1][9].reshape(-1)
    rels = []
    for pidx in range(w.env.pool_steps):
        alone = sl.pool.forward(sl.enc.forward(w.env.frames[pidx][sl.o + 9:sl.o + 10].contiguous()))
        torch.cuda.synchronize()
        a = alone[0].reshape(-1)
        rels.append(((a - b).norm() / b.norm()).item())
    # (a lone frame is another launch shape -- other kernels, another summation order in layers 3-4: equal up to
    #  fp32-accumulation rounding amplified through the trunk, and still clearly the best match among the pool's shifted copies)
    assert min(rels) < 5e-3 and sorted(rels)[1] > 1.5 * min(rels), rels
    _check_properties(w)
    del w
    torch.cuda.empty_cache()
