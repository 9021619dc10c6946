"""GPU parity at the edges of the path's domain (the reference ships no tests; these are the degenerate inputs an RL
worker actually produces): a single actor, every episode resetting at once, no reset at all, saturated PPO clipping,
rollouts shorter than a tile, the sampler's distribution, and the C-ABI's error behaviour."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embodied_clip_amd import synthetic as syn  # noqa: E402
from oracle import policy as opol  # noqa: E402
from oracle import ppo as oppo  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()


def _forward(cfg, sd, feat, goal, h0, masks, T, N):
    from embodied_clip_amd.policy import PolicyHandle
    h = PolicyHandle(**cfg)
    flat = h.flatten(sd, DEV)
    rows = feat.permute(0, 1, 3, 4, 2).reshape(T * N, cfg["spatial"] ** 2, cfg["in_channels"]).contiguous()
    ws = torch.empty(h.workspace_bytes(T, N, False), dtype=torch.uint8, device=DEV)
    hv, hf = h.forward(flat, rows.to(DEV), goal.reshape(-1).to(DEV), h0[0].contiguous().to(DEV),
                       masks.reshape(-1).to(DEV), T, N, ws, for_backward=False)
    torch.cuda.synchronize()
    return hv.view(T, N, -1), hf


@pytest.mark.parametrize("T,N,mask_mode", [(1, 1, "ones"), (9, 1, "zeros"), (4, 7, "zeros"), (5, 2, "ones"), (3, 130, "mixed")])
def test_policy_forward_degenerate_batches_and_masks(T, N, mask_mode):
    cfg = dict(in_channels=64, spatial=3, hidden=32)
    sd = syn.policy_state_dict(1, **cfg)
    g = torch.Generator().manual_seed(T * 100 + N)
    feat = torch.randn(T, N, 64, 3, 3, generator=g).abs()
    goal = syn.synthetic_goals(5, (T, N))
    h0 = torch.randn(1, N, 32, generator=g)
    masks = {"ones": torch.ones(T, N, 1), "zeros": torch.zeros(T, N, 1),
             "mixed": syn.synthetic_masks(9, T, N, p_reset=0.5)}[mask_mode]
    ref_logits, ref_values, ref_h = opol.actor_critic_forward(feat, goal, h0, masks, sd)
    hv, hf = _forward(cfg, sd, feat, goal, h0, masks, T, N)
    assert _rel(hv[..., :6], ref_logits) < 2e-5 and _rel(hv[..., 6:], ref_values) < 2e-5 and _rel(hf, ref_h[0]) < 2e-5
    if mask_mode == "zeros":   # every step starts from a zeroed memory: the initial state must not leak in
        hv2, _ = _forward(cfg, sd, feat, goal, h0 * 0 + 7.0, masks, T, N)
        assert torch.equal(hv, hv2)


def test_gae_all_resets_no_resets_and_single_actor():
    from embodied_clip_amd import ppo
    for T, N, mode in ((1, 1, "ones"), (16, 3, "zeros"), (16, 3, "ones"), (128, 256, "mixed")):
        g = torch.Generator().manual_seed(T + N)
        masks = {"ones": torch.ones(T + 1, N, 1), "zeros": torch.zeros(T + 1, N, 1),
                 "mixed": torch.cat([torch.ones(1, N, 1), syn.synthetic_masks(2, T, N, p_reset=0.3)], 0)}[mode]
        rewards = torch.randn(T, N, 1, generator=g)
        values = torch.randn(T + 1, N, 1, generator=g)
        R = oppo.compute_returns(rewards, values, masks)
        r2, a2, n2 = ppo.compute_returns(rewards.to(DEV), values.to(DEV), masks.to(DEV))
        torch.cuda.synchronize()
        assert _rel(r2, R) < 1e-6, (T, N, mode)
        if mode == "zeros":   # bootstrap term masked everywhere: R[t] = r[t]
            assert torch.allclose(r2[:T].cpu().reshape(T, N, 1), rewards, atol=1e-6)
        if T * N > 1:
            adv, nadv = oppo.normalized_advantages(R, values)
            assert _rel(a2, adv) < 1e-5 and _rel(n2, nadv) < 2e-5, (T, N, mode)


def test_ppo_loss_saturated_clipping_and_extreme_logits():
    """Ratios far outside [1-c, 1+c] on both sides, value errors far outside the value clip, near-one-hot policies."""
    from embodied_clip_amd import ppo
    T, N, A = 5, 11, 6
    g = torch.Generator().manual_seed(4)
    logits = (torch.randn(T, N, A, generator=g) * 8.0).requires_grad_(True)        # near one-hot
    values = (torch.randn(T, N, 1, generator=g) * 5.0).requires_grad_(True)
    actions = torch.randint(0, A, (T, N), generator=g)
    with torch.no_grad():
        lp = opol.categorical_log_prob(logits, actions).unsqueeze(-1)
        old_lp = lp + torch.randn(T, N, 1, generator=g) * 1.5                         # ratios from ~0.05 to ~20
        old_v = values + torch.randn(T, N, 1, generator=g) * 3.0
    returns = torch.randn(T, N, 1, generator=g) * 5.0
    nadv = torch.randn(T, N, 1, generator=g)
    total, info = oppo.ppo_loss(logits, values, actions, old_lp, old_v, returns, nadv)
    total.backward()
    hv = torch.cat([logits, values], -1).detach().reshape(T * N, A + 1).contiguous().to(DEV)
    f = lambda t: t.reshape(-1).contiguous().to(DEV)  # noqa: E731
    dhv, sums = ppo.ppo_loss_raw(hv, f(actions), f(old_lp), f(old_v), f(returns), f(nadv), A)
    torch.cuda.synchronize()
    s = (sums / (T * N)).cpu()
    for got, key in zip(s.tolist(), ("action", "value", "entropy", "ratio_mean")):
        assert abs(got - info[key]) <= 1e-5 * max(1.0, abs(info[key])), (key, got, info[key])
    ref = torch.cat([logits.grad, values.grad], -1).reshape(T * N, A + 1)
    assert _rel(dhv, ref) < 1e-5
    assert torch.isfinite(dhv).all()


def test_sampler_follows_the_categorical_distribution_and_is_slice_invariant():
    from embodied_clip_amd import _lib
    lib = _lib.load()
    N, A = 4096, 6
    logits = torch.tensor([2.0, 0.5, -1.0, 0.0, 1.0, -3.0])
    p = torch.softmax(logits, 0)
    hv = torch.cat([logits.repeat(N, 1), torch.zeros(N, 1)], 1).contiguous().to(DEV)
    acts = torch.empty(N, dtype=torch.int64, device=DEV)
    logp = torch.empty(N, device=DEV); vals = torch.empty(N, device=DEV)
    counts = torch.zeros(A)
    for step in range(8):
        _lib.check(lib.ec_sample_actions(hv.data_ptr(), acts.data_ptr(), logp.data_ptr(), vals.data_ptr(), N, A, 1234, step,
                                         0, _lib.stream_ptr()), "sample")
        torch.cuda.synchronize()
        counts += torch.bincount(acts.cpu(), minlength=A).float()
        assert torch.allclose(logp.cpu(), torch.log(p)[acts.cpu()], atol=1e-6)
    freq = counts / counts.sum()
    assert (freq - p).abs().max() < 0.01, (freq, p)                     # 32768 draws: 4 sigma ~ 0.011
    # slice invariance: rows [1000, 1300) sampled on their own with first_actor = 1000 give the same actions
    sub = torch.empty(300, dtype=torch.int64, device=DEV)
    _lib.check(lib.ec_sample_actions(hv[1000:1300].contiguous().data_ptr(), sub.data_ptr(), logp.data_ptr(), vals.data_ptr(),
                                     300, A, 1234, 7, 1000, _lib.stream_ptr()), "sample")
    torch.cuda.synchronize()
    assert torch.equal(sub.cpu(), acts[1000:1300].cpu())


@pytest.mark.parametrize("B", [1, 3, 5])
def test_trunk_odd_batches_match_oracle(B):
    from embodied_clip_amd.encoder import RN50Trunk
    from oracle import clip_resnet as ocr
    sd = syn.rn50_visual_state_dict(2, layers=(1, 1, 1, 1))
    x = syn.synthetic_rgb(40 + B, B)
    trunk = RN50Trunk(sd, device=DEV)
    got = trunk.to_nchw_f32(trunk.forward(x.to(DEV))).cpu()
    ref = ocr.clip_resnet_preprocessor(x, sd)
    assert got.shape == ref.shape == (B, 2048, 7, 7)
    assert _rel(got, ref) < 2e-2
    # a frame's features do not depend on what else is in the batch (up to fp32-accumulation rounding amplified through the
    # trunk: the launch shape decides which kernel -- and so which fixed summation order -- the late 3x3 convs take)
    one = trunk.to_nchw_f32(trunk.forward(x[:1].to(DEV))).cpu()
    assert _rel(one[0], got[0]) <= 7e-3


def test_cabi_error_codes_not_exceptions():
    """Every entry point returns a negative status for bad arguments / shapes / workspaces instead of crashing."""
    from embodied_clip_amd import _lib
    lib = _lib.load()
    t = torch.zeros(4096, dtype=torch.float32, device=DEV)
    p = t.data_ptr()
    assert lib.ec_conv_bf16(None, p, p, None, p, 1, 8, 8, 64, 64, 1, 0, 1, 0) < 0                    # null input
    assert lib.ec_conv_bf16(p, p, p, None, p, 1, 8, 8, 64, 64, 5, 0, 1, 0) < 0                       # 5x5 kernel
    assert lib.ec_conv_bf16(p, p, p, None, p, 1, 8, 8, 60, 64, 1, 0, 1, 0) < 0                       # Cin % 8
    assert lib.ec_conv_bf16(p, p, p, None, p, 1, 7, 7, 64, 64, 3, 1, 1, 0) < 0                       # pool on odd H
    assert lib.ec_gemm_bf16(p, p, None, None, p, 0, 64, 64, 0, 0) < 0                                # M = 0
    assert lib.ec_gae(p, p, p, p, p, p, p, 0, 4, 0.99, 0.95, 1e-5, 0) < 0                            # T = 0
    assert lib.ec_probe_head(7, p, p, None, 4, 4, -1, None, None, None, p, 0) < 0                    # unknown mode
    assert lib.ec_probe_head(1, p, p, None, 4, 4, -1, None, None, None, p, 0) < 0                    # gather without idx
    assert lib.ec_rn50_forward(None, p, 1, p, 16, p, 0, 0) < 0                                       # null handle
    msg = lib.ec_strerror(lib.ec_conv_bf16(None, p, p, None, p, 1, 8, 8, 64, 64, 1, 0, 1, 0))
    assert msg and len(msg) > 3
    torch.cuda.synchronize()
