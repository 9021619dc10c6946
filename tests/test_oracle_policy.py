"""Oracle pins for the policy / GAE / PPO restatement."""
import torch
import torch.nn as nn

from embodied_clip_amd import synthetic as syn
from oracle import policy as opol
from oracle import ppo as oppo


def test_policy_param_checksum():
    assert opol.param_count(syn.policy_state_dict(0)) == 3_480_775


def test_gru_matches_torch_nn_gru_without_resets():
    sd = syn.policy_state_dict(1, in_channels=64, spatial=2, hidden=32)
    T, N, I, H = 5, 3, 32 * 4, 32
    g = torch.Generator().manual_seed(0)
    x = torch.randn(T, N, I, generator=g); h0 = torch.randn(1, N, H, generator=g)
    gru = nn.GRU(I, H, 1)
    gru.load_state_dict({k.split("rnn.")[1]: v for k, v in sd.items() if "rnn." in k})
    ref, hT = gru(x, h0)
    out, h = opol.rnn_state_encoder(x, h0, torch.ones(T, N, 1), sd)
    assert torch.allclose(out, ref, atol=1e-5) and torch.allclose(h, hT, atol=1e-5)


def test_gru_mask_resets_hidden():
    sd = syn.policy_state_dict(1, in_channels=64, spatial=2, hidden=32)
    T, N, I, H = 4, 2, 128, 32
    g = torch.Generator().manual_seed(1)
    x = torch.randn(T, N, I, generator=g); h0 = torch.randn(1, N, H, generator=g)
    m = torch.ones(T, N, 1); m[2, 1, 0] = 0
    out, _ = opol.rnn_state_encoder(x, h0, m, sd)
    # sampler 1 from t=2 on must equal a fresh run from zero hidden state
    out2, _ = opol.rnn_state_encoder(x[2:, 1:2], torch.zeros(1, 1, H), torch.ones(2, 1, 1), sd)
    assert torch.allclose(out[2:, 1:2], out2, atol=1e-6)


def test_goal_encoder_flatten_is_channel_major():
    sd = syn.policy_state_dict(2, in_channels=16, spatial=3, hidden=8)
    feat = torch.randn(2, 16, 3, 3, generator=torch.Generator().manual_seed(0))
    goal = torch.tensor([3, 7])
    y = opol.goal_encoder(feat, goal, sd)
    assert y.shape == (2, 32 * 9)


def test_gae_closed_form_single_step():
    r = torch.tensor([[[1.0]]]); v = torch.tensor([[[0.5]], [[2.0]]]); m = torch.ones(2, 1, 1)
    R = oppo.compute_returns(r, v, m, 0.99, 0.95)
    assert torch.allclose(R[0], torch.tensor([[1.0 + 0.99 * 2.0]]))
    m[1] = 0
    R = oppo.compute_returns(r, v, m, 0.99, 0.95)
    assert torch.allclose(R[0], torch.tensor([[1.0]]))


def test_adam_matches_torch_optim():
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(50, generator=g)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=3e-4)
    p = p0.clone(); m = [torch.zeros(50)]; v = [torch.zeros(50)]
    for step in range(1, 4):
        gr = torch.randn(50, generator=g)
        p_ref.grad = gr.clone(); opt.step()
        oppo.adam_step([p], [gr.clone()], m, v, step)
    assert torch.allclose(p, p_ref.detach(), atol=1e-7)


def test_ppo_loss_at_ratio_one():
    T, N, A = 3, 2, 6
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(T, N, A, generator=g); actions = torch.randint(0, A, (T, N), generator=g)
    old_lp = opol.categorical_log_prob(logits, actions).unsqueeze(-1)
    values = torch.randn(T, N, 1, generator=g); returns = torch.randn(T, N, 1, generator=g)
    adv = torch.randn(T, N, 1, generator=g)
    total, info = oppo.ppo_loss(logits, values, actions, old_lp, values.clone(), returns, adv)
    assert abs(info["action"] - float(-adv.mean())) < 1e-6
    assert abs(info["value"] - float(0.5 * ((values - returns) ** 2).mean())) < 1e-6
    assert abs(info["ratio_mean"] - 1.0) < 1e-6


def test_recurrent_minibatch_ranges_restates_allenact_generator():
    """[U] RolloutStorage.recurrent_generator: np.round(np.linspace(0, N, M + 1)) cuts, random.shuffle order."""
    import random
    import numpy as np
    from oracle import ppo as oppo
    for N, M in ((5, 2), (60, 1), (60, 7), (256, 2), (9, 9)):
        inds = np.round(np.linspace(0, N, M + 1, endpoint=True)).astype(np.int32)
        want = list(zip(inds[:-1].tolist(), inds[1:].tolist()))
        random.Random(11).shuffle(want)
        got = oppo.recurrent_minibatch_ranges(N, M, random.Random(11))
        assert got == want
        assert sorted(got)[0][0] == 0 and sorted(got)[-1][1] == N and sum(b - a for a, b in got) == N


def test_dual_goal_encoder_is_two_single_towers_sharing_the_goal_embedding():
    """oracle.dual_goal_encoder ([U] ResnetDualTensorGoalEncoder): cat(rgb_x, depth_x) of two goal_encoder passes whose
    compressor / combiner weights are the dual state dict's rgb_* / depth_* entries; parameter order and count of the
    dual layout (25 tensors, GRU input 2 * 32 * S * S)."""
    import torch
    from embodied_clip_amd import synthetic as syn
    from oracle import policy as opol
    C, S, H = 16, 3, 8
    sd = syn.policy_state_dict(3, in_channels=C, spatial=S, hidden=H, dual=1)
    assert tuple(sd.keys()) == syn.POLICY_PARAM_ORDER_DUAL and len(sd) == 25
    assert sd["state_encoder.rnn.weight_ih_l0"].shape == (3 * H, 2 * 32 * S * S)
    g = torch.Generator().manual_seed(4)
    rgb, depth = torch.randn(5, C, S, S, generator=g), torch.randn(5, C, S, S, generator=g)
    goal = torch.randint(0, 12, (5,), generator=g)
    x = opol.dual_goal_encoder(rgb, depth, goal, sd)
    P = "goal_visual_encoder."
    parts = []
    for tag, feat in (("rgb_", rgb), ("depth_", depth)):
        one = {P + "embed_class.weight": sd[P + "embed_class.weight"]}
        for k in ("resnet_compressor.0", "resnet_compressor.2", "target_obs_combiner.0", "target_obs_combiner.2"):
            for wb in ("weight", "bias"):
                one[P + k + "." + wb] = sd[P + tag + k + "." + wb]
        parts.append(opol.goal_encoder(feat, goal, one).view(5, 32, S * S))
    assert torch.equal(x, torch.cat(parts, dim=1).reshape(5, -1))


def test_categorical_head_matches_torch_distributions():
    """[U] allenact ``CategoricalDistr`` subclasses ``torch.distributions.Categorical(logits=...)``: the oracle's written-out
    ``log_prob`` / ``entropy`` against that class (an installed, independent implementation of row a14's arithmetic)."""
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(7, 5, 6, generator=g) * 3.0
    actions = torch.randint(0, 6, (7, 5), generator=g)
    d = torch.distributions.Categorical(logits=logits)
    assert torch.allclose(opol.categorical_log_prob(logits, actions), d.log_prob(actions), atol=1e-6)
    assert torch.allclose(opol.categorical_entropy(logits), d.entropy(), atol=1e-6)
    # saturated logits: p -> {0, 1}, entropy -> 0 without NaN (p log p at p = 0)
    sat = torch.tensor([[60.0, -60.0, -60.0, -60.0, -60.0, -60.0]])
    assert torch.isfinite(opol.categorical_entropy(sat)).all() and float(opol.categorical_entropy(sat)) < 1e-6


def test_gae_reverse_scan_equals_its_definition_as_a_forward_sum():
    """The oracle's GAE is the reverse scan of [U] ``RolloutStorage.compute_returns``.  Independent restatement: the DEFINITION
    A[t] = sum_{k >= t} (gamma tau)^(k - t) * prod_{j = t+1 .. k} m[j] * delta[k],  delta[k] = r[k] + gamma V[k+1] m[k+1] - V[k],
    R[t] = A[t] + V[t], evaluated as explicit O(T^2) sums in float64 with random episode resets."""
    T, N, gamma, tau = 23, 4, 0.99, 0.95
    g = torch.Generator().manual_seed(9)
    r = torch.randn(T, N, 1, generator=g)
    v = torch.randn(T + 1, N, 1, generator=g)
    m = (torch.rand(T + 1, N, 1, generator=g) > 0.2).float()
    R = oppo.compute_returns(r, v, m, gamma, tau)
    r64, v64, m64 = r.double(), v.double(), m.double()
    for t in range(T):
        acc = torch.zeros(N, 1, dtype=torch.float64)
        for k in range(t, T):
            w = (gamma * tau) ** (k - t) * torch.ones(N, 1, dtype=torch.float64)
            for j in range(t + 1, k + 1):
                w = w * m64[j]
            acc += w * (r64[k] + gamma * v64[k + 1] * m64[k + 1] - v64[k])
        assert torch.allclose(R[t].double(), acc + v64[t], atol=1e-5), t
    assert torch.equal(R[T], v[T])
    adv, nadv = oppo.normalized_advantages(R, v)
    assert torch.allclose(adv, R[:-1] - v[:-1]) and abs(float(nadv.mean())) < 1e-5
    assert abs(float(nadv.std()) - 1.0) < 1e-3            # unbiased std (torch default), + eps in the denominator
