"""GPU parity: policy forward/backward, GAE, PPO loss, clip+Adam (through the C-ABI) vs the CPU oracle.

Tolerance (all-fp32 path; the MFMA is an exact fp32 fmaf chain, only the summation ORDER differs from
torch-CPU): forward rel-L2 <= 2e-5; gradients rel-L2 <= 2e-4 per tensor (K up to T*N*49 terms, fp32 atomics).
With bf16-stored features both sides see identical (bf16-representable) inputs, so the same bounds hold.
"""
import pytest
import torch

from embodied_clip_amd import synthetic as syn
from oracle import policy as opol
from oracle import ppo as oppo

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _gemm(dev, A, B, C, M, N, K, sam, sak, sbk, sbn, ldc, flags=0, bias=None, gbias=None, gidx=None, group=0,
          dmask=None, rowscale=None, splitk=1):
    from embodied_clip_amd import _lib
    lib = _lib.load()
    _lib.check(lib.ec_gemm_f32(A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, sam, sak, sbk, sbn, ldc, flags,
                               _lib.ptr(bias), _lib.ptr(gbias), _lib.ptr(gidx), group, _lib.ptr(dmask),
                               _lib.ptr(rowscale), splitk, 0))
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,N,K", [(200, 128, 2048), (147, 32, 128), (50, 1536, 1568), (5, 7, 512), (300, 96, 20)])
def test_gemm_f32_nt(dev, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) * K ** -0.5; b = torch.randn(N, generator=g)
    ref = torch.relu(a @ w.t() + b)
    ad, wd, bd = a.to(dev), w.to(dev), b.to(dev)
    c = torch.empty(M, N, device=dev)
    _gemm(dev, ad, wd, c, M, N, K, K, 1, 1, K, N, flags=4, bias=bd)
    assert _rel(c, ref) < 2e-5, _rel(c, ref)


def test_gemm_three_leading_products_option(dev):
    """EC_GEMM_3PRODUCTS (what EC_GEMM_BWD3=1 passes for ec_policy_backward's large gradient GEMMs; default off): only
    a0 b0 + a0 b1 + a1 b0 of the bf16x3 split.  Against an fp64 product: the six-product result is fp32-exact (~1e-7), the
    three-product one carries the dropped 2^-16 terms (~1e-5 .. 1e-6 after a K = 1536 dot product) -- inside the 2e-4 the
    gradient parity tests allow, and visibly different from the exact path (so the flag is really taken)."""
    M, N, K = 512, 1568, 1536
    g = torch.Generator().manual_seed(5)
    a = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) * K ** -0.5
    ref = a.double() @ w.double().t()
    ad, wd = a.to(dev), w.to(dev)
    c6, c3 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    _gemm(dev, ad, wd, c6, M, N, K, K, 1, 1, K, N)
    _gemm(dev, ad, wd, c3, M, N, K, K, 1, 1, K, N, flags=32)
    e6 = ((c6.cpu().double() - ref).norm() / ref.norm()).item()
    e3 = ((c3.cpu().double() - ref).norm() / ref.norm()).item()
    assert e6 < 1e-6, e6
    assert 1e-7 < e3 < 3e-5, e3
    assert not torch.equal(c3, c6)


@pytest.mark.parametrize("M,N,K,bf16a", [(6272, 128, 2048, True), (128, 1536, 1568, False), (70, 96, 160, False)])
def test_gemm_split_parts_sum_to_the_full_product(dev, M, N, K, bf16a):
    """EC_GEMM_SPLIT_PARTS (flag 16): K slices write separate partial matrices (bias in part 0, no atomics); their sum in
    slice order is the GEMM, and two runs are bit-identical (the act step's compressor / GRU input projections).
    K = 160 with 4 parts leaves the last slice short; invalid combinations are refused."""
    from embodied_clip_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, K, generator=g)
    a = a.abs().to(torch.bfloat16) if bf16a else a
    w = torch.randn(N, K, generator=g) * K ** -0.5
    b = torch.randn(N, generator=g)
    ref = a.double() @ w.double().t() + b.double()
    ad, wd, bd = a.to(dev), w.to(dev), b.to(dev)
    parts = torch.full((4, M, N), float("nan"), device=dev)
    flags = 16 | (1 if bf16a else 0)
    for _ in range(2):
        _gemm(dev, ad, wd, parts, M, N, K, K, 1, 1, K, N, flags=flags, bias=bd, splitk=4)
        if _ == 0:
            first = parts.clone()
    assert torch.equal(first, parts)
    assert torch.isfinite(parts).all()
    tot = ((parts[0] + parts[1]) + parts[2]) + parts[3]
    assert (tot.cpu().double() - ref).abs().max() < 3e-5 * max(1.0, ref.abs().max().item())
    # ReLU / ACCUMULATE cannot be combined with parts
    rc = lib.ec_gemm_f32(ad.data_ptr(), wd.data_ptr(), parts.data_ptr(), M, N, K, K, 1, 1, K, N, flags | 4, bd.data_ptr(),
                         None, None, 0, None, None, 4, 0)
    assert rc != 0


def test_gemm_f32_bf16_operands_nn_tn_epilogues(dev):
    g = torch.Generator().manual_seed(5)
    M, N, K = 333, 200, 260
    # NN with accumulate + rowscale + dmask
    a = torch.randn(M, K, generator=g); b = torch.randn(K, N, generator=g) * K ** -0.5
    c0 = torch.randn(M, N, generator=g); rs = torch.rand(M, generator=g); dm = torch.randn(M, N, generator=g)
    # epilogue order: rowscale -> dmask -> accumulate
    ref = c0 + torch.where(dm > 0, (a @ b) * rs[:, None], torch.zeros(()))
    ad, bd, cd, rsd, dmd = a.to(dev), b.to(dev), c0.clone().to(dev), rs.to(dev), dm.to(dev)
    _gemm(dev, ad, bd, cd, M, N, K, K, 1, N, 1, N, flags=8, dmask=dmd, rowscale=rsd)
    assert _rel(cd, ref) < 2e-5
    # TN split-K with a bf16 B operand (dW1 = dc1^T feat): out[m,n] = sum_k dY[k,m] X[k,n]
    Kb = 5000
    dy = torch.randn(Kb, 128, generator=g); x = torch.randn(Kb, 256, generator=g).to(torch.bfloat16)
    ref = dy.t() @ x.float()
    dyd, xd = dy.to(dev), x.to(dev)
    out = torch.zeros(128, 256, device=dev)
    _gemm(dev, dyd, xd, out, 128, 256, Kb, 1, 128, 256, 1, 256, flags=2 | 8, splitk=7)
    assert _rel(out, ref) < 5e-5
    # NT with a bf16 A operand and a row-group bias table indexed through gidx
    feat = torch.randn(98, 64, generator=g).to(torch.bfloat16); w = torch.randn(32, 64, generator=g)
    tab = torch.randn(5, 32, generator=g); gi = torch.tensor([3, 1], dtype=torch.int32)
    ref = feat.float() @ w.t() + tab[gi.long()].repeat_interleave(49, 0)
    fd, wd, td, gid = feat.to(dev), w.to(dev), tab.to(dev), gi.to(dev)
    o = torch.empty(98, 32, device=dev)
    _gemm(dev, fd, wd, o, 98, 32, 64, 64, 1, 1, 64, 32, flags=1, gbias=td, gidx=gid, group=49)
    assert _rel(o, ref) < 2e-5


@pytest.mark.parametrize("M,N,K,relu", [(1000, 128, 256, True), (256 * 3 + 17, 256, 2048, False), (40000, 128, 2048, True),
                                           (5000, 128, 768, True)])
def test_gemm_bf16a_x3_pingpong_matches_fp64(dev, M, N, K, relu):
    """bf16 activations x fp32 weights (three bf16 planes) on the 8-wave ping-pong kernel: exact-fp32 grade product
    (the compressor's first 1x1 conv over stored features); ragged last tile, two N tiles, bias / ReLU."""
    from embodied_clip_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + K)
    a = torch.randn(M, K, generator=g).abs().to(torch.bfloat16)
    w = torch.randn(N, K, generator=g) * K ** -0.5
    b = torch.randn(N, generator=g)
    ref = a.double() @ w.double().t() + b.double()
    ref = torch.relu(ref) if relu else ref
    ad, wd, bd = a.to(dev), w.to(dev), b.to(dev)
    planes = torch.empty(N, 3, K, dtype=torch.bfloat16, device=dev)
    out = torch.full((M, N), float("nan"), device=dev)
    _lib.check(lib.ec_split3_bf16(wd.data_ptr(), planes.data_ptr(), N, K, 0))
    _lib.check(lib.ec_gemm_bf16a_x3(ad.data_ptr(), planes.data_ptr(), bd.data_ptr(), out.data_ptr(), M, N, K, 1 if relu else 0, 0))
    torch.cuda.synchronize()
    # the planes reconstruct W to ~2^-24 relative
    rec = planes.float().sum(1)
    assert (rec.cpu() - w).abs().max() <= 2 ** -22 * w.abs().max()
    assert _rel(out, ref.float()) < 2e-6
    assert (out.cpu().double() - ref).abs().max() < 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("M,NX", [(64, 256), (1000, 512), (49 * 37 * 4, 2048), (70001, 256), (4001, 768)])
def test_dw_tn_x3_transpose_read_matches_fp64(dev, M, NX):
    """dW[128, NX] += dY^T X with both operands token-major (LDS transpose reads), dY as three bf16 planes: ragged
    last K-tile, empty trailing splits, accumulation into a non-zero dW; the operands are asymmetric and random so
    a transposed or permuted fragment cannot pass."""
    from embodied_clip_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + NX)
    dy = torch.randn(M, 128, generator=g) * torch.rand(1, 128, generator=g)
    x = torch.randn(M, NX, generator=g).abs().to(torch.bfloat16)
    w0 = torch.randn(128, NX, generator=g)
    ref = w0.double() + dy.double().t() @ x.double()
    dyd, xd, wd = dy.to(dev), x.to(dev), w0.to(dev)
    planes = torch.empty(M, 3, 128, dtype=torch.bfloat16, device=dev)
    _lib.check(lib.ec_split3_bf16(dyd.data_ptr(), planes.data_ptr(), M, 128, 0))
    ns = lib.ec_dw_tn_x3_splits(M, NX)
    assert ns >= 1
    part = torch.full((ns, 128, NX), float("nan"), device=dev)
    _lib.check(lib.ec_dw_tn_x3(planes.data_ptr(), xd.data_ptr(), part.data_ptr(), wd.data_ptr(), M, NX, 0))
    torch.cuda.synchronize()
    err = (wd.cpu().double() - ref).abs().max().item()
    assert err < 3e-6 * max(1.0, ref.abs().max().item()) * max(1.0, (M / 1000) ** 0.5), err


def _policy_case(T, N, C=64, S=3, H=32, seed=0, bf16=False):
    cfg = dict(in_channels=C, spatial=S, hidden=H)
    sd = syn.policy_state_dict(seed, **cfg)
    g = torch.Generator().manual_seed(seed + 1)
    feat = torch.randn(T, N, C, S, S, generator=g).abs()       # post-ReLU features are non-negative
    if bf16:
        feat = feat.to(torch.bfloat16).float()
    goal = syn.synthetic_goals(seed + 2, (T, N))
    h0 = torch.randn(1, N, H, generator=g) * 0.5
    masks = syn.synthetic_masks(seed + 3, T, N, p_reset=0.2)
    return cfg, sd, feat, goal, h0, masks


@pytest.mark.parametrize("T,N,bf16,S", [(1, 5, False, 3), (6, 4, False, 3), (5, 3, True, 3), (3, 5, False, 7), (4, 37, True, 7)])
def test_policy_forward_matches_oracle(dev, T, N, bf16, S):
    from embodied_clip_amd.policy import PolicyHandle
    cfg, sd, feat, goal, h0, masks = _policy_case(T, N, S=S, bf16=bf16)
    ref_logits, ref_values, ref_h = opol.actor_critic_forward(feat, goal, h0, masks, sd)
    h = PolicyHandle(**cfg)
    flat = h.flatten(sd, dev)
    rows = feat.permute(0, 1, 3, 4, 2).reshape(T * N, cfg["spatial"] ** 2, cfg["in_channels"]).contiguous()
    rows = rows.to(torch.bfloat16) if bf16 else rows
    ws = torch.empty(h.workspace_bytes(T, N, False), dtype=torch.uint8, device=dev)
    hv, hf = h.forward(flat, rows.to(dev), goal.reshape(-1).to(dev), h0[0].contiguous().to(dev),
                       masks.reshape(-1).to(dev), T, N, ws, for_backward=False)
    torch.cuda.synchronize()
    hv = hv.view(T, N, -1)
    assert _rel(hv[..., :6], ref_logits) < 2e-5
    assert _rel(hv[..., 6:], ref_values) < 2e-5
    assert _rel(hf, ref_h[0]) < 2e-5


def _loss_inputs(T, N, seed):
    g = torch.Generator().manual_seed(seed)
    actions = torch.randint(0, 6, (T, N), generator=g)
    old_lp = -torch.rand(T, N, 1, generator=g) * 2.5
    old_v = torch.randn(T, N, 1, generator=g)
    returns = torch.randn(T, N, 1, generator=g)
    nadv = torch.randn(T, N, 1, generator=g)
    return actions, old_lp, old_v, returns, nadv


def test_ppo_loss_forward_backward_matches_oracle(dev):
    from embodied_clip_amd import ppo
    T, N, A = 7, 9, 6
    g = torch.Generator().manual_seed(2)
    logits = torch.randn(T, N, A, generator=g, requires_grad=True)
    values = torch.randn(T, N, 1, generator=g, requires_grad=True)
    actions, old_lp, old_v, returns, nadv = _loss_inputs(T, N, 3)
    # put old_log_probs near the new ones so both clip branches are exercised
    with torch.no_grad():
        old_lp = opol.categorical_log_prob(logits, actions).unsqueeze(-1) + 0.25 * torch.randn(T, N, 1, generator=g)
        old_v = values + 0.2 * torch.randn(T, N, 1, generator=g)
    total, info = oppo.ppo_loss(logits, values, actions, old_lp, old_v, returns, nadv)
    total.backward()
    hv = torch.cat([logits, values], -1).detach().reshape(T * N, A + 1).contiguous().to(dev)
    f = lambda t: t.reshape(-1).contiguous().to(dev)
    dhv, sums = ppo.ppo_loss_raw(hv, f(actions), f(old_lp), f(old_v), f(returns), f(nadv), A)
    torch.cuda.synchronize()
    s = (sums / (T * N)).cpu()
    assert abs(s[0].item() - info["action"]) < 1e-5 and abs(s[1].item() - info["value"]) < 1e-5
    assert abs(s[2].item() - info["entropy"]) < 1e-5 and abs(s[3].item() - info["ratio_mean"]) < 1e-5
    ref = torch.cat([logits.grad, values.grad], -1).reshape(T * N, A + 1)
    assert _rel(dhv, ref) < 1e-5, _rel(dhv, ref)


def test_gae_matches_oracle(dev):
    from embodied_clip_amd import ppo
    T, N = 16, 4
    masks = torch.cat([torch.ones(1, N, 1), syn.synthetic_masks(4, T, N, p_reset=0.15)], 0)
    rewards = syn.synthetic_rewards(5, masks[1:])
    values = torch.randn(T + 1, N, 1, generator=torch.Generator().manual_seed(6))
    R = oppo.compute_returns(rewards, values, masks)
    adv, nadv = oppo.normalized_advantages(R, values)
    r2, a2, n2 = ppo.compute_returns(rewards.to(dev), values.to(dev), masks.to(dev))
    torch.cuda.synchronize()
    assert _rel(r2, R) < 1e-6 and _rel(a2, adv) < 1e-5 and _rel(n2, nadv) < 1e-5


@pytest.mark.parametrize("T,N,bf16,S,C,H", [(6, 4, False, 3, 64, 32), (8, 4, True, 3, 64, 32), (3, 5, False, 7, 64, 32),
                                            (4, 37, True, 7, 64, 32), (3, 19, True, 7, 256, 32),
                                            (4, 19, True, 7, 64, 512), (3, 33, False, 3, 64, 512), (2, 16, True, 7, 64, 512),
                                            (64, 16, True, 3, 64, 512)])
def test_policy_backward_and_update_step_match_oracle(dev, T, N, bf16, S, C, H):
    """One full optimiser step of HOT LOOP B: forward, PPO loss, backward, clip, Adam.
    S = 7 is the reference's 7x7 feature map: the fused tail kernels (tail_fwd_kernel / tail_bwd_kernel) run there, with
    ragged last tiles (T*N*49 not a multiple of 32) and row groups straddling tiles; S = 3 keeps the GEMM path covered.
    C = 256 with bf16 features additionally takes dW1 through the transpose-read kernel (dc1 as bf16 planes).
    H = 512 (the reference's hidden size) runs the learn pass's recurrences on the 16 x 16-tile step kernels
    (gru_step_fwd512_kernel / gru_step_bwd512_kernel): N = 19 / 33 leave ragged last actor tiles, N = 16 exactly one;
    T * N = 1024 rows additionally takes the GRU's bias gradients through the 16-B column-sum kernel (colsum4_kernel)."""
    from embodied_clip_amd import ppo
    from embodied_clip_amd.policy import PolicyHandle
    cfg, sd, feat, goal, h0, masks = _policy_case(T, N, C=C, S=S, H=H, seed=7, bf16=bf16)
    actions, old_lp, old_v, returns, nadv = _loss_inputs(T, N, 8)
    with torch.no_grad():
        lg, vv, _ = opol.actor_critic_forward(feat, goal, h0, masks, sd)
        old_lp = opol.categorical_log_prob(lg, actions).unsqueeze(-1) + 0.2 * torch.randn(T, N, 1)
        old_v = vv + 0.2 * torch.randn(T, N, 1)
    sd_ref = {k: v.clone() for k, v in sd.items()}
    batch = dict(feat=feat, goal=goal, h0=h0, masks=masks, actions=actions, old_log_probs=old_lp, old_values=old_v,
                 returns=returns, norm_adv=nadv)
    info, ref_grads = oppo.ppo_update_step(sd_ref, batch, {}, lr=3e-4, max_grad_norm=0.5)

    h = PolicyHandle(**cfg)
    flat = h.flatten(sd, dev)
    rows = feat.permute(0, 1, 3, 4, 2).reshape(T * N, cfg["spatial"] ** 2, cfg["in_channels"]).contiguous()
    rows = (rows.to(torch.bfloat16) if bf16 else rows).to(dev)
    m = masks.reshape(-1).to(dev)
    ws = torch.empty(h.workspace_bytes(T, N, True), dtype=torch.uint8, device=dev)
    hv, _ = h.forward(flat, rows, goal.reshape(-1).to(dev), h0[0].contiguous().to(dev), m, T, N, ws)
    f = lambda t: t.reshape(-1).contiguous().to(dev)
    dhv, sums = ppo.ppo_loss_raw(hv, f(actions), f(old_lp), f(old_v), f(returns), f(nadv), 6)
    grads = torch.zeros_like(flat)
    h.backward(flat, rows, m, T, N, ws, dhv, None, grads)
    torch.cuda.synchronize()
    total = ((sums[0] + 0.5 * sums[1] + 0.01 * sums[2]) / (T * N)).item()
    assert abs(total - info["ppo_total"]) < 1e-5 * max(1.0, abs(info["ppo_total"]))
    gv = h.views(grads)
    for name, gref in ref_grads.items():
        assert _rel(gv[name], gref) < 2e-4, (name, _rel(gv[name], gref))
    opt = ppo.FlatAdam(flat, lr=3e-4, max_grad_norm=0.5)
    opt.step(grads)
    torch.cuda.synchronize()
    assert abs(opt.grad_norm() - info["grad_norm"]) < 1e-4 * info["grad_norm"]
    pv = h.views(flat)
    for name, pref in sd_ref.items():
        # Adam's first step moves every weight by ~lr regardless of gradient scale: compare the UPDATE
        # (an element whose gradient is within ~100x of Adam's eps = 1e-8 is ill-conditioned -- its step is
        # lr * g / (|g| + eps) -- so those are only bounded by the step size)
        upd, upd_ref = pv[name].cpu() - sd[name], pref - sd[name]
        well = ref_grads[name].abs() > 1e-6
        assert ((upd - upd_ref).abs() * well).max() < 0.05 * 3e-4 + 1e-7, name
        assert upd.abs().max() <= 3e-4 * 1.001 + 1e-7, name


def test_actor_critic_module_autograd_surface(dev):
    """The drop-in nn.Module: names, memory spec, autograd-visible forward, flat grad bucket."""
    from embodied_clip_amd import spaces
    from embodied_clip_amd.policy import Memory, ResnetTensorObjectNavActorCritic
    from embodied_clip_amd.ppo import PPO
    T, N = 4, 3
    cfg, sd, feat, goal, h0, masks = _policy_case(T, N, C=64, S=3, H=32, seed=11)
    obs_space = spaces.Dict({"rgb_clip_resnet": spaces.Box(-1e9, 1e9, (64, 3, 3)), "goal": spaces.Discrete(12)})
    model = ResnetTensorObjectNavActorCritic(spaces.Discrete(6), obs_space, "goal", "rgb_clip_resnet", hidden_size=32,
                                             state_dict=sd, device=dev)
    assert [n for n, _ in model.named_parameters()] == list(syn.POLICY_PARAM_ORDER)
    assert model._recurrent_memory_specification()["rnn"][0] == (("layer", 1), ("sampler", None), ("hidden", 32))
    assert model.recurrent_memory_specification["rnn"][1] == torch.float32 and not model.is_blind
    mem = Memory().check_append("rnn", h0.to(dev), 1)
    out, mem2 = model({"rgb_clip_resnet": feat.to(dev), "goal": goal.to(dev)}, mem, None, masks.to(dev))
    ref_logits, ref_values, ref_h = opol.actor_critic_forward(feat, goal, h0, masks, sd)
    # CategoricalDistr holds NORMALISED logits (torch.distributions.Categorical semantics, as upstream)
    assert _rel(out.distributions.logits, torch.log_softmax(ref_logits, -1)) < 2e-5 and _rel(out.values, ref_values) < 2e-5
    assert _rel(mem2.tensor("rnn"), ref_h) < 2e-5
    actions, old_lp, old_v, returns, nadv = _loss_inputs(T, N, 12)
    batch = dict(actions=actions.to(dev), old_action_log_probs=old_lp.to(dev), values=old_v.to(dev),
                 returns=returns.to(dev), norm_adv_targ=nadv.to(dev), adv_targ=nadv.to(dev))
    total, info = PPO().loss(0, batch, out)
    total.backward()
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    lg, vv, _ = opol.actor_critic_forward(feat, goal, h0, masks, leaves)
    tref, iref = oppo.ppo_loss(lg, vv, actions, old_lp, old_v, returns, nadv)
    tref.backward()
    assert abs(float(total) - float(tref)) < 1e-5
    for n, p in model.named_parameters():
        assert p.grad is not None and _rel(p.grad, leaves[n].grad) < 2e-4, n
    assert model.flat_grads.abs().sum() > 0          # grads landed in the single flat bucket


def test_dual_rgbd_actor_critic_module_surface(dev):
    """The drop-in module with BOTH preprocessor uuids = [U] ResnetDualTensorGoalEncoder: upstream's parameter names,
    fp32 NCHW observations of both towers, autograd-visible forward, gradients vs the oracle."""
    from embodied_clip_amd import spaces
    from embodied_clip_amd.policy import Memory, ResnetTensorObjectNavActorCritic
    from embodied_clip_amd.ppo import PPO
    T, N, C, S, H = 3, 4, 64, 3, 32
    sd = syn.policy_state_dict(21, in_channels=C, spatial=S, hidden=H, dual=1)
    g = torch.Generator().manual_seed(22)
    rgb, depth = torch.randn(T, N, C, S, S, generator=g).abs(), torch.randn(T, N, C, S, S, generator=g).abs()
    goal = syn.synthetic_goals(23, (T, N))
    h0 = torch.randn(1, N, H, generator=g) * 0.5
    masks = syn.synthetic_masks(24, T, N, p_reset=0.2)
    obs_space = spaces.Dict({"rgb_clip_resnet": spaces.Box(-1e9, 1e9, (C, S, S)), "depth_clip_resnet": spaces.Box(-1e9, 1e9, (C, S, S)),
                             "goal": spaces.Discrete(12)})
    model = ResnetTensorObjectNavActorCritic(spaces.Discrete(6), obs_space, "goal", "rgb_clip_resnet", "depth_clip_resnet",
                                             hidden_size=H, state_dict=sd, device=dev)
    names = [n for n, _ in model.named_parameters()]
    assert sorted(names) == sorted(syn.POLICY_PARAM_ORDER_DUAL) and len(names) == 25
    assert "goal_visual_encoder.depth_target_obs_combiner.2.bias" in names and "goal_visual_encoder.rgb_resnet_compressor.0.weight" in names
    mem = Memory().check_append("rnn", h0.to(dev), 1)
    out, mem2 = model({"rgb_clip_resnet": rgb.to(dev), "depth_clip_resnet": depth.to(dev), "goal": goal.to(dev)}, mem, None,
                      masks.to(dev))
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    lg, vv, hf = opol.actor_critic_forward((rgb, depth), goal, h0, masks, leaves)
    assert _rel(out.distributions.logits, torch.log_softmax(lg, -1)) < 2e-5 and _rel(out.values, vv) < 2e-5
    assert _rel(mem2.tensor("rnn"), hf) < 2e-5
    actions, old_lp, old_v, returns, nadv = _loss_inputs(T, N, 25)
    batch = dict(actions=actions.to(dev), old_action_log_probs=old_lp.to(dev), values=old_v.to(dev),
                 returns=returns.to(dev), norm_adv_targ=nadv.to(dev), adv_targ=nadv.to(dev))
    total, _ = PPO().loss(0, batch, out)
    total.backward()
    tref, _ = oppo.ppo_loss(lg, vv, actions, old_lp, old_v, returns, nadv)
    tref.backward()
    assert abs(float(total.detach()) - float(tref.detach())) < 1e-5
    for n, p in model.named_parameters():
        assert p.grad is not None and _rel(p.grad, leaves[n].grad) < 2e-4, n


def test_module_learn_pass_uses_bf16_rows_only_when_the_fp32_storage_holds_bf16_values(dev):
    """The reference's tensor contract hands the policy fp32 NCHW features; ClipResNetPreprocessor's are bf16 values
    widened to fp32.  In the learn pass (T > 1, grad enabled) the module converts such a storage ONCE per version to bf16
    NHWC rows (ec_nchw_f32_to_nhwc_bf16 checks exactness while converting) and re-uses them for the following epochs; a
    storage with any other value keeps the fp32 path.  Results match the oracle either way."""
    from embodied_clip_amd import spaces
    from embodied_clip_amd.policy import Memory, ResnetTensorObjectNavActorCritic
    T, N = 4, 3
    cfg, sd, feat, goal, h0, masks = _policy_case(T, N, C=64, S=3, H=32, seed=31, bf16=True)   # bf16-exact fp32 values
    obs_space = spaces.Dict({"rgb_clip_resnet": spaces.Box(-1e9, 1e9, (64, 3, 3)), "goal": spaces.Discrete(12)})
    model = ResnetTensorObjectNavActorCritic(spaces.Discrete(6), obs_space, "goal", "rgb_clip_resnet", hidden_size=32,
                                             state_dict=sd, device=dev)
    storage = torch.zeros(T + 1, N, 64, 3, 3, device=dev)
    storage[:T] = feat.to(dev)
    fresh = lambda: Memory().check_append("rnn", h0.to(dev), 1)      # (Memory.set_tensor updates the object it is given)
    ref_logits, ref_values, _ = opol.actor_critic_forward(feat, goal, h0, masks, sd)

    def run():
        out, _ = model({"rgb_clip_resnet": storage[:T], "goal": goal.to(dev)}, fresh(), None, masks.to(dev))
        assert _rel(out.distributions.logits, torch.log_softmax(ref_logits, -1)) < 2e-5 and _rel(out.values, ref_values) < 2e-5
        return out

    run()
    first = model._rows_cache[0][2]
    assert first is not None and first.dtype == torch.bfloat16 and first.shape == (T * N, 9, 64)
    out = run()                                            # second epoch: the same storage version -> the same rows
    assert model._rows_cache[0][2] is first
    out.values.sum().backward()                            # (the bf16 rows feed the backward too)
    assert all(p.grad is not None for p in model.parameters())
    with torch.no_grad():                                  # act steps never take this path
        model({"rgb_clip_resnet": storage[:1], "goal": goal[:1].to(dev)}, fresh(), None, masks[:1].to(dev))
    assert model._rows_cache[0][2] is first
    storage[0, 0, 0, 0, 0] += 1e-3                         # not a bf16 any more (and a new storage version)
    feat2 = storage[:T].cpu()
    ref_logits, ref_values, _ = opol.actor_critic_forward(feat2, goal, h0, masks, sd)
    run()
    assert model._rows_cache[0][2] is None                 # fp32 path


def test_ppo_variants_unclipped_value_loss_clip_decay_and_bad_actions(dev):
    """PPO options behind the reference's configs ([U] losses/ppo.py): use_clipped_value_loss=False,
    clip_param * clip_decay(step_count), and loud failure on an out-of-range action id."""
    from embodied_clip_amd import ppo
    from embodied_clip_amd.policy import ActorCriticOutput, CategoricalDistr
    T, N, A = 5, 8, 6
    g = torch.Generator().manual_seed(4)
    logits = torch.randn(T, N, A, generator=g, requires_grad=True)
    values = torch.randn(T, N, 1, generator=g, requires_grad=True)
    actions, old_lp, old_v, returns, nadv = _loss_inputs(T, N, 5)
    with torch.no_grad():
        old_lp = opol.categorical_log_prob(logits, actions).unsqueeze(-1) + 0.25 * torch.randn(T, N, 1, generator=g)
        old_v = values + 0.2 * torch.randn(T, N, 1, generator=g)
    hv = torch.cat([logits, values], -1).detach().reshape(T * N, A + 1).contiguous().to(dev)
    f = lambda t: t.reshape(-1).contiguous().to(dev)
    # (1) unclipped value loss, raw kernel: forward sums and gradient
    total, info = oppo.ppo_loss(logits, values, actions, old_lp, old_v, returns, nadv, use_clipped_value_loss=False)
    total.backward()
    dhv, sums = ppo.ppo_loss_raw(hv, f(actions), f(old_lp), f(old_v), f(returns), f(nadv), A,
                                 use_clipped_value_loss=False)
    torch.cuda.synchronize()
    s = (sums / (T * N)).cpu()
    assert abs(s[1].item() - info["value"]) < 1e-5 and abs(s[0].item() - info["action"]) < 1e-5
    assert _rel(dhv, torch.cat([logits.grad, values.grad], -1).reshape(T * N, A + 1)) < 1e-5
    # (2) the shim applies clip_param * clip_decay(step_count) (a linear decay to 0 over 100 steps, step 40 -> 0.06)
    decay = lambda step: max(0.0, 1.0 - step / 100.0)
    lg = logits.detach().to(dev).requires_grad_(True)
    vv = values.detach().to(dev).requires_grad_(True)
    out = ActorCriticOutput(CategoricalDistr(logits=lg), vv, {})
    batch = dict(actions=actions.to(dev), old_action_log_probs=old_lp.to(dev), values=old_v.to(dev),
                 returns=returns.to(dev), norm_adv_targ=nadv.to(dev), adv_targ=nadv.to(dev))
    t40, _ = ppo.PPO(clip_param=0.1, clip_decay=decay).loss(40, batch, out)
    l2, v2 = logits.detach().clone().requires_grad_(True), values.detach().clone().requires_grad_(True)
    tref, _ = oppo.ppo_loss(l2, v2, actions, old_lp, old_v, returns, nadv, clip_param=0.1 * 0.6)
    assert abs(float(t40) - float(tref)) < 1e-5
    t0, _ = ppo.PPO(clip_param=0.1, clip_decay=decay).loss(0, batch, out)
    tref0, _ = oppo.ppo_loss(l2, v2, actions, old_lp, old_v, returns, nadv, clip_param=0.1)
    assert abs(float(t0) - float(tref0)) < 1e-5 and abs(float(t0) - float(t40)) > 1e-6
    with pytest.raises(NotImplementedError):
        ppo.PPO(entropy_method_name="conditional_entropy")
    # (3) an action id outside [0, A) poisons the loss
    bad = actions.clone(); bad[0, 0] = A
    _, sums = ppo.ppo_loss_raw(hv, f(bad), f(old_lp), f(old_v), f(returns), f(nadv), A)
    torch.cuda.synchronize()
    assert torch.isnan(sums[0]).item()


def test_allenact_engine_call_sequence(dev):
    """Replays what [U] AllenAct's OnPolicyTrainer does with the model: act steps (T=1, no_grad) that round-trip the
    recurrent state through ``Memory`` (step_squeeze / sampler_select as ``RolloutStorage`` uses them), then a
    ``recurrent_generator``-shaped batch -> ``PPO.loss`` -> ``backward()`` -> per-parameter ``.grad``, an
    ``optimizer.zero_grad()`` (set_to_none) and a second update -- against the oracle at every stage."""
    from embodied_clip_amd import spaces
    from embodied_clip_amd.policy import Memory, ResnetTensorObjectNavActorCritic
    from embodied_clip_amd.ppo import PPO
    T, N = 5, 4
    cfg, sd, feat, goal, h0, masks = _policy_case(T, N, C=64, S=3, H=32, seed=21)
    obs_space = spaces.Dict({"rgb_clip_resnet": spaces.Box(-1e9, 1e9, (64, 3, 3)), "goal": spaces.Discrete(12)})
    model = ResnetTensorObjectNavActorCritic(spaces.Discrete(6), obs_space, goal_sensor_uuid="goal",
                                             rgb_resnet_preprocessor_uuid="rgb_clip_resnet", hidden_size=32,
                                             state_dict=sd, device=dev)
    spec = model.recurrent_memory_specification
    (dims, dtype), = spec.values()
    sampler_dim = [d[0] for d in dims].index("sampler")
    # --- act: storage keeps memory as [step, layer, sampler, hidden]; the engine feeds step_squeeze(t) to the model
    stored = torch.zeros(T + 1, 1, N, 32)
    stored[0] = h0
    actions, logps, vals = [], [], []
    h_ref = h0.clone()
    with torch.no_grad():
        for t in range(T):
            mem = Memory().check_append("rnn", stored[t:t + 1].to(dev), sampler_dim + 1).step_squeeze(0)
            assert mem.sampler_dim("rnn") == sampler_dim and mem.tensor("rnn").shape == (1, N, 32)
            obs = {"rgb_clip_resnet": feat[t:t + 1].to(dev), "goal": goal[t:t + 1].to(dev)}
            out, mem2 = model(obs, mem, None, masks[t:t + 1].to(dev))
            lg_ref, v_ref, h_ref = opol.actor_critic_forward(feat[t:t + 1], goal[t:t + 1], h_ref, masks[t:t + 1], sd)
            assert _rel(out.distributions.probs_tensor, torch.softmax(lg_ref, -1)) < 2e-5
            a = out.distributions.sample()
            assert a.shape == (1, N)
            actions.append(a.cpu()); logps.append(out.distributions.log_prob(a).cpu()); vals.append(out.values.cpu())
            assert abs(float(logps[-1].sum()) - float(opol.categorical_log_prob(lg_ref, a.cpu()).sum())) < 1e-4
            stored[t + 1] = mem2.tensor("rnn").cpu()
            assert _rel(stored[t + 1], h_ref) < 2e-5
            # a sampler paused / dropped: memory narrows along the sampler dim
            assert mem2.sampler_select([0, 2]).tensor("rnn").shape == (1, 2, 32)
    # --- learn: one recurrent minibatch (num_mini_batch=1 == all samplers), memory = the rollout's first step
    actions = torch.cat(actions); old_lp = torch.cat(logps).unsqueeze(-1); old_v = torch.cat(vals)
    returns = old_v + 0.3; nadv = torch.randn(T, N, 1, generator=torch.Generator().manual_seed(9))
    batch = dict(actions=actions.to(dev), old_action_log_probs=old_lp.to(dev), values=old_v.to(dev),
                 returns=returns.to(dev), norm_adv_targ=nadv.to(dev), adv_targ=nadv.to(dev))
    opt = torch.optim.Adam(model.parameters(), lr=3e-4)
    sd_ref = {k: v.clone() for k, v in sd.items()}
    st = {}
    ref_batch = dict(feat=feat, goal=goal, h0=h0, masks=masks, actions=actions, old_log_probs=old_lp, old_values=old_v,
                     returns=returns, norm_adv=nadv)
    for it in range(2):
        mem = Memory().check_append("rnn", stored[0:1].to(dev), sampler_dim + 1).step_squeeze(0)
        out, _ = model({"rgb_clip_resnet": feat.to(dev), "goal": goal.to(dev)}, mem, actions.to(dev), masks.to(dev))
        total, info = PPO().loss(it, batch, out)
        opt.zero_grad()                      # recent torch: grads -> None; the flat bucket must survive this
        total.backward()
        bucket = model.handle.views(model.flat_grads)      # (re)binds every .grad into the single flat bucket
        for n, p in model.named_parameters():
            assert p.grad is not None and p.grad.data_ptr() == bucket[n].data_ptr(), n
        info_ref, _ = oppo.ppo_update_step(sd_ref, ref_batch, st, max_grad_norm=1e9)
        assert abs(info["ppo_total"] - info_ref["ppo_total"]) < 1e-4 * max(1.0, abs(info_ref["ppo_total"]))
        opt.step()
        model.ensure_flat()
    for n, p in model.named_parameters():
        upd, upd_ref = p.detach().cpu() - sd[n], sd_ref[n] - sd[n]
        assert (upd - upd_ref).abs().max() < 0.15 * 2 * 3e-4 + 1e-7, n


@pytest.mark.parametrize("T,N,bf16,S,C,H", [(5, 3, False, 3, 64, 32), (3, 5, True, 7, 64, 32), (4, 19, True, 7, 256, 512),
                                            (1, 6, True, 7, 64, 32)])
def test_dual_rgbd_policy_forward_backward_update_match_oracle(dev, T, N, bf16, S, C, H):
    """[U] ResnetDualTensorGoalEncoder (RGB + depth; readme_files/baselines_habitat.md:75): ec_policy_cfg.dual = 1 --
    two feature tensors, each stream with its own compressor / combiner, shared goal embedding, cat(rgb_x, depth_x)
    flattened into the GRU.  Act step (inference plan), learn forward, PPO loss, backward, clip + Adam vs the oracle's
    dual_goal_encoder through torch-CPU autograd; 25 parameter tensors."""
    from embodied_clip_amd import ppo
    from embodied_clip_amd.policy import PolicyHandle
    cfg = dict(in_channels=C, spatial=S, hidden=H, dual=1)
    sd = syn.policy_state_dict(11, **cfg)
    assert len(sd) == 25 and sd["state_encoder.rnn.weight_ih_l0"].shape == (3 * H, 2 * 32 * S * S)
    g = torch.Generator().manual_seed(12)
    rgb = torch.randn(T, N, C, S, S, generator=g).abs()
    depth = torch.randn(T, N, C, S, S, generator=g).abs() * 0.7
    if bf16:
        rgb, depth = rgb.to(torch.bfloat16).float(), depth.to(torch.bfloat16).float()
    goal = syn.synthetic_goals(13, (T, N))
    h0 = torch.randn(1, N, H, generator=g) * 0.5
    masks = syn.synthetic_masks(14, T, N, p_reset=0.2)
    actions, old_lp, old_v, returns, nadv = _loss_inputs(T, N, 15)
    with torch.no_grad():
        lg, vv, hf_ref = opol.actor_critic_forward((rgb, depth), goal, h0, masks, sd)
        old_lp = opol.categorical_log_prob(lg, actions).unsqueeze(-1) + 0.2 * torch.randn(T, N, 1)
        old_v = vv + 0.2 * torch.randn(T, N, 1)
    h = PolicyHandle(**cfg)
    assert len(h.offsets) == 25
    flat = h.flatten(sd, dev)
    to_rows = lambda f: (lambda r: (r.to(torch.bfloat16) if bf16 else r).to(dev))(
        f.permute(0, 1, 3, 4, 2).reshape(T * N, S * S, C).contiguous())
    r1, r2 = to_rows(rgb), to_rows(depth)
    m = masks.reshape(-1).to(dev)
    gl, h0d = goal.reshape(-1).to(dev), h0[0].contiguous().to(dev)
    # inference plan (the act step's kernels)
    wsi = torch.empty(h.workspace_bytes(T, N, False), dtype=torch.uint8, device=dev)
    hv_i, hf_i = h.forward(flat, r1, gl, h0d, m, T, N, wsi, for_backward=False, feat2=r2)
    torch.cuda.synchronize()
    assert _rel(hv_i.view(T, N, -1)[..., :6], lg) < 2e-5 and _rel(hv_i.view(T, N, -1)[..., 6:], vv) < 2e-5
    assert _rel(hf_i, hf_ref[0]) < 2e-5
    # learn pass + one optimiser step
    sd_ref = {k: v.clone() for k, v in sd.items()}
    batch = dict(feat=(rgb, depth), goal=goal, h0=h0, masks=masks, actions=actions, old_log_probs=old_lp, old_values=old_v,
                 returns=returns, norm_adv=nadv)
    info, ref_grads = oppo.ppo_update_step(sd_ref, batch, {}, lr=3e-4, max_grad_norm=0.5)
    ws = torch.empty(h.workspace_bytes(T, N, True), dtype=torch.uint8, device=dev)
    hv, _ = h.forward(flat, r1, gl, h0d, m, T, N, ws, feat2=r2)
    f = lambda t: t.reshape(-1).contiguous().to(dev)
    dhv, sums = ppo.ppo_loss_raw(hv, f(actions), f(old_lp), f(old_v), f(returns), f(nadv), 6)
    grads = torch.zeros_like(flat)
    h.backward(flat, r1, m, T, N, ws, dhv, None, grads, feat2=r2)
    torch.cuda.synchronize()
    assert torch.equal(hv, hv_i) or _rel(hv, hv_i) < 1e-5         # (the two plans sum K in different orders)
    total = ((sums[0] + 0.5 * sums[1] + 0.01 * sums[2]) / (T * N)).item()
    assert abs(total - info["ppo_total"]) < 1e-5 * max(1.0, abs(info["ppo_total"]))
    gv = h.views(grads)
    for name, gref in ref_grads.items():
        assert _rel(gv[name], gref) < 2e-4, (name, _rel(gv[name], gref))
    with pytest.raises(Exception):                                 # the depth features are mandatory for a dual handle
        h.forward(flat, r1, gl, h0d, m, T, N, ws)


def test_fused_clip_adam_is_a_torch_optimizer_drop_in(dev):
    """``FusedClipAdam`` (what ``Builder(FusedClipAdam, dict(lr=..., max_grad_norm=0.5))`` instantiates in an AllenAct config,
    INTEGRATION.md) against ``clip_grad_norm_`` + ``torch.optim.Adam`` -- the reference's ``backprop_step`` -- on (a) the
    policy module, whose parameters and grads are views of flat buckets (zero-copy path: the kernel runs on the module's own
    buffers), also after ``zero_grad()`` has dropped the grads; (b) free-standing parameters (flattened once, grads gathered);
    torch's per-parameter ``state`` / ``state_dict()`` layout; LR schedules through ``param_groups``."""
    from embodied_clip_amd import spaces
    from embodied_clip_amd.policy import ResnetTensorObjectNavActorCritic
    from embodied_clip_amd.ppo import FusedClipAdam
    cfg = dict(in_channels=64, spatial=3, hidden=32)
    sd = syn.policy_state_dict(21, **cfg)
    obs_space = spaces.Dict({"rgb_clip_resnet": spaces.Box(low=-1, high=1, shape=(64, 3, 3)), "goal": spaces.Discrete(12)})
    mk = lambda: ResnetTensorObjectNavActorCritic(spaces.Discrete(6), obs_space, goal_sensor_uuid="goal",  # noqa: E731
                                                  rgb_resnet_preprocessor_uuid="rgb_clip_resnet", hidden_size=32, state_dict=sd, device=dev)
    m_f, m_t = mk(), mk()
    opt_f = FusedClipAdam(m_f.parameters(), lr=1e-3, max_grad_norm=0.5)
    opt_t = torch.optim.Adam(m_t.parameters(), lr=1e-3)
    assert isinstance(opt_f, torch.optim.Optimizer)
    flat_ptr = m_f.flat_params.data_ptr()
    for it in range(4):
        gen = torch.Generator().manual_seed(100 + it)
        grads = {n: torch.randn(p.shape, generator=gen) * (3.0 if it % 2 == 0 else 0.01) for n, p in m_f.named_parameters()}
        for m, opt in ((m_f, opt_f), (m_t, opt_t)):
            opt.zero_grad()                                       # (grads -> None: the next bind re-creates the flat views)
            m.ensure_flat()
            for n, p in m.named_parameters():
                p.grad.copy_(grads[n].to(dev))
        torch.nn.utils.clip_grad_norm_(m_t.parameters(), 0.5)
        for o in (opt_f, opt_t):
            o.param_groups[0]["lr"] = 1e-3 * (1.0 - 0.1 * it)    # what LambdaLR does between steps
            o.step()
        gn = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).item()
        assert abs(opt_f.grad_norm() - gn) < 1e-4 * gn
    assert m_f.flat_params.data_ptr() == flat_ptr                # zero-copy: still the module's own bucket
    for (n, a), (_, b) in zip(m_f.named_parameters(), m_t.named_parameters()):
        assert (a - b).abs().max().item() < 2e-6, n
    st = opt_f.state_dict()
    assert set(st["state"][0]) >= {"step", "exp_avg", "exp_avg_sq"} and float(st["state"][0]["step"]) == 4.0
    p0 = next(iter(m_f.parameters()))
    assert opt_f.state[p0]["exp_avg"].shape == p0.shape
    for (n, a), (_, b) in zip(m_f.named_parameters(), m_t.named_parameters()):
        assert (opt_f.state[a]["exp_avg"] - opt_t.state[b]["exp_avg"]).abs().max().item() < 1e-6, n
    # (b) free-standing parameters of odd sizes
    gen = torch.Generator().manual_seed(5)
    shapes = [(7, 5), (3,), (11, 2, 3)]
    pf = [torch.nn.Parameter(torch.randn(s, generator=gen).to(dev)) for s in shapes]
    pt = [torch.nn.Parameter(p.detach().clone()) for p in pf]
    of, ot = FusedClipAdam(pf, lr=2e-3, max_grad_norm=None), torch.optim.Adam(pt, lr=2e-3)
    for it in range(3):
        for a, b in zip(pf, pt):
            g = torch.randn(a.shape, generator=gen).to(dev)
            a.grad, b.grad = g.clone(), g.clone()
        of.step(); ot.step()
    for a, b in zip(pf, pt):
        assert (a - b).abs().max().item() < 2e-6
    with pytest.raises(NotImplementedError):
        FusedClipAdam(pf, weight_decay=0.1)
    # checkpoint round trip (AllenAct saves optimizer.state_dict()): a fresh optimiser that loads it continues identically
    of2 = FusedClipAdam(pf, lr=2e-3, max_grad_norm=None)
    of2.load_state_dict(of.state_dict())
    for a, b in zip(pf, pt):
        g = torch.randn(a.shape, generator=gen).to(dev)
        a.grad, b.grad = g.clone(), g.clone()
    of2.step(); ot.step()
    for a, b in zip(pf, pt):
        assert (a - b).abs().max().item() < 2e-6
    assert float(of2.state[pf[0]]["step"]) == 4.0
    # (c) two param_groups whose tensors INTERLEAVE in the module's flat bucket (separate lr for every second tensor): a group's
    # covering span would contain the other group's tensors -- each group must step its own tensors only, once,
    # under its own lr; and a list order that differs from the storage order stays on the in-place path
    m_g, m_h = mk(), mk()
    names = [n for n, _ in m_g.named_parameters()]
    ga = [n for i, n in enumerate(names) if i % 2 == 0]
    grp = lambda m: [dict(params=[p for n, p in m.named_parameters() if n in ga], lr=1e-3),       # noqa: E731
                     dict(params=[p for n, p in m.named_parameters() if n not in ga], lr=5e-3)]
    og, oh = FusedClipAdam(grp(m_g), max_grad_norm=None), torch.optim.Adam(grp(m_h))
    for it in range(3):
        gen = torch.Generator().manual_seed(300 + it)
        grads = {n: torch.randn(p.shape, generator=gen) for n, p in m_g.named_parameters()}
        for m, opt in ((m_g, og), (m_h, oh)):
            opt.zero_grad()
            m.ensure_flat()
            for n, p in m.named_parameters():
                p.grad.copy_(grads[n].to(dev))
            opt.step()
    for (n, a), (_, b) in zip(m_g.named_parameters(), m_h.named_parameters()):
        assert (a - b).abs().max().item() < 2e-6, n
    mm = mk()
    mm.ensure_flat()
    ps = [p.data for p in mm.parameters()]
    assert FusedClipAdam._span(ps) is not None and FusedClipAdam._span(ps[::-1]) == FusedClipAdam._span(ps)
    assert FusedClipAdam._span(ps[::2]) is None                  # gaps with foreign tensors inside: not one span
    with pytest.raises(NotImplementedError):
        oh16 = FusedClipAdam([torch.nn.Parameter(torch.zeros(4, dtype=torch.float16, device=dev))], lr=1e-3)
        oh16.param_groups[0]["params"][0].grad = torch.zeros(4, dtype=torch.float16, device=dev)
        oh16.step()


def test_infer_reuse_rebuilds_tables_when_geometry_or_dtype_changed(dev):
    """EC_POLICY_INFER_REUSE promises 'the tables an EC_POLICY_INFER call left in THIS workspace are valid'.  Which tables exist
    (and where) depends on (T, N, feature dtype): a REUSE call that does not match the call that built them (fp32 features first,
    bf16 next; another N) must rebuild rather than read uninitialised tables -- and a matching one must still reuse."""
    from embodied_clip_amd.policy import PolicyHandle
    h = PolicyHandle()
    sd = syn.policy_state_dict(31)
    flat = h.flatten(sd, dev)
    N = 8
    g = torch.Generator().manual_seed(3)
    f32 = torch.randn(N, 49, 2048, generator=g).abs().to(torch.bfloat16).float().to(dev)
    b16 = f32.to(torch.bfloat16)
    goal = syn.synthetic_goals(5, (N,)).to(dev)
    h0 = (torch.randn(N, 512, generator=g) * 0.3).to(dev)
    m = torch.ones(N, device=dev)
    ws = torch.empty(h.workspace_bytes(1, 16, False), dtype=torch.uint8, device=dev)
    ref_b, _ = h.forward(flat, b16, goal, h0, m, 1, N, torch.empty_like(ws), for_backward=False)
    ref_f, _ = h.forward(flat, f32, goal, h0, m, 1, N, torch.empty_like(ws), for_backward=False)
    ws.fill_(0xFF)                                                      # NaN patterns wherever a table is not (re)built
    a, _ = h.forward(flat, f32, goal, h0, m, 1, N, ws, for_backward=False)
    b, _ = h.forward(flat, b16, goal, h0, m, 1, N, ws, for_backward=False, reuse_tables=True)      # dtype flipped: rebuild
    c, _ = h.forward(flat, b16, goal, h0, m, 1, N, ws, for_backward=False, reuse_tables=True)      # matching: reuse
    d, _ = h.forward(flat, b16[:4].contiguous(), goal[:4].contiguous(), h0[:4].contiguous(), m[:4].contiguous(), 1, 4, ws,
                     for_backward=False, reuse_tables=True)                                          # N changed: rebuild
    torch.cuda.synchronize()
    assert torch.equal(a, ref_f) and torch.equal(b, ref_b) and torch.equal(c, ref_b)
    assert torch.isfinite(d).all() and _rel(d, ref_b[:4]) < 1e-5


@pytest.mark.parametrize("T,N,S,H", [(1, 5, 3, 32), (6, 4, 3, 32), (4, 37, 7, 32), (16, 64, 7, 512)])
def test_backward_event_marks_the_recurrent_section_final(dev, T, N, S, H):
    """``ec_policy_backward3``: the event is recorded where the gradients of GRU + heads (``recurrent_section()``: one
    contiguous run of the flat bucket, weight_ih_l0 ... critic.fc.bias) are FINAL -- a stream that waits for it reads exactly
    what the finished backward holds there (bit for bit: nothing after the event may touch the section), while the goal
    encoder's section is still being written.  This is what lets the data-parallel worker start that section's all-reduce
    under the rest of the backward."""
    from embodied_clip_amd import ppo
    from embodied_clip_amd.policy import PolicyHandle
    cfg, sd, feat, goal, h0, masks = _policy_case(T, N, C=64, S=S, H=H, seed=11, bf16=True)
    actions, old_lp, old_v, returns, nadv = _loss_inputs(T, N, 9)
    h = PolicyHandle(**cfg)
    flat = h.flatten(sd, dev)
    rec = h.recurrent_section()
    names = list(h.offsets)
    o_ih, o_cb = h.offsets["state_encoder.rnn.weight_ih_l0"], h.offsets["critic.fc.bias"]
    assert rec.start == o_ih[0] and rec.stop >= o_cb[0] + o_cb[1] and rec.stop == h.flat_size      # single encoder: to the end
    assert names.index("state_encoder.rnn.weight_ih_l0") == 9 and names[-1] == "critic.fc.bias"
    assert (rec.stop - rec.start) > 0.5 * h.flat_size or H < 512
    rows = feat.permute(0, 1, 3, 4, 2).reshape(T * N, S * S, 64).contiguous().to(torch.bfloat16).to(dev)
    m = masks.reshape(-1).to(dev)
    ws = torch.empty(h.workspace_bytes(T, N, True), dtype=torch.uint8, device=dev)
    hv, _ = h.forward(flat, rows, goal.reshape(-1).to(dev), h0[0].contiguous().to(dev), m, T, N, ws)
    f = lambda t: t.reshape(-1).contiguous().to(dev)
    dhv, _ = ppo.ppo_loss_raw(hv, f(actions), f(old_lp), f(old_v), f(returns), f(nadv), 6)
    side, ev = torch.cuda.Stream(), torch.cuda.Event()
    ev.record()
    for _ in range(3):
        grads = torch.zeros_like(flat)
        torch.cuda.synchronize()
        h.backward(flat, rows, m, T, N, ws, dhv, None, grads, recurrent_ready=ev)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            early = grads[rec].clone()
        torch.cuda.synchronize()
        assert torch.equal(early, grads[rec])
        assert float(early.abs().max()) > 0 and float(grads[:rec.start].abs().max()) > 0
    # and without an event the plain entry point gives the same gradients (split-K atomics: to rounding)
    g2 = torch.zeros_like(flat)
    h.backward(flat, rows, m, T, N, ws, dhv, None, g2)
    torch.cuda.synchronize()
    assert _rel(g2, grads) < 1e-5


def test_recurrent_section_of_the_dual_encoder_stops_before_the_depth_stream(dev):
    from embodied_clip_amd.policy import PolicyHandle
    h = PolicyHandle(in_channels=64, spatial=3, hidden=32, dual=1)
    rec = h.recurrent_section()
    names = list(h.offsets)
    assert len(names) == 25
    first_depth = names[17]
    assert rec.start == h.offsets["state_encoder.rnn.weight_ih_l0"][0] and rec.stop == h.offsets[first_depth][0] < h.flat_size


@pytest.mark.parametrize("fast", ["1", "0"])
def test_learn_pass_policy_gemm_modes_match_oracle_at_pingpong_size(fast):
    """EC_POLICY_FAST (default 1: the learn pass's compressor conv over the stored features on the two leading bf16 planes of W1,
    dW1 on two planes of dc1, the backward's large gradient GEMMs on three bf16x3 products) and the fp32-exact mode (0), each in
    its own process, at T*N*49 = 37,632 rows -- where c1 really takes the 8-wave ping-pong kernel -- against the oracle's forward
    and autograd gradients at the UNCHANGED tolerances: forward 2e-5, gradients 2e-4 rel-L2 per tensor."""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "_policy_mode_check.py")], env={**os.environ, "EC_POLICY_FAST": fast},
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["mode"] == fast and out["rows"] >= 256 * 128
    assert out["logits"] < 2e-5 and out["values"] < 2e-5, out
    for name, e in out["grads"].items():
        assert e < 2e-4, (name, e, out)
    print(out)


def test_policy_act_is_forward_plus_sample(dev):
    """``ec_policy_act`` (the act step's heads launch also samples) == ``ec_policy_forward2(T = 1, inference)`` followed by
    ``ec_sample_actions``: hv, the new memory, the actions, their log-probabilities and the values bit for bit -- first call
    (tables built) and a reuse call, actor offsets keyed globally."""
    from embodied_clip_amd import _lib
    from embodied_clip_amd.policy import PolicyHandle
    lib = _lib.load()
    h = PolicyHandle()
    flat = h.flatten(syn.policy_state_dict(5), dev)
    N = 37
    g = torch.Generator().manual_seed(9)
    feat = (torch.randn(N, 49, 2048, generator=g).abs() * 0.5).to(torch.bfloat16).to(dev)
    goal = syn.synthetic_goals(6, (N,)).to(dev)
    h0 = (torch.randn(N, 512, generator=g) * 0.3).to(dev)
    m = (torch.rand(N, generator=g) > 0.2).float().to(dev)
    ws_a = torch.empty(h.workspace_bytes(1, N, False), dtype=torch.uint8, device=dev)
    ws_b = torch.empty_like(ws_a)
    for call, reuse in enumerate((False, True)):
        hv_a, hf_a = h.forward(flat, feat, goal, h0, m, 1, N, ws_a, for_backward=False, reuse_tables=reuse)
        act_a = torch.zeros(N, dtype=torch.int64, device=dev); lp_a = torch.zeros(N, device=dev); v_a = torch.zeros(N, device=dev)
        _lib.check(lib.ec_sample_actions(hv_a.data_ptr(), act_a.data_ptr(), lp_a.data_ptr(), v_a.data_ptr(), N, 6, 123, 40 + call, 1000, 0))
        hv_b = torch.empty_like(hv_a); hf_b = torch.empty_like(hf_a)
        act_b = torch.zeros(N, dtype=torch.int64, device=dev); lp_b = torch.zeros(N, device=dev); v_b = torch.zeros(N, device=dev)
        h.act(flat, feat, goal, h0, m, N, ws_b, hv_b, hf_b, act_b, lp_b, v_b, 123, 40 + call, 1000, reuse_tables=reuse)
        torch.cuda.synchronize()
        assert torch.equal(hv_a, hv_b) and torch.equal(hf_a, hf_b)
        assert torch.equal(act_a, act_b) and torch.equal(lp_a, lp_b) and torch.equal(v_a, v_b)
        assert len(set(act_a.tolist())) > 1 and (lp_a < 0).all()
