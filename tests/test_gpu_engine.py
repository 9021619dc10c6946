"""GPU parity of the whole worker iteration (engine) against the CPU oracle on identical frames/actions."""
import pytest
import torch

from embodied_clip_amd import synthetic as syn
from oracle import clip_resnet as ocr
from oracle import policy as opol
from oracle import ppo as oppo

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()


def _check_updates(pv, sd0, sd_ref, step_grads, steps, lr=3e-4):
    """Parameter updates after ``steps`` Adam steps, HIP vs oracle.  Adam's step is g / (|g| + 1e-8)-like: for an element
    whose gradient stays at the fp32 noise floor of its tensor (every step's |g| < 1e-4 of the tensor's largest, where the
    HIP gradient's ~1e-6-of-max rounding difference -- tools/diag_gru_grad.py -- is no longer small against g itself) SIZE
    AND SIGN of the step are decided by rounding, so those elements only get the bound a full sign flip can reach
    (2 lr per step); every other element must agree to 0.15 lr per step."""
    for name, pref in sd_ref.items():
        upd, upd_ref = pv[name].cpu() - sd0[name], pref - sd0[name]
        d = (upd - upd_ref).abs()
        gmax = torch.stack([g[name].abs() for g in step_grads]).amax(0)
        well = gmax > 1e-4 * gmax.max()
        assert d.max() <= 2 * steps * lr + 1e-7, (name, d.max())
        if well.any():
            assert d[well].max() < 0.15 * steps * lr + 1e-7, (name, d[well].max())


def test_worker_num_mini_batch_matches_oracle():
    """[U] allenact recurrent_generator(num_mini_batch=M): contiguous sampler ranges in shuffled order, one optimiser
    step per minibatch (update_repeats x M steps per rollout).  N = 5, M = 2 -> ranges [0,2) and [2,5): both are
    partial slices, i.e. the staging-copy path; the oracle replays the same shuffle stream."""
    import random
    from embodied_clip_amd.engine import Worker
    T, N, R, M = 3, 5, 2, 2
    enc_sd, pol_sd = syn.rn50_visual_state_dict(0), syn.policy_state_dict(0)
    w = Worker(N, T=T, device="cuda:0", seed=3, update_repeats=R, encoder_sd=enc_sd, policy_sd=pol_sd, num_mini_batch=M)
    w.collect_rollout()
    w.compute_returns()
    torch.cuda.synchronize()
    S, C = w.S, w.C
    feat_gpu = w.feat.float().cpu().view(T + 1, N, S, S, C).permute(0, 1, 4, 2, 3).contiguous()
    masks = w.env.masks.cpu().unsqueeze(-1)
    batch = dict(feat=feat_gpu[:T], goal=w.env.goals.cpu()[:T], h0=torch.zeros(1, N, w.H), masks=masks[:T],
                 actions=w.actions.cpu(), old_log_probs=w.logp.cpu().unsqueeze(-1),
                 old_values=w.values[:T].cpu().unsqueeze(-1), returns=w.returns[:T].cpu().unsqueeze(-1),
                 norm_adv=w.nadv.cpu().unsqueeze(-1))
    sd_ref = {k: v.clone() for k, v in pol_sd.items()}
    st, rng, seen, step_grads = {}, random.Random(3), [], []
    for _ in range(R):
        for (s0, s1) in oppo.recurrent_minibatch_ranges(N, M, rng):
            seen.append((s0, s1))
            info, g_ = oppo.ppo_update_step(sd_ref, oppo.slice_batch(batch, s0, s1), st)
            step_grads.append(g_)
    assert sorted(seen[:M]) == [(0, 2), (2, 5)] and st["step"] == R * M
    w.update()
    torch.cuda.synchronize()
    got = w.loss_info()
    assert abs(got["ppo_total"] - info["ppo_total"]) < 5e-4 * max(1.0, abs(info["ppo_total"]))
    _check_updates(w.policy.views(w.params), pol_sd, sd_ref, step_grads, R * M)
    assert w.opt.step_count == R * M


def test_worker_iteration_matches_oracle():
    from embodied_clip_amd.engine import Worker
    assert torch.cuda.is_available()
    T, N, R = 3, 2, 2
    enc_sd, pol_sd = syn.rn50_visual_state_dict(0), syn.policy_state_dict(0)
    w = Worker(N, T=T, device="cuda:0", seed=3, update_repeats=R, encoder_sd=enc_sd, policy_sd=pol_sd)
    w.collect_rollout()
    w.compute_returns()
    torch.cuda.synchronize()
    S, C = w.S, w.C
    # (1) encoder: every stored feature row vs the fp32 oracle on the same frames
    frames = w.env.frames.cpu()
    feat_gpu = w.feat.float().cpu().view(T + 1, N, S, S, C).permute(0, 1, 4, 2, 3).contiguous()   # [T+1,N,C,S,S]
    for t in range(T + 1):
        ref = ocr.clip_resnet_preprocessor(frames[t % frames.shape[0]], enc_sd)
        assert _rel(feat_gpu[t], ref) < 2e-2, (t, _rel(feat_gpu[t], ref))
    # (2) act steps: replay the oracle policy on the GPU's own (bf16-exact) features and actions
    masks = w.env.masks.cpu().unsqueeze(-1)
    goals = w.env.goals.cpu()
    actions = w.actions.cpu()
    assert int(actions.min()) >= 0 and int(actions.max()) < 6
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
    # (3) GAE + advantage normalisation
    rewards = w.env.rewards.cpu().unsqueeze(-1)
    Rr = oppo.compute_returns(rewards, vals, masks)
    _, nadv = oppo.normalized_advantages(Rr, vals)
    assert _rel(w.returns.unsqueeze(-1), Rr) < 1e-4
    assert _rel(w.nadv.unsqueeze(-1), nadv) < 1e-3
    # (4) update_repeats optimiser steps
    sd_ref = {k: v.clone() for k, v in pol_sd.items()}
    batch = dict(feat=feat_gpu[:T], goal=goals[:T], h0=torch.zeros(1, N, w.H), masks=masks[:T], actions=actions,
                 old_log_probs=w.logp.cpu().unsqueeze(-1), old_values=w.values[:T].cpu().unsqueeze(-1),
                 returns=w.returns[:T].cpu().unsqueeze(-1), norm_adv=w.nadv.cpu().unsqueeze(-1))
    st, step_grads = {}, []
    for _ in range(R):
        info, g_ = oppo.ppo_update_step(sd_ref, batch, st)
        step_grads.append(g_)
    w.update()
    torch.cuda.synchronize()
    got = w.loss_info()
    assert abs(got["ppo_total"] - info["ppo_total"]) < 2e-4 * max(1.0, abs(info["ppo_total"]))
    assert abs(got["grad_norm"] - info["grad_norm"]) < 2e-3 * info["grad_norm"]
    _check_updates(w.policy.views(w.params), pol_sd, sd_ref, step_grads, R)
    w.after_update()
    assert torch.equal(w.feat[0], w.feat[T])


def test_two_stream_encode_matches_one_stream():
    """Encoding the two halves of the actor batch on two HIP streams is a scheduling choice: launches of 32 instead of 64
    frames, i.e. other kernels for the late 3x3 convs (the image-resident K-split kernel from 32 frames down) with another
    fixed summation order: the same features up to fp32-accumulation rounding -- which a bf16 network amplifies to ~5e-3
    rel-L2 over its 16 blocks -- and, from the near-identical logits, the same sampled actions but
    for a few near-ties."""
    from embodied_clip_amd.engine import Worker
    enc_sd = syn.rn50_visual_state_dict(0)
    w1 = Worker(64, T=1, device="cuda:0", seed=3, update_repeats=1, encoder_sd=enc_sd, encoder_streams=1)
    w2 = Worker(64, T=1, device="cuda:0", seed=3, update_repeats=1, encoder_sd=enc_sd, encoder_streams=2)
    assert not w1.enc_streams and len(w2.enc_streams) == 2
    w1.iteration(); w2.iteration()
    torch.cuda.synchronize()
    assert _rel(w1.feat, w2.feat) <= 7e-3
    assert (w1.actions == w2.actions).float().mean().item() >= 0.9
    # the policy update is run-to-run non-deterministic at rounding level (split-K fp32 atomics) and Adam's first
    # step maps a near-zero gradient to +-lr, so parameters can only be compared to within two steps of lr = 3e-4
    assert (w1.params - w2.params).abs().max().item() <= 2 * 3e-4 + 1e-7


def test_action_synchronous_order_gives_the_same_rollout():
    """``Worker(sync_actions=True)``: the order a real vectorised env forces (actions D2H and waited for every env step,
    [U] VectorSampledTasks.step(actions)) changes the schedule only -- features, actions, log-probs, values and the
    parameters after the update are those of the free-running order, bit for bit."""
    from embodied_clip_amd.engine import Worker
    T, N = 4, 64                                        # two slices of 32
    enc_sd, pol_sd = syn.rn50_visual_state_dict(0), syn.policy_state_dict(0)
    ws = [Worker(N, T=T, device="cuda:0", seed=5, update_repeats=1, encoder_sd=enc_sd, policy_sd=pol_sd, sync_actions=s)
          for s in (False, True, "slice")]
    for w in ws:
        w.iteration()
    torch.cuda.synchronize()
    a, b, c = ws
    # ... and with one env per slice (the host waits per slice; the slices stay out of phase)
    assert torch.equal(a.actions, c.actions) and torch.equal(a.logp, c.logp) and torch.equal(a.values, c.values)
    assert torch.equal(a.feat, c.feat) and c.env._k == a.env._k
    assert b.ns == 2 and b._actions_host.shape == (N,)
    assert torch.equal(a.actions, b.actions) and torch.equal(a.logp, b.logp) and torch.equal(a.values, b.values)
    assert torch.equal(a.feat, b.feat)
    # (the policy update is run-to-run non-deterministic at rounding level -- split-K fp32 atomics in the weight-gradient
    #  GEMMs -- and Adam's first step maps a near-zero gradient to +-lr: parameters agree to within two steps of lr = 3e-4)
    assert (a.params - b.params).abs().max().item() <= 2 * 3e-4 + 1e-7
    assert torch.equal(b._actions_host, b.actions[T - 1].cpu())          # the host copy of the last step's actions


def test_set_params_invalidates_the_act_tables():
    """the act workspaces cache weight-derived tables; `set_params` / `invalidate_act_tables` drop them, so a
    parameter write outside `update()` cannot leave the act steps on stale goal-embedding / W_ih tables."""
    from embodied_clip_amd.engine import Worker
    T, N = 2, 4
    enc_sd = syn.rn50_visual_state_dict(0)
    sd_a, sd_b = syn.policy_state_dict(0), syn.policy_state_dict(1)
    wa = Worker(N, T=T, device="cuda:0", seed=5, update_repeats=1, encoder_sd=enc_sd, policy_sd=sd_a)
    wb = Worker(N, T=T, device="cuda:0", seed=5, update_repeats=1, encoder_sd=enc_sd, policy_sd=sd_b)
    wa.collect_rollout()                               # tables of sd_a are now cached in wa's act workspaces
    wa.set_params(wb.params)
    assert not any(sl.act_tables_valid for sl in wa.slices)
    wa.h.zero_(); wa.h_next.zero_(); wa.env._k = wb.env._k = 1; wa.iter = wb.iter = 0
    wa.collect_rollout(); wb.collect_rollout()
    torch.cuda.synchronize()
    assert torch.equal(wa.values, wb.values) and torch.equal(wa.logp, wb.logp)


def test_slice_streams_run_concurrently_in_every_worker_of_a_process():
    """The HIP runtime binds a stream to a hardware queue at its first submission; two streams on one queue serialise.  Before
    ``_lib.concurrent_streams`` every SECOND two-slice worker of a process ran its two encoder launches one after the other
    (48 k instead of 63 k env-frames/s -- which is what ``bench.py``'s `sync_actions` key reported for three rounds).  Every
    worker's slice streams must overlap: both busy takes about as long as one alone (2 x = serialised)."""
    import ctypes as C
    import gc
    from embodied_clip_amd import _lib
    from embodied_clip_amd.engine import Worker
    lib = _lib.load()
    for i in range(4):
        w = Worker(64, T=4, device="cuda:0", seed=i)
        assert len(w.slices) == 2
        w.iteration()
        torch.cuda.synchronize()
        r = C.c_float()
        _lib.check(lib.ec_stream_pair_overlap(w.slices[0].stream.cuda_stream, w.slices[1].stream.cuda_stream, 300, C.byref(r)))
        assert 0.5 < r.value < 1.5, (i, r.value)
        del w
        gc.collect()
    st = _lib.concurrent_streams(3, "cuda:0")
    assert len({s.cuda_stream for s in st}) == 3


def test_copy_streams_of_a_host_frame_worker_run_beside_both_slice_streams():
    """Frames in pinned host memory: two slice streams + two copy streams = the runtime's four hardware queues.  A copy stream on
    the OTHER slice's compute queue made the third worker of a process run at 42 instead of 59 k env-frames/s
    (tools/h2d_probe.py): all four come from one verified pool now."""
    import ctypes as C
    import gc
    from embodied_clip_amd import _lib
    from embodied_clip_amd.engine import Worker
    lib = _lib.load()
    for i in range(3):
        w = Worker(64, T=4, device="cuda:0", seed=i, frames_host=True, frames_u8=True)
        w.iteration()
        torch.cuda.synchronize()
        st = [sl.stream for sl in w.slices] + [sl.copy_stream for sl in w.slices]
        assert len({s.cuda_stream for s in st}) == 4
        for a in range(4):
            for b in range(a + 1, 4):
                r = C.c_float()
                _lib.check(lib.ec_stream_pair_overlap(st[a].cuda_stream, st[b].cuda_stream, 300, C.byref(r)))
                assert 0.5 < r.value < 1.5, (i, a, b, r.value)
        del w, st
        gc.collect()



@pytest.mark.parametrize("n_actors,slices", [(6, 1), (64, 2)])
def test_two_runs_from_one_seed_end_bit_identical(n_actors, slices):
    """Reproducibility (the reference's only hook is ``pl.seed_everything(1)``, primitive_probing/train.py:117): two workers built
    from one seed run ``iteration()`` twice each and must end with IDENTICAL parameters, Adam moments, rollout scalars and loss
    sums -- no floating-point atomics are left on the default path (advantage statistics, loss sums and the gradient norm fold in
    fixed orders; bias gradients, the tail's dE1 tables and the split-K weight-gradient GEMMs go through ordered partial sums).
    64 actors = two actor slices on two streams (T = 8: every learn pass has 25,088 feature rows per slice, i.e. split-K
    weight gradients, several row blocks per column sum and many partial sets in the tail's fold)."""
    from embodied_clip_amd.engine import Worker
    enc_sd, pol_sd = syn.rn50_visual_state_dict(0), syn.policy_state_dict(0)
    outs = []
    for _ in range(2):
        w = Worker(n_actors, T=8, device="cuda:0", seed=11, update_repeats=2, encoder_sd=enc_sd, policy_sd=pol_sd,
                   encoder_streams=slices)
        assert w.ns == slices
        for _it in range(2):
            w.iteration()
        torch.cuda.synchronize()
        outs.append(dict(params=w.params.clone(), m=w.opt.m.clone(), v=w.opt.v.clone(), actions=w.actions.clone(),
                         logp=w.logp.clone(), returns=w.returns.clone(), nadv=w.nadv.clone(),
                         sums=torch.stack([sl.sums for sl in w.slices]).clone(), gn=w.opt.sumsq[0].clone()))
        del w
        torch.cuda.empty_cache()
    a, b = outs
    assert (a["params"] - torch.zeros_like(a["params"])).abs().max() > 0
    for k in a:
        assert torch.equal(a[k], b[k]), (k, (a[k].double() - b[k].double()).abs().max().item())
