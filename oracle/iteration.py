"""Oracle: one full DD-PPO worker iteration on the host CPU (the CPU baseline).

Same structure as the product's ``embodied_clip_amd.engine.Worker.iteration``
and as [U] AllenAct ``OnPolicyTrainer``: T x [encode N frames +
policy act step + sample], GAE, update_repeats x [policy forward over [T,N],
PPO loss, backward, clip, Adam] -- all in torch-CPU fp32 through the oracle
restatements.  Used by tests (tiny sizes) and by ``bench.py``'s ``cpu_baseline``
leg (bounded sample).  TEST INFRASTRUCTURE ONLY -- never imported by the product.
"""
from __future__ import annotations

import time
from typing import Dict

import torch

from . import clip_resnet as ocr
from . import policy as opol
from . import ppo as oppo


def run_iteration(enc_sd, pol_sd, frames: torch.Tensor, goals: torch.Tensor, masks: torch.Tensor,
                  rewards: torch.Tensor, T: int, N: int, update_repeats: int = 4, seed: int = 0,
                  opt_state=None, num_mini_batch: int = 1, mb_rng=None) -> Dict[str, float]:
    """frames: fp32 NHWC [P, N, R, R, 3] pool (cycled); goals [T+1,N]; masks [T+1,N,1]; rewards [T,N,1].
    pol_sd is updated in place.  Returns timing + loss info."""
    g = torch.Generator().manual_seed(seed)
    H = pol_sd["state_encoder.rnn.weight_hh_l0"].shape[1]
    t0 = time.perf_counter()
    feats, actions, logps, values = [], [], [], []
    h = torch.zeros(1, N, H)
    h_start = h.clone()
    k = 0
    with torch.no_grad():
        feats.append(ocr.clip_resnet_preprocessor(frames[k % frames.shape[0]], enc_sd)); k += 1
        for t in range(T):
            lg, v, h = opol.actor_critic_forward(feats[t][None], goals[t][None], h, masks[t][None], pol_sd)
            a = torch.multinomial(torch.softmax(lg[0], -1), 1, generator=g).squeeze(-1)
            actions.append(a); logps.append(opol.categorical_log_prob(lg, a[None])[0]); values.append(v[0])
            feats.append(ocr.clip_resnet_preprocessor(frames[k % frames.shape[0]], enc_sd)); k += 1
        _, v, _ = opol.actor_critic_forward(feats[T][None], goals[T][None], h, masks[T][None], pol_sd)
        values.append(v[0])
        values = torch.stack(values)                       # [T+1, N, 1]
        returns = oppo.compute_returns(rewards, values, masks)
        _, nadv = oppo.normalized_advantages(returns, values)
    t_rollout = time.perf_counter() - t0
    batch = dict(feat=torch.stack(feats[:T]), goal=goals[:T], h0=h_start, masks=masks[:T],
                 actions=torch.stack(actions), old_log_probs=torch.stack(logps).unsqueeze(-1),
                 old_values=values[:T], returns=returns[:T], norm_adv=nadv)
    opt_state = {} if opt_state is None else opt_state
    info = {}
    if num_mini_batch > 1 and mb_rng is None:
        import random
        mb_rng = random.Random(seed)
    for _ in range(update_repeats):
        if num_mini_batch == 1:
            info, _ = oppo.ppo_update_step(pol_sd, batch, opt_state)
        else:
            for (s0, s1) in oppo.recurrent_minibatch_ranges(N, num_mini_batch, mb_rng):
                info, _ = oppo.ppo_update_step(pol_sd, oppo.slice_batch(batch, s0, s1), opt_state)
    dt = time.perf_counter() - t0
    info.update(seconds=dt, seconds_rollout=t_rollout, frames=T * N, frames_per_s=T * N / dt)
    return info
