"""Oracle: GAE returns, PPO loss, global-norm clip + Adam.

Restates allenai/allenact (~v0.5.0; launched by the reference at
``readme_files/baselines_robothor_objectnav.md:48-51``):

  * ``allenact/algorithms/onpolicy_sync/storage.py``
    ``RolloutStorage.compute_returns`` (use_gae=True)
  * ``allenact/algorithms/onpolicy_sync/losses/ppo.py`` ``PPO.loss_per_step``
    with ``PPOConfig`` = clip_param 0.1, value_loss_coef 0.5,
    entropy_coef 0.01, use_clipped_value_loss True, normalize_advantage True
  * the experiment mixin's optimiser: ``Adam(lr=3e-4)``, max_grad_norm 0.5,
    gamma 0.99, tau(gae_lambda) 0.95, num_steps 128, update_repeats 4,
    num_mini_batch 1

  TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

from . import policy as opolicy


def compute_returns(rewards: torch.Tensor, values: torch.Tensor, masks: torch.Tensor,
                    gamma: float = 0.99, tau: float = 0.95) -> torch.Tensor:
    """GAE: rewards [T,N,1]; values [T+1,N,1] (values[T] = next_value);
    masks [T+1,N,1] (masks[t+1]=0 if the episode ended at step t).
    gae = delta + gamma*tau*m[t+1]*gae;  R[t] = gae + V[t].  -> R [T+1,N,1]."""
    T = rewards.shape[0]
    returns = torch.zeros_like(values)
    returns[T] = values[T]
    gae = torch.zeros_like(values[0])
    for t in reversed(range(T)):
        delta = rewards[t] + gamma * values[t + 1] * masks[t + 1] - values[t]
        gae = delta + gamma * tau * masks[t + 1] * gae
        returns[t] = gae + values[t]
    return returns


def normalized_advantages(returns: torch.Tensor, values: torch.Tensor, eps: float = 1e-5):
    """adv = R[:-1] - V[:-1]; (adv - mean)/(std + eps), unbiased std over the
    worker-local minibatch (num_mini_batch=1 -> the whole local rollout)."""
    adv = returns[:-1] - values[:-1]
    return adv, (adv - adv.mean()) / (adv.std() + eps)


def ppo_loss(logits: torch.Tensor, values: torch.Tensor, actions: torch.Tensor,
             old_log_probs: torch.Tensor, old_values: torch.Tensor, returns: torch.Tensor,
             norm_adv: torch.Tensor, clip_param: float = 0.1, value_loss_coef: float = 0.5,
             entropy_coef: float = 0.01, use_clipped_value_loss: bool = True):
    """``PPO.loss``.  logits [T,N,A]; values/old_values/returns/norm_adv [T,N,1];
    actions [T,N] int64; old_log_probs [T,N,1].  ``clip_param`` is the already-decayed
    ``clip_param * clip_decay(step_count)`` of upstream; ``use_clipped_value_loss=False`` is upstream's
    ``0.5 * (returns - values).pow(2)`` branch.
    Returns (total scalar, info dict of python floats)."""
    logp = opolicy.categorical_log_prob(logits, actions).unsqueeze(-1)
    ent = opolicy.categorical_entropy(logits).unsqueeze(-1)
    ratio = torch.exp(logp - old_log_probs)
    surr1 = ratio * norm_adv
    surr2 = torch.clamp(ratio, 1.0 - clip_param, 1.0 + clip_param) * norm_adv
    action_loss = -torch.where(surr2 < surr1, surr2, surr1)  # -min(surr1, surr2)
    if use_clipped_value_loss:
        v_clipped = old_values + (values - old_values).clamp(-clip_param, clip_param)
        value_loss = 0.5 * torch.max((values - returns).pow(2), (v_clipped - returns).pow(2))
    else:
        value_loss = 0.5 * (returns - values).pow(2)
    ent_loss = -ent
    la, lv, le = action_loss.mean(), value_loss.mean(), ent_loss.mean()
    total = la + value_loss_coef * lv + entropy_coef * le
    info = {"ppo_total": float(total.detach()), "value": float(lv.detach()), "action": float(la.detach()),
            "entropy": float(le.detach()), "ratio_mean": float(ratio.mean().detach())}
    return total, info


def clip_grad_norm_(grads: List[torch.Tensor], max_norm: float = 0.5) -> float:
    """torch.nn.utils.clip_grad_norm_: coef = max_norm/(total_norm+1e-6), clamped to 1."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return float(total)


def adam_step(params: List[torch.Tensor], grads: List[torch.Tensor], exp_avg: List[torch.Tensor],
              exp_avg_sq: List[torch.Tensor], step: int, lr: float = 3e-4, b1: float = 0.9, b2: float = 0.999,
              eps: float = 1e-8) -> None:
    """torch.optim.Adam (no weight decay, no amsgrad), ``step`` is 1-based:
    m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
    p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)."""
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    for p, g, m, v in zip(params, grads, exp_avg, exp_avg_sq):
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / bc1)


def ppo_update_step(sd: Dict[str, torch.Tensor], batch: Dict[str, torch.Tensor], opt_state: Dict,
                    lr: float = 3e-4, max_grad_norm: float = 0.5, grad_scale: float = 1.0):
    """One optimiser step of HOT LOOP B: policy forward over
    [T,N], PPO loss, backward, (grad *= local/global batch), clip, Adam.
    ``sd`` tensors are updated in place.  Returns (info, grads)."""
    names = list(sd.keys())
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    logits, values, _ = opolicy.actor_critic_forward(batch["feat"], batch["goal"], batch["h0"],
                                                     batch["masks"], leaves)
    total, info = ppo_loss(logits, values, batch["actions"], batch["old_log_probs"], batch["old_values"],
                           batch["returns"], batch["norm_adv"])
    grads = torch.autograd.grad(total, [leaves[k] for k in names], allow_unused=True)
    grads = [torch.zeros_like(sd[k]) if g is None else g * grad_scale for k, g in zip(names, grads)]
    raw = [g.clone() for g in grads]
    info["grad_norm"] = clip_grad_norm_(grads, max_grad_norm)
    if "step" not in opt_state:
        opt_state["step"] = 0
        opt_state["m"] = [torch.zeros_like(sd[k]) for k in names]
        opt_state["v"] = [torch.zeros_like(sd[k]) for k in names]
    opt_state["step"] += 1
    with torch.no_grad():
        adam_step([sd[k] for k in names], grads, opt_state["m"], opt_state["v"], opt_state["step"], lr=lr)
    return info, dict(zip(names, raw))


def recurrent_minibatch_ranges(num_samplers: int, num_mini_batch: int, rng) -> list:
    """[U] allenact/algorithms/onpolicy_sync/storage.py ``RolloutStorage.recurrent_generator``: samplers are split at
    ``np.round(np.linspace(0, num_samplers, num_mini_batch + 1))`` into contiguous ranges, ``random.shuffle``d;
    every minibatch keeps whole T-step sequences.  ``rng``: a ``random.Random``."""
    assert num_samplers >= num_mini_batch
    inds = [int(round(i * num_samplers / num_mini_batch)) for i in range(num_mini_batch + 1)]
    pairs = list(zip(inds[:-1], inds[1:]))
    rng.shuffle(pairs)
    return pairs


def slice_batch(batch: Dict[str, torch.Tensor], s0: int, s1: int) -> Dict[str, torch.Tensor]:
    """The columns (samplers) [s0, s1) of a [T, N, ...] rollout batch; ``h0`` is [1, N, H]."""
    out = {}
    for k, v in batch.items():
        out[k] = v[:, s0:s1].contiguous()
    return out
