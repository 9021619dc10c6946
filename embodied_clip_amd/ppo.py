"""PPO loss, GAE returns and the fused clip+Adam step over the HIP kernels.

Mirrors [U] allenai/allenact ~v0.5.0 (launched by the reference at
readme_files/baselines_robothor_objectnav.md:48-51):

  * ``PPO`` == ``allenact/algorithms/onpolicy_sync/losses/ppo.py`` ``PPO``
    (``loss(step_count, batch, actor_critic_output) -> (scalar, info)``) with
    ``PPOConfig`` defaults clip_param 0.1, value_loss_coef 0.5, entropy_coef 0.01,
    use_clipped_value_loss True, normalize_advantage True;
  * ``compute_returns`` == ``RolloutStorage.compute_returns(next_value, use_gae=True, gamma, tau)``;
  * ``FlatAdam`` == ``clip_grad_norm_(max_grad_norm)`` + ``torch.optim.Adam.step()`` as ONE launch
    over the policy's flat parameter bucket.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import _lib
from .allenact_compat import AbstractActorCriticLoss

PPOConfig = dict(clip_param=0.1, value_loss_coef=0.5, entropy_coef=0.01)


def ppo_loss_raw(hv, actions, old_logp, old_values, returns, norm_adv, A: int, clip_param=0.1, value_loss_coef=0.5,
                 entropy_coef=0.01, grad_scale: float = 1.0, dhv: Optional[torch.Tensor] = None,
                 sums: Optional[torch.Tensor] = None, use_clipped_value_loss: bool = True):
    """Fused loss forward+backward.  hv [B, A+1] fp32 contiguous; everything else flat [B].
    Returns (dhv [B, A+1], sums4 float64 device tensor = sum over B of {action, value, -entropy, ratio}).
    An action id outside [0, A) makes sums4[0] (and so the loss) NaN."""
    lib = _lib.load()
    B = hv.shape[0]
    if dhv is None:
        dhv = torch.empty_like(hv)
    if sums is None:
        sums = torch.empty(4, dtype=torch.float64, device=hv.device)
    with _lib.tensor_guard(hv):
        _lib.check(lib.ec_ppo_loss_ex(hv.data_ptr(), actions.data_ptr(), old_logp.data_ptr(), old_values.data_ptr(),
                                      returns.data_ptr(), norm_adv.data_ptr(), dhv.data_ptr(), sums.data_ptr(), B, A,
                                      clip_param, clip_param if use_clipped_value_loss else -1.0, value_loss_coef,
                                      entropy_coef, grad_scale, _lib.stream_ptr()), "ec_ppo_loss_ex")
    return dhv, sums


class _PPOLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hv, actions, old_logp, old_values, returns, norm_adv, A, clip, vc, ec, vclipped=True):
        dhv, sums = ppo_loss_raw(hv, actions, old_logp, old_values, returns, norm_adv, A, clip, vc, ec,
                                 use_clipped_value_loss=vclipped)
        B = hv.shape[0]
        total = ((sums[0] + vc * sums[1] + ec * sums[2]) / B).to(torch.float32)
        ctx.save_for_backward(dhv)
        ctx.mark_non_differentiable(sums)
        return total, sums

    @staticmethod
    def backward(ctx, gtotal, _gs):
        (dhv,) = ctx.saved_tensors
        return (dhv * gtotal,) + (None,) * 10


class PPO(AbstractActorCriticLoss):
    """Drop-in for AllenAct's ``PPO`` loss (an ``AbstractActorCriticLoss``; the real ABC when allenact is importable)."""

    def __init__(self, clip_param=0.1, value_loss_coef=0.5, entropy_coef=0.01, use_clipped_value_loss=True,
                 clip_decay=None, entropy_method_name="entropy", normalize_advantage=True, **kwargs):
        super().__init__()
        if entropy_method_name != "entropy":
            # upstream calls getattr(distributions, entropy_method_name)(); only the categorical entropy is fused
            raise NotImplementedError(f"entropy_method_name={entropy_method_name!r}: only 'entropy' is implemented")
        self.clip_param, self.value_loss_coef, self.entropy_coef = clip_param, value_loss_coef, entropy_coef
        self.use_clipped_value_loss = use_clipped_value_loss
        self.clip_decay = clip_decay if clip_decay is not None else (lambda x: 1.0)   # upstream default
        self.normalize_advantage = normalize_advantage

    def loss(self, step_count: int, batch: Dict[str, torch.Tensor], actor_critic_output, *args,
             **kwargs) -> Tuple[torch.Tensor, Dict[str, float]]:
        logits = actor_critic_output.distributions.logits
        values = actor_critic_output.values
        A = logits.shape[-1]
        B = logits.numel() // A
        hv = torch.cat([logits.reshape(B, A), values.reshape(B, 1)], dim=1).contiguous()   # autograd-visible
        adv = batch["norm_adv_targ"] if self.normalize_advantage else batch["adv_targ"]
        f = lambda t, dt=torch.float32: t.reshape(B).to(dt).contiguous()
        clip_param = self.clip_param * float(self.clip_decay(step_count))   # upstream: clip_param * clip_decay(step_count)
        total, sums = _PPOLossFn.apply(hv, f(batch["actions"], torch.int64), f(batch["old_action_log_probs"]),
                                       f(batch["values"]), f(batch["returns"]), f(adv), A, clip_param,
                                       self.value_loss_coef, self.entropy_coef, self.use_clipped_value_loss)
        s = (sums / B).tolist()
        info = {"ppo_total": float(total.detach()), "value": s[1], "action": s[0], "entropy": s[2], "ratio": s[3]}
        return total, info


def compute_returns(rewards, values, masks, gamma=0.99, tau=0.95, eps=1e-5, out=None):
    """rewards [T,N,1]; values/masks [T+1,N,1] (values[T] = next_value) ->
    (returns [T+1,N,1], adv [T,N,1], norm_adv [T,N,1])."""
    lib = _lib.load()
    T, N = rewards.shape[:2]
    dev = rewards.device
    rewards, values, masks = (t.reshape(t.shape[0], N).to(torch.float32).contiguous() for t in (rewards, values, masks))
    ret = torch.empty((T + 1, N), dtype=torch.float32, device=dev)
    adv = torch.empty((T, N), dtype=torch.float32, device=dev)
    nadv = torch.empty((T, N), dtype=torch.float32, device=dev)
    stats = torch.empty(2, dtype=torch.float64, device=dev)
    with _lib.tensor_guard(rewards):
        _lib.check(lib.ec_gae(rewards.data_ptr(), values.data_ptr(), masks.data_ptr(), ret.data_ptr(), adv.data_ptr(),
                              nadv.data_ptr(), stats.data_ptr(), T, N, gamma, tau, eps, _lib.stream_ptr()), "ec_gae")
    return ret.unsqueeze(-1), adv.unsqueeze(-1), nadv.unsqueeze(-1)


class FlatAdam:
    """Global-norm clip + Adam on one flat fp32 bucket (one launch, no host sync)."""

    def __init__(self, flat_params: torch.Tensor, lr=3e-4, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=0.5):
        self.p = flat_params
        self.m = torch.zeros_like(flat_params)
        self.v = torch.zeros_like(flat_params)
        self.sumsq = torch.zeros(1, dtype=torch.float64, device=flat_params.device)
        self.lr, self.betas, self.eps, self.max_grad_norm = lr, betas, eps, max_grad_norm
        self.step_count = 0
        self.lib = _lib.load()

    def step(self, flat_grads: torch.Tensor, lr: Optional[float] = None):
        self.step_count += 1
        with _lib.tensor_guard(self.p):
            _lib.check(self.lib.ec_clip_adam_step(self.p.data_ptr(), flat_grads.data_ptr(), self.m.data_ptr(),
                                                  self.v.data_ptr(), self.sumsq.data_ptr(), self.p.numel(),
                                                  self.max_grad_norm, self.lr if lr is None else lr, self.betas[0],
                                                  self.betas[1], self.eps, self.step_count, _lib.stream_ptr()),
                       "ec_clip_adam_step")

    def grad_norm(self) -> float:
        return float(self.sumsq.sqrt().item())


def linear_decay_lr(base_lr: float, step: int, total_steps: int) -> float:
    """``LambdaLR(LinearDecay(steps=total_steps))`` of the RoboTHOR ObjectNav mixin."""
    return base_lr * max(0.0, 1.0 - min(step, total_steps) / float(total_steps))
