"""PPO loss, GAE returns and the fused clip+Adam step over the HIP kernels.

Mirrors [U] allenai/allenact ~v0.5.0 (launched by the reference at
readme_files/baselines_robothor_objectnav.md:48-51):

  * ``PPO`` == ``allenact/algorithms/onpolicy_sync/losses/ppo.py`` ``PPO``
    (``loss(step_count, batch, actor_critic_output) -> (scalar, info)``) with
    ``PPOConfig`` defaults clip_param 0.1, value_loss_coef 0.5, entropy_coef 0.01,
    use_clipped_value_loss True, normalize_advantage True;
  * ``compute_returns`` == ``RolloutStorage.compute_returns(next_value, use_gae=True, gamma, tau)``;
  * ``FlatAdam`` == ``clip_grad_norm_(max_grad_norm)`` + ``torch.optim.Adam.step()`` as ONE launch
    over the policy's flat parameter bucket; ``FusedClipAdam`` is the same launch behind the
    ``torch.optim.Optimizer`` interface (what an AllenAct ``optimizer_builder`` instantiates).
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
        self.lib = _lib.load()
        # scratch of the norm reduction ([0] = ||g||^2 after a step; block partials and a ticket counter behind it)
        self.sumsq = torch.zeros(self.lib.ec_clip_adam_scratch_doubles(), dtype=torch.float64, device=flat_params.device)
        self.lr, self.betas, self.eps, self.max_grad_norm = lr, betas, eps, max_grad_norm
        self.step_count = 0

    def step(self, flat_grads: torch.Tensor, lr: Optional[float] = None):
        self.step_count += 1
        with _lib.tensor_guard(self.p):
            _lib.check(self.lib.ec_clip_adam_step(self.p.data_ptr(), flat_grads.data_ptr(), self.m.data_ptr(),
                                                  self.v.data_ptr(), self.sumsq.data_ptr(), self.p.numel(),
                                                  self.max_grad_norm, self.lr if lr is None else lr, self.betas[0],
                                                  self.betas[1], self.eps, self.step_count, _lib.stream_ptr()),
                       "ec_clip_adam_step")

    def grad_norm(self) -> float:
        return float(self.sumsq[0].sqrt().item())


class FusedClipAdam(torch.optim.Optimizer):
    """``torch.optim.Adam`` drop-in (same ``param_groups`` / ``lr`` / ``betas`` / ``eps`` keys, per-parameter ``state`` with
    ``step`` / ``exp_avg`` / ``exp_avg_sq`` so that ``state_dict()`` checkpoints keep torch's layout) whose ``step()`` is ONE
    ``ec_clip_adam_step`` launch per parameter group -- global-norm gradient clipping (``max_grad_norm``; ``None`` = off, e.g.
    when the engine has already called ``clip_grad_norm_``) + Adam over a flat fp32 bucket, no host synchronisation.

    [U] AllenAct builds its optimiser from the experiment config (``optimizer_builder=Builder(optim.Adam, dict(lr=lr))``,
    the RoboTHOR ObjectNav mixin behind readme_files/baselines_robothor_objectnav.md:51) and steps it in
    ``OnPolicyTrainer.backprop_step`` after ``nn.utils.clip_grad_norm_``; the swap is that one ``Builder`` line
    (INTEGRATION.md).  Parameters that already are views of one flat buffer -- ``ResnetTensorObjectNavActorCritic`` keeps
    its 17 tensors and their ``.grad`` s that way -- are used IN PLACE (zero copies); any other parameter set is flattened
    once (``p.data`` re-pointed into a new flat buffer) and its gradients are gathered into a flat scratch per step.
    Not implemented (raises): ``weight_decay != 0``, ``amsgrad``, sparse or non-fp32 parameters, a group in which only some
    parameters have a gradient."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, max_grad_norm=None):
        if weight_decay != 0.0 or amsgrad:
            raise NotImplementedError("FusedClipAdam: weight_decay / amsgrad are not implemented (AllenAct's PPO configs use neither)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0.0, amsgrad=False, max_grad_norm=max_grad_norm))
        self.lib = _lib.load()
        self._buckets: Dict[int, dict] = {}

    def load_state_dict(self, state_dict):
        """torch's loader replaces the per-parameter state tensors: drop the flat buckets, the next ``step()`` rebuilds them
        from the restored ``exp_avg`` / ``exp_avg_sq`` (and re-binds the state entries as views of the new flat moments)."""
        super().load_state_dict(state_dict)
        self._buckets = {}

    @staticmethod
    def _span(tensors, max_gap: int = 3):
        """(storage offset of the first element, numel of the covering span) when ``tensors`` are non-overlapping contiguous
        fp32 views of ONE storage that lie ADJACENT to one another (in any list order; gaps of at most ``max_gap`` elements,
        the 16-byte alignment padding of the flat buckets); else None.  Adjacency matters: the kernel updates the whole
        covering span, so a span with another group's tensors inside it would step those under this group's lr as well."""
        st = tensors[0].untyped_storage().data_ptr()
        for t in tensors:
            if t.dtype != torch.float32 or not t.is_contiguous() or t.untyped_storage().data_ptr() != st:
                return None
        lo, end = None, None
        for t in sorted(tensors, key=lambda x: x.storage_offset()):
            o = t.storage_offset()
            if end is not None and (o < end or o - end > max_gap):
                return None
            lo = o if lo is None else lo
            end = o + t.numel()
        return lo, end - lo

    @staticmethod
    def _flat_view(t: torch.Tensor, off: int, n: int) -> torch.Tensor:
        return torch.empty(0, dtype=torch.float32, device=t.device).set_(t.untyped_storage(), off, (n,), (1,))

    def _bucket(self, gi: int, ps):
        b = self._buckets.get(gi)
        if b is not None and b["ptrs"] == tuple(p.data_ptr() for p in ps):
            return b
        if any(p.dtype != torch.float32 for p in ps):
            raise NotImplementedError("FusedClipAdam: non-fp32 parameters")
        span = self._span([p.data for p in ps])
        if span is None:        # flatten once: the parameters become views of one new buffer (values preserved)
            flat = torch.cat([p.data.reshape(-1) for p in ps])
            o = 0
            for p in ps:
                p.data = flat[o:o + p.numel()].view(p.shape)
                o += p.numel()
            span = (0, flat.numel())
        off, n = span
        flat = self._flat_view(ps[0].data, off, n)
        m, v = torch.zeros_like(flat), torch.zeros_like(flat)
        rel = [p.data.storage_offset() - off for p in ps]
        for p, r in zip(ps, rel):
            stt = self.state[p]
            if "exp_avg" in stt and stt["exp_avg"].numel() == p.numel():      # resumed from a torch-format checkpoint
                m[r:r + p.numel()].copy_(stt["exp_avg"].reshape(-1))
                v[r:r + p.numel()].copy_(stt["exp_avg_sq"].reshape(-1))
            stt["exp_avg"], stt["exp_avg_sq"] = m[r:r + p.numel()].view(p.shape), v[r:r + p.numel()].view(p.shape)
            stt.setdefault("step", torch.tensor(0.0))
        b = self._buckets[gi] = dict(ptrs=tuple(p.data_ptr() for p in ps), off=off, n=n, flat=flat, m=m, v=v, rel=rel,
                                     sumsq=torch.zeros(self.lib.ec_clip_adam_scratch_doubles(), dtype=torch.float64, device=flat.device),
                                     gscratch=None)
        return b

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.numel() > 0]
            if not ps:
                continue
            if any(p.grad is None for p in ps):
                if all(p.grad is None for p in ps):
                    continue
                raise NotImplementedError("FusedClipAdam: a parameter group with gradients on only some of its parameters")
            if any(p.grad.is_sparse for p in ps):
                raise NotImplementedError("FusedClipAdam: sparse gradients")
            b = self._bucket(gi, ps)
            gs = [p.grad for p in ps]
            gspan = self._span(gs)
            if gspan is not None and gspan[1] == b["n"] and all(g.storage_offset() - gspan[0] == r for g, r in zip(gs, b["rel"])):
                gflat = self._flat_view(gs[0], gspan[0], gspan[1])          # the module's flat gradient bucket, in place
            else:                                                           # gather (plumbing copies; the arithmetic stays in the kernel)
                if b["gscratch"] is None:
                    b["gscratch"] = torch.zeros_like(b["flat"])
                gflat = b["gscratch"]
                for g, r in zip(gs, b["rel"]):
                    gflat[r:r + g.numel()].copy_(g.reshape(-1))
            step = int(self.state[ps[0]]["step"]) + 1
            mg = group.get("max_grad_norm")
            with _lib.tensor_guard(b["flat"]):
                _lib.check(self.lib.ec_clip_adam_step(b["flat"].data_ptr(), gflat.data_ptr(), b["m"].data_ptr(), b["v"].data_ptr(),
                                                      b["sumsq"].data_ptr(), b["n"], float(mg) if mg else 0.0, float(group["lr"]),
                                                      group["betas"][0], group["betas"][1], group["eps"], step, _lib.stream_ptr()),
                           "ec_clip_adam_step")
            for p in ps:
                self.state[p]["step"] = torch.tensor(float(step))
        return loss

    def grad_norm(self, group: int = 0) -> float:
        """Global gradient norm the last ``step()`` of ``group`` saw (one host sync; for logging)."""
        return float(self._buckets[group]["sumsq"][0].sqrt().item())


def linear_decay_lr(base_lr: float, step: int, total_steps: int) -> float:
    """``LambdaLR(LinearDecay(steps=total_steps))`` of the RoboTHOR ObjectNav mixin."""
    return base_lr * max(0.0, 1.0 - min(step, total_steps) / float(total_steps))
