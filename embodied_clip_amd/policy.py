"""Drop-in ``ResnetTensorObjectNavActorCritic`` (an AllenAct ``ActorCriticModel``)
whose forward AND backward are the hand-written gfx950 kernels behind
``ec_policy_forward`` / ``ec_policy_backward`` (include/ec_amd.h).

Mirrors [U] allenai/allenact ~v0.5.0
``projects/objectnav_baselines/models/object_nav_models.py`` (the model the
reference's RoboTHOR config instantiates:
readme_files/baselines_robothor_objectnav.md:51) -- same constructor keywords,
``forward(observations, memory, prev_actions, masks)`` contract, recurrent
memory specification and parameter names, so checkpoints,
optimisers and the engine's per-parameter all-reduce keep working.

MI355X-first differences, invisible through the API:
  * all 17 parameters are views into ONE flat fp32 buffer (and their grads into
    one flat grad buffer): a single RCCL all-reduce bucket and one fused
    clip+Adam launch instead of 17 of each;
  * the forward is autograd-visible through a ``torch.autograd.Function`` whose
    backward is the HIP backward (no torch autograd graph inside the policy).
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Any, Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib, spaces
from .synthetic import POLICY_PARAM_ORDER, policy_param_order, policy_param_shapes


# ---- AllenAct base abstractions (the real ones when allenact is importable) ----------------------
from .allenact_compat import ActorCriticModel, ActorCriticOutput, CategoricalDistr, Memory  # noqa: E402,F401


# ---- handle ----------------------------------------------------------------------------------

class PolicyHandle:
    """``ec_policy_t`` plus the flat-buffer layout."""

    def __init__(self, **cfg: int):
        self.lib = _lib.load()
        self.cfg = dict(in_channels=2048, spatial=7, hidden=512, goal_dims=32, num_goals=12, num_actions=6,
                        compress_hid=128, compress_out=32, comb_hid=128, comb_out=32, fusion=0, dual=0)
        self.cfg.update(cfg)
        self.param_order = policy_param_order(self.cfg["dual"])
        c = _lib.PolicyCfg(**self.cfg)
        h = C.c_void_p()
        _lib.check(self.lib.ec_policy_create(C.byref(h), C.byref(c)), "ec_policy_create")
        self.h = h
        self.flat_size = self.lib.ec_policy_flat_size(h)
        self.shapes = policy_param_shapes(**self.cfg)
        self.offsets: "OrderedDict[str, Tuple[int, int]]" = OrderedDict()
        assert self.lib.ec_policy_num_param_tensors(h) == len(self.param_order)
        for i, name in enumerate(self.param_order):
            off, num = C.c_size_t(), C.c_size_t()
            _lib.check(self.lib.ec_policy_param_offset(h, i, C.byref(off), C.byref(num)))
            n = 1
            for d in self.shapes[name]:
                n *= d
            assert n == num.value, (name, n, num.value)
            self.offsets[name] = (off.value, num.value)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.ec_policy_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def set_goal_table(self, table: torch.Tensor) -> None:
        """``fusion=1``: frozen goal table f32 [num_goals, in_channels] (CLIP text embeddings of the goal prompts)."""
        assert table.dtype == torch.float32 and table.is_contiguous()
        assert tuple(table.shape) == (self.cfg["num_goals"], self.cfg["in_channels"]), table.shape
        self._goal_table = table          # the handle borrows the pointer: keep the tensor alive
        _lib.check(self.lib.ec_policy_set_goal_table(self.h, table.data_ptr()), "ec_policy_set_goal_table")

    @property
    def A(self):
        return self.cfg["num_actions"]

    @property
    def H(self):
        return self.cfg["hidden"]

    def workspace_bytes(self, T: int, N: int, backward: bool) -> int:
        return self.lib.ec_policy_workspace_bytes(self.h, T, N, int(backward))

    def flatten(self, sd: Dict[str, torch.Tensor], device) -> torch.Tensor:
        flat = torch.zeros(self.flat_size, dtype=torch.float32, device=device)
        for name, (off, num) in self.offsets.items():
            flat[off:off + num].copy_(sd[name].reshape(-1).to(device=device, dtype=torch.float32))
        return flat

    def views(self, flat: torch.Tensor) -> "OrderedDict[str, torch.Tensor]":
        return OrderedDict((n, flat[o:o + k].view(self.shapes[n])) for n, (o, k) in self.offsets.items())

    def forward(self, flat_params, feat, goal, h0, masks, T, N, ws, hv=None, h_final=None, for_backward: bool = True,
                reuse_tables: bool = False, feat2=None):
        """feat: bf16 or fp32 NHWC rows [T*N, S*S, C]; returns (hv [T*N, A+1], h_final [N,H]).
        ``for_backward=True`` (default) keeps every activation ``backward`` needs and wants a workspace of
        ``workspace_bytes(T, N, True)``; ``False`` is the inference-only act step (``workspace_bytes(T, N, False)``).
        The kernel plan follows this flag, never the size of ``ws``.  ``reuse_tables`` (inference only): the weight-derived
        tables a previous ``for_backward=False`` call left in THIS ``ws`` are still valid (same parameters: the later act
        steps of a rollout)."""
        assert feat.is_contiguous() and feat.dtype in (torch.bfloat16, torch.float32)
        dev = flat_params.device
        with _lib.tensor_guard(flat_params):
            mode = 1 if for_backward else (2 if reuse_tables else 0)
            return self._forward(flat_params, feat, goal, h0, masks, T, N, ws, hv, h_final, dev, mode, feat2)

    def act(self, flat_params, feat, goal, h0, masks, N, ws, hv, h_final, actions, logp, values, seed: int, step: int,
            first_actor: int, reuse_tables: bool = False, feat2=None):
        """The act step in one call (``ec_policy_act``): the T = 1 inference forward whose heads launch also samples
        ``actions`` / ``logp`` (and copies ``values``) -- the results of ``forward(for_backward=False)`` + ``ec_sample_actions``."""
        assert feat.is_contiguous() and feat.dtype in (torch.bfloat16, torch.float32)
        with _lib.tensor_guard(flat_params):
            _lib.check(self.lib.ec_policy_act(
                self.h, flat_params.data_ptr(), feat.data_ptr(), _lib.ptr(feat2), int(feat.dtype == torch.bfloat16), goal.data_ptr(),
                h0.data_ptr(), masks.data_ptr(), N, ws.data_ptr(), ws.numel() * ws.element_size(), int(reuse_tables),
                hv.data_ptr(), h_final.data_ptr(), actions.data_ptr(), logp.data_ptr(), _lib.ptr(values), seed, step, first_actor,
                _lib.stream_ptr()), "ec_policy_act")
        return hv, h_final

    def _forward(self, flat_params, feat, goal, h0, masks, T, N, ws, hv, h_final, dev, for_backward=True, feat2=None):
        if hv is None:
            hv = torch.empty((T * N, self.A + 1), dtype=torch.float32, device=dev)
        if h_final is None:
            h_final = torch.empty((N, self.H), dtype=torch.float32, device=dev)
        if self.cfg["dual"]:       # RGB + depth: feat = the RGB features, feat2 = the depth features (same shape / dtype)
            assert feat2 is not None and feat2.is_contiguous() and feat2.dtype == feat.dtype and feat2.shape == feat.shape
        _lib.check(self.lib.ec_policy_forward2(
            self.h, flat_params.data_ptr(), feat.data_ptr(), _lib.ptr(feat2), int(feat.dtype == torch.bfloat16), goal.data_ptr(),
            h0.data_ptr(), masks.data_ptr(), T, N, ws.data_ptr(), ws.numel() * ws.element_size(), int(for_backward),
            hv.data_ptr(), h_final.data_ptr(), _lib.stream_ptr()), "ec_policy_forward2")
        return hv, h_final

    def backward(self, flat_params, feat, masks, T, N, ws, dhv, dh_final, flat_grads, feat2=None, recurrent_ready=None):
        """``recurrent_ready``: a ``torch.cuda.Event`` (already recorded once, so that its handle exists) the library records
        on the current stream when ``flat_grads[self.recurrent_section()]`` is final (``ec_policy_backward3``)."""
        with _lib.tensor_guard(flat_params):
            return self._backward(flat_params, feat, masks, T, N, ws, dhv, dh_final, flat_grads, feat2, recurrent_ready)

    def _backward(self, flat_params, feat, masks, T, N, ws, dhv, dh_final, flat_grads, feat2=None, recurrent_ready=None):
        if self.cfg["dual"]:
            assert feat2 is not None and feat2.is_contiguous() and feat2.dtype == feat.dtype and feat2.shape == feat.shape
        ev = None
        if recurrent_ready is not None:
            ev = int(recurrent_ready.cuda_event)
            assert ev, "record the event once before handing it over (torch creates the hipEvent_t lazily)"
        _lib.check(self.lib.ec_policy_backward3(
            self.h, flat_params.data_ptr(), feat.data_ptr(), _lib.ptr(feat2), int(feat.dtype == torch.bfloat16), masks.data_ptr(),
            T, N, ws.data_ptr(), ws.numel() * ws.element_size(), dhv.data_ptr(), _lib.ptr(dh_final), flat_grads.data_ptr(),
            ev, _lib.stream_ptr()), "ec_policy_backward3")
        return flat_grads

    def recurrent_section(self) -> slice:
        """The contiguous part of the flat bucket whose gradients ``ec_policy_backward3`` finishes first (GRU + both heads:
        ``state_encoder.rnn.weight_ih_l0`` ... ``critic.fc.bias``); the rest is the goal encoder's."""
        o0 = self.offsets["state_encoder.rnn.weight_ih_l0"][0]
        o1, k1 = self.offsets["critic.fc.bias"]
        end = o1 + k1
        later = [o for (o, _) in self.offsets.values() if o >= end]      # (the depth stream's tensors of the dual encoder)
        return slice(o0, min(later) if later else self.flat_size)


class _PolicyFn(torch.autograd.Function):
    """Autograd bridge: HIP forward, HIP backward.  ``owner`` (the nn.Module, or None) lends a scratch gradient
    bucket.  Workspaces come from torch's caching allocator, one per forward: a learn workspace lives exactly as long
    as the autograd node that saved it (so ``retain_graph`` / two live forwards can never share one), an act workspace
    dies with the call; reuse is the allocator's, stream-ordered.  Whether the forward keeps the activations of a
    backward is stated explicitly (``for_backward=need_grad``), so an act step always takes the act-step kernel plan."""

    @staticmethod
    def forward(ctx, handle: PolicyHandle, owner, flat, feat, feat2, goal, h0, masks, T, N, *params):
        need_grad = any(ctx.needs_input_grad)   # (grad mode is always off inside Function.forward)
        ws = torch.empty(handle.workspace_bytes(T, N, need_grad), dtype=torch.uint8, device=flat.device)
        hv, h_final = handle.forward(flat, feat, goal, h0, masks, T, N, ws, for_backward=need_grad, feat2=feat2)
        if need_grad:
            ctx.handle, ctx.owner, ctx.T, ctx.N = handle, owner, T, N
            ctx.feat2 = feat2                    # (the depth stream's features of the dual encoder, or None: frozen input)
            ctx.save_for_backward(flat, feat, masks, ws)
        return hv, h_final

    @staticmethod
    def backward(ctx, dhv, dh_final):
        flat, feat, masks, ws = ctx.saved_tensors
        h: PolicyHandle = ctx.handle
        owner = ctx.owner
        # the scratch bucket may only be lent when every .grad is already bound (autograd then accumulates INTO the
        # bound views and never keeps a reference to what is returned here)
        lend = owner is not None and all(p.grad is not None for p in owner.parameters())
        if lend:
            if owner._g_scratch is None or owner._g_scratch.device != flat.device:
                owner._g_scratch = torch.empty_like(flat)
            g = owner._g_scratch.zero_()
        else:
            g = torch.zeros_like(flat)
        dhv = dhv.contiguous() if dhv is not None else torch.zeros((ctx.T * ctx.N, h.A + 1), device=flat.device)
        dhf = dh_final.contiguous() if dh_final is not None else None
        h.backward(flat, feat, masks, ctx.T, ctx.N, ws, dhv, dhf, g, feat2=ctx.feat2)
        grads = tuple(g[o:o + k].view(h.shapes[n]) for n, (o, k) in h.offsets.items())
        return (None,) * 10 + grads


class _Holder(nn.Module):
    """Creates parameter sub-module paths like ``goal_visual_encoder.resnet_compressor.0.weight``."""


def _set_nested(root: nn.Module, dotted: str, p: nn.Parameter):
    parts = dotted.split(".")
    m = root
    for part in parts[:-1]:
        if not hasattr(m, part):
            m.add_module(part, _Holder())
        m = getattr(m, part)
    m.register_parameter(parts[-1], p)


class ResnetTensorObjectNavActorCritic(ActorCriticModel):
    """[U] ``ResnetTensorObjectNavActorCritic(action_space, observation_space, goal_sensor_uuid,
    rgb_resnet_preprocessor_uuid=None, depth_resnet_preprocessor_uuid=None, hidden_size=512, goal_dims=32,
    resnet_compressor_hidden_out_dims=(128, 32), combiner_hidden_out_dims=(128, 32))`` -- an
    ``allenact...policy.ActorCriticModel`` (the real ABC when allenact is importable).

    Exactly one of the two preprocessor uuids selects the single-tower ``ResnetTensorGoalEncoder`` (RGB for the CLIP
    configs of the reference; a depth-only tower is the same arithmetic on the depth features).  Both at once is
    upstream's ``ResnetDualTensorGoalEncoder`` (RGB-D: readme_files/baselines_habitat.md:75 "replace rgb with rgbd"):
    each stream has its own compressor and combiner (parameter names ``goal_visual_encoder.{rgb,depth}_resnet_compressor.*``
    / ``{rgb,depth}_target_obs_combiner.*``), the goal embedding is shared, and ``cat([rgb_x, depth_x], dim=1)`` is
    flattened into the GRU (``ec_policy_cfg.dual``).
    """

    def __init__(self, action_space, observation_space, goal_sensor_uuid: str,
                 rgb_resnet_preprocessor_uuid: Optional[str] = None, depth_resnet_preprocessor_uuid: Optional[str] = None,
                 hidden_size: int = 512, goal_dims: int = 32, resnet_compressor_hidden_out_dims=(128, 32),
                 combiner_hidden_out_dims=(128, 32), state_dict: Optional[Dict[str, torch.Tensor]] = None,
                 device="cuda"):
        super().__init__(action_space=action_space, observation_space=observation_space)
        if rgb_resnet_preprocessor_uuid is None and depth_resnet_preprocessor_uuid is None:
            raise ValueError("one of rgb_resnet_preprocessor_uuid / depth_resnet_preprocessor_uuid is required")
        self.goal_uuid = goal_sensor_uuid
        self.dual = rgb_resnet_preprocessor_uuid is not None and depth_resnet_preprocessor_uuid is not None
        self.resnet_uuid = (rgb_resnet_preprocessor_uuid if rgb_resnet_preprocessor_uuid is not None
                            else depth_resnet_preprocessor_uuid)
        self.depth_uuid = depth_resnet_preprocessor_uuid if self.dual else None
        self._hidden_size = hidden_size
        self._g_scratch: Optional[torch.Tensor] = None
        if self.resnet_uuid not in observation_space.spaces:
            raise NotImplementedError("blind agent (no visual tensor in the observation space) is not built")
        if self.dual and self.depth_uuid not in observation_space.spaces:
            raise ValueError(f"{self.depth_uuid!r} is not in the observation space")
        rs = observation_space.spaces[self.resnet_uuid].shape        # (C, S, S)
        if self.dual and tuple(observation_space.spaces[self.depth_uuid].shape) != tuple(rs):
            raise ValueError("the RGB and depth feature tensors must have the same shape")
        num_goals = getattr(observation_space.spaces[self.goal_uuid], "n", 12)
        self.handle = PolicyHandle(in_channels=rs[0], spatial=rs[1], hidden=hidden_size, goal_dims=goal_dims,
                                   num_goals=num_goals, num_actions=action_space.n,
                                   compress_hid=resnet_compressor_hidden_out_dims[0],
                                   compress_out=resnet_compressor_hidden_out_dims[1],
                                   comb_hid=combiner_hidden_out_dims[0], comb_out=combiner_hidden_out_dims[1],
                                   dual=int(self.dual))
        dev = torch.device(device)
        if state_dict is None:
            from .synthetic import policy_state_dict
            state_dict = policy_state_dict(0, **self.handle.cfg)
        self._flat = self.handle.flatten(state_dict, dev)
        self._flat_grad = torch.zeros_like(self._flat)
        for name, v in self.handle.views(self._flat).items():
            _set_nested(self, name, nn.Parameter(v))
        self._bind_grads()

    # -- flat bucket maintenance -----------------------------------------------------------------
    def _named(self):
        """[(name, Parameter)] in the handle's order.  Cached: ``named_parameters()`` walks the module tree (~0.2 ms of host
        time per call, three calls per forward -- it showed in the act step of the plugin route); the Parameter OBJECTS are
        stable (``.to()`` / ``load_state_dict`` change ``.data`` in place), and ``_apply`` / ``register_parameter`` drop the
        cache in case a caller replaces them."""
        c = self.__dict__.get("_named_cache")
        if c is None:
            d = dict(self.named_parameters())
            c = self.__dict__["_named_cache"] = [(n, d[n]) for n in self.handle.param_order]
        return c

    def _apply(self, fn, *args, **kwargs):
        self.__dict__["_named_cache"] = None
        return super()._apply(fn, *args, **kwargs)

    def register_parameter(self, name, param):
        self.__dict__["_named_cache"] = None
        return super().register_parameter(name, param)

    def _bind_grads(self):
        """(Re)bind every ``p.grad`` to its view of the flat gradient bucket.  ``optimizer.zero_grad()`` of recent
        torch sets grads to None: the view is then zeroed and bound again; a foreign grad tensor is copied in."""
        for (n, p), (_, g) in zip(self._named(), self.handle.views(self._flat_grad).items()):
            if p.grad is None:
                g.zero_()
            elif p.grad.data_ptr() != g.data_ptr():
                g.copy_(p.grad)
            else:
                continue
            p.grad = g

    def ensure_flat(self):
        """Re-establish 'parameters are views of one flat buffer' after .to()/.cuda()/load."""
        ok = all(p.data_ptr() == self._flat.data_ptr() + 4 * off and p.device == self._flat.device
                 for (n, p), (off, _) in zip(self._named(), self.handle.offsets.values()))
        if ok:
            self._bind_grads()
            return
        dev = self._named()[0][1].device
        self._flat = self.handle.flatten({n: p.data for n, p in self._named()}, dev)
        self._flat_grad = torch.zeros_like(self._flat)
        for (n, p), v in zip(self._named(), self.handle.views(self._flat).values()):
            p.data = v
        self._bind_grads()

    @property
    def flat_params(self) -> torch.Tensor:
        self.ensure_flat()
        return self._flat

    @property
    def flat_grads(self) -> torch.Tensor:
        self.ensure_flat()
        return self._flat_grad

    # -- learn-pass feature rows ------------------------------------------------------------------
    def _bf16_rows(self, feat: torch.Tensor, T: int, N: int, slot: int = 0):
        """Learn pass over the reference's fp32 NCHW rollout storage: ``ClipResNetPreprocessor.process`` returns the
        trunk's bf16 features widened to fp32, so the storage holds values that ARE bf16 -- for such a tensor the rows
        go to the kernels as bf16 NHWC (half the bytes, the 8-wave compressor kernel and the transpose-read weight-gradient
        kernel instead of the generic fp32-operand GEMMs; the products are the same numbers).  ``ec_nchw_f32_to_nhwc_bf16``
        checks exactness while it converts; anything else keeps the fp32 path (returns None).  The converted rows are kept
        while the SAME storage (base tensor, view geometry, version counter) comes back: the update epochs of a rollout."""
        import weakref
        if (feat.dtype != torch.float32 or not feat.is_cuda or not feat.is_contiguous() or feat.requires_grad
                or feat.dim() != 5 or self.handle.cfg["in_channels"] % 64 != 0):
            return None
        c = self.handle.cfg
        if c["spatial"] ** 2 * 65 * 4 > 64 * 1024:       # the conversion kernel's LDS tile (EC_ERR_SHAPE beyond): fp32 path
            return None
        base = feat._base if feat._base is not None else feat
        try:                                             # (inference-mode tensors have no version counter)
            version = base._version
        except RuntimeError:
            return None
        # NOTE the cache key relies on the storage's VERSION COUNTER: the storage must be filled through torch ops (or through
        # RN50Trunk.to_nchw_f32(out=...), which bumps the counter itself) -- a raw-pointer write from foreign code would
        # leave stale rows here
        key = (feat.data_ptr(), tuple(feat.shape), tuple(feat.stride()), version)
        cache = getattr(self, "_rows_cache", None)
        if cache is None:
            cache = self._rows_cache = {}
        hit = cache.get(slot)
        if hit is not None and hit[0]() is base and hit[1] == key:
            return hit[2]
        rows = torch.empty((T * N, c["spatial"] ** 2, c["in_channels"]), dtype=torch.bfloat16, device=feat.device)
        flag = torch.zeros(1, dtype=torch.int32, device=feat.device)
        with _lib.tensor_guard(feat):
            rc = self.handle.lib.ec_nchw_f32_to_nhwc_bf16(feat.data_ptr(), rows.data_ptr(), T * N, c["spatial"] ** 2,
                                                          c["in_channels"], flag.data_ptr(), _lib.stream_ptr())
        if rc != 0:                                            # a geometry the fast path does not take: the fp32 path serves it
            cache[slot] = (weakref.ref(base), key, None)
            return None
        out = rows if int(flag.item()) == 0 else None          # (one host sync per rollout storage version)
        cache[slot] = (weakref.ref(base), key, out)
        return out

    # -- ActorCriticModel surface ----------------------------------------------------------------
    @property
    def recurrent_hidden_state_size(self) -> int:
        return self._hidden_size

    @property
    def num_recurrent_layers(self) -> int:
        return 1

    @property
    def is_blind(self) -> bool:
        """True if the model has no visual input ([U] ``goal_visual_encoder.is_blind``); never, once constructed."""
        return False

    def _recurrent_memory_specification(self):
        return dict(rnn=((("layer", self.num_recurrent_layers), ("sampler", None),
                          ("hidden", self.recurrent_hidden_state_size)), torch.float32))

    def forward(self, observations: Dict[str, torch.Tensor], memory: Memory, prev_actions: torch.Tensor,
                masks: torch.Tensor):
        """observations[resnet_uuid]: [T,N,C,S,S] fp32 (or bf16/fp32 channels-last [T,N,S,S,C] from
        ``ClipResNetPreprocessor.process_bf16_nhwc``); observations[goal_uuid]: [T,N] int;
        memory 'rnn': [1,N,H]; masks: [T,N,1].  -> (ActorCriticOutput, Memory)."""
        self.ensure_flat()
        goal = observations[self.goal_uuid]
        T, N = masks.shape[:2]
        c = self.handle.cfg

        def to_rows(feat, slot=0):
            if feat.shape[-1] == c["in_channels"] and feat.shape[-2] == c["spatial"]:
                rows = feat.reshape(T * N, c["spatial"] ** 2, c["in_channels"]).contiguous()
            else:   # the reference's NCHW layout -> NHWC rows (lossless re-layout)
                fast = self._bf16_rows(feat, T, N, slot) if (T > 1 and torch.is_grad_enabled()) else None
                if fast is not None:
                    return fast
                rows = feat.reshape(T * N, c["in_channels"], -1).transpose(1, 2).contiguous()
            return rows if rows.dtype in (torch.bfloat16, torch.float32) else rows.float()

        rows = to_rows(observations[self.resnet_uuid])
        rows2 = None
        if self.dual:
            rows2 = to_rows(observations[self.depth_uuid], 1)
            if rows2.dtype != rows.dtype:
                rows, rows2 = rows.float(), rows2.float()
        goal = goal.reshape(T * N).to(torch.int64).contiguous()
        h0 = memory.tensor("rnn").reshape(N, self._hidden_size).to(torch.float32).contiguous()
        m = masks.reshape(T * N).to(torch.float32).contiguous()
        if torch.is_grad_enabled():
            params = [p for _, p in self._named()]
            hv, h_final = _PolicyFn.apply(self.handle, self, self._flat, rows, rows2, goal, h0, m, T, N, *params)
        else:   # act steps (the engine's no_grad rollout): straight to the inference plan, no autograd node, no 17-tensor argument list
            ws = torch.empty(self.handle.workspace_bytes(T, N, False), dtype=torch.uint8, device=self._flat.device)
            hv, h_final = self.handle.forward(self._flat, rows, goal, h0, m, T, N, ws, for_backward=False, feat2=rows2)
        A = self.handle.A
        hv = hv.view(T, N, A + 1)
        out = ActorCriticOutput(distributions=CategoricalDistr(logits=hv[..., :A]), values=hv[..., A:], extras={})
        return out, memory.set_tensor("rnn", h_final.view(1, N, self._hidden_size))
