"""Drop-in ``ClipResNetPreprocessor`` / ``ClipViTPreprocessor``.

Same constructor keywords, attributes and ``process`` / ``to`` behaviour as
[U] ``allenact_plugins/clip_plugin/clip_preprocessors.py`` (the plugin the
reference installs: readme_files/baselines_robothor_objectnav.md:25, used by
the experiment config named at :51), so RoboTHOR/Habitat experiment configs
keep working unchanged; the arithmetic behind ``process`` is the hand-written
gfx950 HIP path (include/ec_amd.h) instead of ``clip.load(...).visual``
(reference call site: primitive_probing/generate_data/thor_image_features.py:57-67).

Weights: ``clip`` / network access are unavailable in the build image, so the
visual tower's ``state_dict`` (OpenAI key layout) is supplied via
``state_dict=`` / ``weights_path=`` / ``$EC_CLIP_WEIGHTS_DIR/<type>.pt``; if the
``clip`` package is importable it is used exactly as the reference does.
"""
from __future__ import annotations

import os
from typing import Any, Dict, List, Optional

import numpy as np
import torch

from . import _lib, spaces
from .allenact_compat import Preprocessor


def load_checkpoint_state_dict(path: str) -> Dict[str, torch.Tensor]:
    """Read a CLIP checkpoint the way ``clip.load`` does ([U] openai/CLIP clip/clip.py): the published files
    (``RN50.pt``, ``ViT-B-32.pt``) are TorchScript archives -> ``torch.jit.load(...).state_dict()``; anything else
    is tried as a plain ``state_dict`` file (``torch.load``, tensors only)."""
    try:
        return dict(torch.jit.load(path, map_location="cpu").eval().state_dict())
    except RuntimeError:
        return dict(torch.load(path, map_location="cpu", weights_only=True))   # a dict of tensors, nothing unpickled beyond that


def visual_state_dict(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """The visual tower's entries of a whole-CLIP ``state_dict`` with the ``visual.`` prefix removed (a dict that
    already is a visual tower is returned unchanged).  Weights may be fp16 (CLIP's published checkpoints are):
    the packers fold / round from fp32 copies."""
    if any(k.startswith("visual.") for k in sd):
        return {k[len("visual."):]: v for k, v in sd.items() if k.startswith("visual.")}
    return sd


def _load_visual_state_dict(clip_model_type: str, state_dict, weights_path):
    if state_dict is not None:
        return visual_state_dict(state_dict)
    cand = weights_path
    if cand is None and os.environ.get("EC_CLIP_WEIGHTS_DIR"):
        cand = os.path.join(os.environ["EC_CLIP_WEIGHTS_DIR"], clip_model_type.replace("/", "-") + ".pt")
    if cand is not None and os.path.exists(cand):
        return visual_state_dict(load_checkpoint_state_dict(cand))
    try:  # exactly what the reference does (thor_image_features.py:57,59)
        import clip  # type: ignore
        model, _ = clip.load(clip_model_type, device="cpu")
        return model.visual.float().state_dict()
    except ImportError as e:
        raise RuntimeError(
            f"No CLIP weights for {clip_model_type!r}: pass state_dict=/weights_path=, set EC_CLIP_WEIGHTS_DIR, "
            "or install openai/CLIP") from e


class _PreprocessorBase(Preprocessor):
    """An ``allenact.base_abstractions.preprocessor.Preprocessor`` (the real ABC when allenact is importable,
    ``allenact_compat``'s restatement otherwise)."""

    CLIP_RGB_MEANS = (0.48145466, 0.4578275, 0.40821073)
    CLIP_RGB_STDS = (0.26862954, 0.26130258, 0.27577711)

    def __init__(self, input_uuids: List[str], output_uuid: str, observation_space):
        super().__init__(input_uuids=input_uuids, output_uuid=output_uuid, observation_space=observation_space)

    def to(self, device: torch.device):
        self.device = torch.device(device)
        self._model = None   # rebuilt lazily on the new device
        self._twin = None
        self._streams = self._copy_stream = None      # (streams belong to the old device)
        return self


class ClipResNetPreprocessor(_PreprocessorBase):
    """[U] ``ClipResNetPreprocessor(rgb_input_uuid, clip_model_type, pool, device, device_ids, output_uuid)``.

    ``process(obs)``: ``obs[rgb]`` fp32 NHWC [N,224,224,3] (already normalised
    with CLIP_RGB_MEANS/STDS by the sensor) -> fp32 NCHW [N,2048,7,7]
    (``pool=False``) or [N,2048] (``pool=True``), on ``device``.
    """

    def __init__(self, rgb_input_uuid: str, clip_model_type: str, pool: bool,
                 device: Optional[torch.device] = None, device_ids: Optional[List[torch.device]] = None,
                 output_uuid: str = "rgb_clip_resnet", state_dict=None, weights_path: Optional[str] = None,
                 chunk: int = 0, **kwargs: Any):
        assert clip_model_type in ("RN50", "RN50x16")
        if clip_model_type == "RN50":
            output_shape = (2048, 7, 7)
        else:   # width 96, layers (6, 8, 18, 8); the plugin feeds it the same 224x224 frames -> 7x7 map
            output_shape = (3072, 7, 7)
        if pool:
            output_shape = output_shape[:1]
        self.clip_model_type = clip_model_type
        self.pool = pool
        self.device = torch.device("cuda") if device is None else torch.device(device)
        self.device_ids = device_ids or []
        self._state_dict = state_dict
        self._weights_path = weights_path
        self._chunk = chunk
        self._model = None
        low, high = -np.inf, np.inf
        super().__init__([rgb_input_uuid], output_uuid,
                         spaces.Box(low=low, high=high, shape=output_shape, dtype=np.float32))

    @property
    def resnet(self):
        if self._model is None:   # lazy, like the reference plugin
            from .encoder import RN50Trunk
            sd = _load_visual_state_dict(self.clip_model_type, self._state_dict, self._weights_path)
            self._model = RN50Trunk(sd, device=self.device, chunk=self._chunk)
        return self._model

    def _process_host_pipelined(self, x: torch.Tensor) -> torch.Tensor:
        """HOST frames (what the simulators hand over), >= 64 of them, full feature maps: the batch is cut in two halves
        that run copy -> trunk -> NCHW conversion on two HIP streams, so one half's PCIe copy hides behind the other
        half's encoder and the two encoder launches overlap as they do in ``engine.Worker`` (same per-frame results:
        a frame's features do not depend on how the batch is sliced).

        ASYNCHRONOUS with respect to the caller's stream (round 4): nothing here depends on work the caller has in
        flight (the input is host memory, the staging buffers, the two encoder handles and their workspaces are this
        path's own -- never the primary handle the caller's stream uses -- and the output is allocated on a side stream), so the copies and the encoder launches start at once -- beside the policy's act step of the previous
        env step, which is still running on the caller's stream -- and the caller's stream merely WAITS (an event, no
        host sync) for the two halves before whatever consumes the returned tensor.  The host blocks only until the
        H2D copies have left ``x`` (the caller may refill its frame buffer as soon as this returns)."""
        from .encoder import RN50Trunk
        trunk = self.resnet
        if getattr(self, "_twin", None) is None:
            # TWO dedicated handles (borrowing the primary trunk's packed weights), one per side stream: a handle's cached
            # workspace is then only ever touched by launches on its own stream, in that stream's order.  The primary handle
            # (and its workspace) stays with the caller's stream -- process() for < 64 frames / device input / pool=True and
            # process_bf16_nhwc() -- so a direct call followed by a pipelined one (or mixed use of one preprocessor) can never
            # put two launches on one workspace.
            self._twin = [RN50Trunk(None, device=self.device, chunk=self._chunk, weights_from=trunk) for _ in range(2)]
            for t in self._twin:
                t.set_conv8_min_tiles(50)         # two launches in flight: the lower 8-wave dispatch threshold (ec_rn50_set_conv8_min_tiles)
        if getattr(self, "_streams", None) is None:
            # two compute streams + the copy stream, verified to run concurrently (streams that land on one hardware queue
            # would serialise the two halves and their copies: _lib.concurrent_streams)
            # The check is a timing heuristic; in a trainer process that shares its GPU with other AllenAct workers it can fail
            # without anything being wrong.  The drop-in classes then run on plain streams (possibly serialised, i.e. slower)
            # with a warning instead of aborting training; EC_PLUGIN_VERIFY_STREAMS=0 skips the check.  (The benchmark engine,
            # whose numbers rest on two launches really being in flight, keeps the hard error: engine.Worker.)
            st = None
            if os.environ.get("EC_PLUGIN_VERIFY_STREAMS", "1") != "0":
                try:
                    st = _lib.concurrent_streams(3, self.device)
                except RuntimeError as e:
                    import warnings
                    warnings.warn(f"ClipResNetPreprocessor: {e}; falling back to unverified HIP streams (the two halves of a batch may serialise)")
            if st is None:
                st = [torch.cuda.Stream(device=self.device) for _ in range(3)]
            self._streams, self._copy_stream = st[:2], st[2]
            self._stage, self._stage_free, self._calls = {}, {}, 0
        N = x.shape[0]
        nck = max(2, int(os.environ.get("EC_PLUGIN_CHUNKS", "2")))   # pieces of the batch (alternating over the two streams)
        cuts = [round(i * N / nck) for i in range(nck + 1)]
        spans = [(cuts[i], cuts[i + 1]) for i in range(nck) if cuts[i + 1] > cuts[i]]
        cur = torch.cuda.current_stream(self.device)
        par = self._calls & 1                                   # staging buffers are double-buffered over calls
        self._calls += 1
        halves, copied = [], []
        sync_caller = os.environ.get("EC_PLUGIN_ASYNC", "1") == "0"      # (A/B switch: 0 = start behind the caller's stream, as round 3 did)
        if sync_caller:
            self._copy_stream.wait_stream(cur)
            for st in self._streams:
                st.wait_stream(cur)
        with torch.cuda.stream(self._copy_stream):
            # both copies go back to back on ONE copy stream (the SDMA queue); each compute stream waits for its half only
            for k, (a, b) in enumerate(spans):
                key = (par, k, b - a, tuple(x.shape[1:]), x.dtype)
                xs = self._stage.get(key)
                if xs is None:                                   # the preprocessor's OWN device staging (no allocator hand-over
                    xs = self._stage[key] = torch.empty((b - a,) + tuple(x.shape[1:]), dtype=x.dtype, device=self.device)
                free = self._stage_free.get(key)                 # ... between streams); reused once its last reader is done
                if free is not None:
                    self._copy_stream.wait_event(free)
                xs.copy_(x[a:b], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self._copy_stream)
                halves.append((key, xs)); copied.append(ev)
        with torch.cuda.stream(self._streams[0]):               # allocated on a side stream: no dependency on the caller's stream
            out = torch.empty((N, trunk.out_channels, trunk.out_spatial, trunk.out_spatial), dtype=torch.float32, device=self.device)
            alloc = torch.cuda.Event()
            alloc.record(self._streams[0])
        self._streams[1].wait_event(alloc)
        out.record_stream(self._streams[1])
        out.record_stream(cur)
        for k, ((a, b), (key, xs), ev) in enumerate(zip(spans, halves, copied)):
            tr, st = self._twin[k & 1], self._streams[k & 1]
            st.wait_event(ev)
            with torch.cuda.stream(st):
                if xs.dtype == torch.uint8:
                    f = tr.forward_u8(xs, mean=self.CLIP_RGB_MEANS, std=self.CLIP_RGB_STDS)
                else:
                    f = tr.forward(xs if xs.dtype == torch.float32 else xs.to(torch.float32))
                tr.to_nchw_f32(f, out[a:b])
                done = torch.cuda.Event()
                done.record(st)
            self._stage_free[key] = done
        for st in self._streams:
            cur.wait_stream(st)                                  # an event wait on the caller's stream; the host does not block
        # the copies were asynchronous (pinned source): the caller may refill `x` for the next env step as soon as this
        # returns, so wait for the LAST copy here (the encoder launches keep running behind it)
        copied[-1].synchronize()
        return out

    def process(self, obs: Dict[str, Any], *args: Any, **kwargs: Any) -> torch.Tensor:
        x = obs[self.input_uuids[0]]
        if x.shape[-1] == 1:      # depth input: repeated to 3 channels, as upstream does
            x = x.expand(*x.shape[:-1], 3)
        if (not self.pool and isinstance(x, torch.Tensor) and not x.is_cuda and x.dim() == 4 and x.shape[0] >= 64
                and x.shape[-1] == 3 and self.device.type == "cuda"):
            return self._process_host_pipelined(x)
        trunk = self.resnet
        if x.dtype == torch.uint8:   # raw frames: /255 and CLIP mean/std are fused into the stem kernel
            feat = trunk.forward_u8(x.to(self.device).contiguous(), mean=self.CLIP_RGB_MEANS, std=self.CLIP_RGB_STDS)
        else:
            feat = trunk.forward(x.to(self.device, dtype=torch.float32).contiguous())
        return trunk.spatial_mean(feat) if self.pool else trunk.to_nchw_f32(feat)

    def process_bf16_nhwc(self, rgb: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """MI355X-native fast path: device fp32 NHWC frame -> bf16 NHWC [N,7,7,2048] written
        straight into ``out`` (a slice of the rollout feature buffer)."""
        return self.resnet.forward(rgb, out)


class ClipViTPreprocessor(_PreprocessorBase):
    """[U] ``ClipViTPreprocessor(rgb_input_uuid, clip_model_type, class_emb_only, device, device_ids, output_uuid)``
    (allenact master; named by BASELINE.json).  ``process(obs)``: fp32 NHWC [N,224,224,3] ->
    fp32 [N,50,768] tokens after ``resblocks[:-1]`` (``class_emb_only`` -> [N,768])."""

    def __init__(self, rgb_input_uuid: str, clip_model_type: str, class_emb_only: bool = False,
                 device: Optional[torch.device] = None, device_ids: Optional[List[torch.device]] = None,
                 output_uuid: str = "rgb_clip_vit", state_dict=None, weights_path: Optional[str] = None, **kwargs: Any):
        assert clip_model_type in ("ViT-B/32", "ViT-B/16", "ViT-L/14")
        # (tokens, width, heads): ViT-B/32 runs the 64-token MFMA attention core (benchmarked); B/16 and L/14 run the
        # general LDS attention core -- functional, not tuned
        tokens, width, self._heads = {"ViT-B/32": (50, 768, 12), "ViT-B/16": (197, 768, 12),
                                      "ViT-L/14": (257, 1024, 16)}[clip_model_type]
        output_shape = (width,) if class_emb_only else (tokens, width)
        self.clip_model_type = clip_model_type
        self.class_emb_only = class_emb_only
        self.device = torch.device("cuda") if device is None else torch.device(device)
        self.device_ids = device_ids or []
        self._state_dict, self._weights_path = state_dict, weights_path
        self._model = None
        super().__init__([rgb_input_uuid], output_uuid,
                         spaces.Box(low=-np.inf, high=np.inf, shape=output_shape, dtype=np.float32))

    @property
    def vit(self):
        if self._model is None:
            from .encoder import ViTEmbedder
            sd = _load_visual_state_dict(self.clip_model_type, self._state_dict, self._weights_path)
            self._model = ViTEmbedder(sd, device=self.device, heads=self._heads)
        return self._model

    def process(self, obs: Dict[str, Any], *args: Any, **kwargs: Any) -> torch.Tensor:
        x = obs[self.input_uuids[0]].to(self.device, dtype=torch.float32).contiguous()
        m = self.vit
        return m.to_f32(m.forward(x), self.class_emb_only)
