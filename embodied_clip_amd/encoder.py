"""Host side of the frozen CLIP-ResNet encoder: BN folding, weight packing,
and the handle around ``ec_rn50_*`` (include/ec_amd.h).

Mirrors what the reference does around ``clip_model.visual``:
``freeze_model`` (primitive_probing/generate_data/thor_image_features.py:26-33)
makes every BatchNorm an eval-mode affine, which is folded into the preceding
conv here, once, on the host at load time.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib

BN_EPS = 1e-5


def _fold(w: torch.Tensor, sd: Dict[str, torch.Tensor], bn: str) -> Tuple[torch.Tensor, torch.Tensor]:
    g, b = sd[bn + ".weight"].float(), sd[bn + ".bias"].float()
    mu, var = sd[bn + ".running_mean"].float(), sd[bn + ".running_var"].float()
    s = g / torch.sqrt(var + BN_EPS)
    return w.float() * s.view(-1, 1, 1, 1), b - mu * s


def _layers(sd) -> List[int]:
    out = []
    for li in range(1, 5):
        n = 0
        while f"layer{li}.{n}.conv1.weight" in sd:
            n += 1
        out.append(n)
    return out


def pack_rn50(sd: Dict[str, torch.Tensor]):
    """OpenAI-CLIP ``visual.state_dict()`` -> (cfg, stem_w f32 [27,Cout], w bf16 flat, bias f32 flat)
    in the execution order ``ec_rn50_create`` documents."""
    sd = {k[len("visual."):] if k.startswith("visual.") else k: v.detach().cpu() for k, v in sd.items()}
    width = sd["conv3.weight"].shape[0]
    layers = _layers(sd)
    ws: List[torch.Tensor] = []
    bs: List[torch.Tensor] = []

    def add(conv, bn, keep_f32=False):
        w, b = _fold(sd[conv + ".weight"], sd, bn)
        bs.append(b)
        w = w.permute(0, 2, 3, 1).contiguous()      # [Cout, kh, kw, Cin]
        if keep_f32:
            return w
        ws.append(w.reshape(-1).to(torch.bfloat16))
        return None

    # Stem channels are padded to the 32-channel granule of the conv kernels (RN50x16: 48 -> 64) with zero weights
    # and biases: a padded channel is exactly relu(0) = 0 and multiplies zero weights downstream.
    sc_real = sd["conv1.weight"].shape[0]
    sc = (sc_real + 31) // 32 * 32

    def pad_out(w, b):      # [Cout, ...] -> [sc, ...]
        if w.shape[0] == sc:
            return w, b
        return (torch.cat([w, w.new_zeros((sc - w.shape[0],) + tuple(w.shape[1:]))]), torch.cat([b, b.new_zeros(sc - b.shape[0])]))

    def pad_in(w):          # [Cout, kh, kw, Cin] -> [Cout, kh, kw, sc]
        if w.shape[3] == sc:
            return w
        return torch.cat([w, w.new_zeros(tuple(w.shape[:3]) + (sc - w.shape[3],))], dim=3)

    w1, b1 = _fold(sd["conv1.weight"], sd, "bn1")
    w1, b1 = pad_out(w1.permute(0, 2, 3, 1).contiguous(), b1)          # [sc, 3, 3, 3]
    bs.append(b1)
    stem_w = w1.reshape(sc, 27).t().contiguous()                        # [(ky,kx,ci), sc]
    w2, b2 = _fold(sd["conv2.weight"], sd, "bn2")
    w2, b2 = pad_out(pad_in(w2.permute(0, 2, 3, 1).contiguous()), b2)  # [sc, 3, 3, sc]
    bs.append(b2)
    ws.append(w2.reshape(-1).to(torch.bfloat16))
    w3, b3 = _fold(sd["conv3.weight"], sd, "bn3")
    bs.append(b3)
    ws.append(pad_in(w3.permute(0, 2, 3, 1).contiguous()).reshape(-1).to(torch.bfloat16))
    for li, n in enumerate(layers, start=1):
        for b in range(n):
            p = f"layer{li}.{b}"
            add(p + ".conv1", p + ".bn1")
            add(p + ".conv2", p + ".bn2")
            add(p + ".conv3", p + ".bn3")
            if (p + ".downsample.0.weight") in sd:
                add(p + ".downsample.0", p + ".downsample.1")
    return (width, layers), stem_w, torch.cat(ws), torch.cat(bs)


class ClipResizeCrop:
    """``Resize(n_px, BICUBIC)`` + ``CenterCrop(n_px)`` of CLIP's ``_transform`` on raw uint8 frames, bit-exact with
    Pillow (``clip_preprocess``: primitive_probing/generate_data/thor_image_features.py:108).  The coefficient tables
    are built on the host once per frame geometry (``ec_clip_resize_table``) and cached on the device."""

    def __init__(self, device="cuda", n_px: int = 224):
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.n_px = n_px
        self._tables: Dict[Tuple[int, int], Tuple[torch.Tensor, int]] = {}

    def table(self, H: int, W: int) -> Tuple[torch.Tensor, int]:
        """(host int32 table, LDS rows) for H x W frames -- pure host arithmetic, no GPU call."""
        n = self.lib.ec_clip_resize_table_ints(H, W, self.n_px)
        if n == 0:
            raise _lib.EcError(f"unsupported frame geometry {H}x{W} -> {self.n_px}")
        t = torch.empty(n, dtype=torch.int32)
        _lib.check(self.lib.ec_clip_resize_table(H, W, self.n_px, t.data_ptr(), n), "ec_clip_resize_table")
        return t, int(t[6])

    @_lib.on_device
    def __call__(self, frames_u8: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """frames_u8: device uint8 [B, H, W, 3] -> uint8 [B, n_px, n_px, 3]."""
        assert frames_u8.is_cuda and frames_u8.dtype == torch.uint8 and frames_u8.is_contiguous() and frames_u8.shape[3] == 3
        B, H, W, _ = frames_u8.shape
        if (H, W) not in self._tables:
            t, rows = self.table(H, W)
            self._tables[(H, W)] = (t.to(self.device), rows)
        tab, rows = self._tables[(H, W)]
        if out is None:
            out = torch.empty((B, self.n_px, self.n_px, 3), dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.ec_clip_resize_crop_u8(frames_u8.data_ptr(), tab.data_ptr(), rows, out.data_ptr(), B, H, W,
                                                   self.n_px, _lib.stream_ptr()), "ec_clip_resize_crop_u8")
        return out


class RN50Trunk:
    """Frozen CLIP ModifiedResNet trunk on one MI355X.

    ``forward(rgb_nhwc_f32) -> feat bf16 [B, S, S, C]`` (NHWC).  The handle
    borrows the packed device weights held by this object.
    """

    def __init__(self, state_dict: Dict[str, torch.Tensor], device="cuda", input_resolution: int = 224,
                 chunk: int = 0, weights_from: Optional["RN50Trunk"] = None):
        """``weights_from``: another trunk on the same device whose packed weight tensors this handle borrows (the
        engine's per-slice handles then read ONE copy of the 76 MB of weights: one L2 / Infinity-Cache footprint for
        the two concurrent launches instead of two)."""
        self.lib = _lib.load()
        self.device = torch.device(device)
        if weights_from is not None:
            assert weights_from.device == self.device and weights_from.input_resolution == input_resolution
            width, layers = weights_from._arch
            self.stem_w, self.w, self.bias = weights_from.stem_w, weights_from.w, weights_from.bias
        else:
            (width, layers), stem_w, w, bias = self._pack(state_dict)
            self.stem_w = stem_w.to(self.device)
            self.w = w.to(self.device)
            self.bias = bias.to(self.device)
        self._arch = (width, layers)
        self.input_resolution = input_resolution
        self.chunk = chunk
        self.h = self._create(width, layers, input_resolution)
        self.out_channels = self.lib.ec_rn50_out_channels(self.h)
        self.out_spatial = self.lib.ec_rn50_out_spatial(self.h)
        self._ws: Optional[torch.Tensor] = None
        self._resize: Optional[ClipResizeCrop] = None

    _pack = staticmethod(pack_rn50)

    def _create(self, width, layers, input_resolution):
        h = C.c_void_p()
        arr = (C.c_int * 4)(*layers)
        _lib.check(self.lib.ec_rn50_create(C.byref(h), width, arr, input_resolution, self.stem_w.data_ptr(),
                                           self.w.data_ptr(), self.w.numel(), self.bias.data_ptr(),
                                           self.bias.numel()), "ec_rn50_create")
        return h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.ec_rn50_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def plan_hash(self) -> str:
        """Hex hash of the launch plan + library version (keys the PMC summaries under profiles/)."""
        return f"{self.lib.ec_rn50_plan_hash(self.h):016x}"

    def set_conv8_min_tiles(self, n: int) -> None:
        """Dispatch threshold of THIS handle's conv launches (0 = default; see ``ec_rn50_set_conv8_min_tiles``)."""
        _lib.check(self.lib.ec_rn50_set_conv8_min_tiles(self.h, int(n)), "ec_rn50_set_conv8_min_tiles")

    def _workspace(self, n: int) -> torch.Tensor:
        need = self.lib.ec_rn50_workspace_bytes(self.h, n)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    @_lib.on_device
    def forward(self, rgb: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """rgb: device fp32 [B, R, R, 3] contiguous (the RGB sensor's normalised frame).
        out: optional bf16 [B, S, S, C] destination (e.g. a slice of the rollout buffer)."""
        assert rgb.is_cuda and rgb.dtype == torch.float32 and rgb.is_contiguous()
        B, R, R2, c3 = rgb.shape
        assert R == self.input_resolution and R2 == R and c3 == 3, rgb.shape
        S, Cc = self.out_spatial, self.out_channels
        if out is None:
            out = torch.empty((B, S, S, Cc), dtype=torch.bfloat16, device=self.device)
        assert out.is_contiguous() and out.dtype == torch.bfloat16 and out.numel() == B * S * S * Cc
        chunk = self.chunk if self.chunk > 0 else B
        ws = self._workspace(min(chunk, B))
        _lib.check(self.lib.ec_rn50_forward(self.h, rgb.data_ptr(), B, ws.data_ptr(), ws.numel(), out.data_ptr(),
                                            chunk, _lib.stream_ptr()), "ec_rn50_forward")
        return out

    @_lib.on_device
    def forward_u8(self, rgb_u8: torch.Tensor, out: Optional[torch.Tensor] = None,
                   mean=(0.48145466, 0.4578275, 0.40821073), std=(0.26862954, 0.26130258, 0.27577711)) -> torch.Tensor:
        """rgb_u8: device uint8 [B, H, W, 3] raw frames.  Frames that are not already R x R first go through CLIP's
        Resize(R, BICUBIC) + CenterCrop(R) (bit-exact with Pillow); /255 and the CLIP mean/std are fused into the stem."""
        assert rgb_u8.is_cuda and rgb_u8.dtype == torch.uint8 and rgb_u8.is_contiguous() and rgb_u8.shape[3] == 3
        if rgb_u8.shape[1] != self.input_resolution or rgb_u8.shape[2] != self.input_resolution:
            if self._resize is None:
                self._resize = ClipResizeCrop(self.device, self.input_resolution)
            rgb_u8 = self._resize(rgb_u8)
        B, R = rgb_u8.shape[0], rgb_u8.shape[1]
        S, Cc = self.out_spatial, self.out_channels
        if out is None:
            out = torch.empty((B, S, S, Cc), dtype=torch.bfloat16, device=self.device)
        chunk = self.chunk if self.chunk > 0 else B
        ws = self._workspace(min(chunk, B))
        m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
        _lib.check(self.lib.ec_rn50_forward_u8(self.h, rgb_u8.data_ptr(), m3, s3, B, ws.data_ptr(), ws.numel(),
                                               out.data_ptr(), chunk, _lib.stream_ptr()), "ec_rn50_forward_u8")
        return out

    @_lib.on_device
    def to_nchw_f32(self, feat: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        B = feat.shape[0]
        S, Cc = self.out_spatial, self.out_channels
        o = out if out is not None else torch.empty((B, Cc, S, S), dtype=torch.float32, device=self.device)
        assert o.is_contiguous() and o.dtype == torch.float32 and o.numel() == B * Cc * S * S
        _lib.check(self.lib.ec_nhwc_bf16_to_nchw_f32(feat.data_ptr(), o.data_ptr(), B, S * S, Cc, _lib.stream_ptr()),
                   "ec_nhwc_bf16_to_nchw_f32")
        if out is not None and not torch.is_inference(o):
            # the kernel wrote through the raw pointer: bump the tensor's version counter (a zero-element in-place op, no
            # launch), so that caches keyed on it (policy._bf16_rows) see the storage as changed
            o.view(-1)[:0].zero_()
        return o

    @_lib.on_device
    def spatial_mean(self, feat: torch.Tensor) -> torch.Tensor:
        B = feat.shape[0]
        S, Cc = self.out_spatial, self.out_channels
        o = torch.empty((B, Cc), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.ec_spatial_mean_bf16(feat.data_ptr(), o.data_ptr(), B, S * S, Cc, _lib.stream_ptr()),
                   "ec_spatial_mean_bf16")
        return o


IMAGENET_RGB_MEANS = (0.485, 0.456, 0.406)      # thor_image_features.py:41
IMAGENET_RGB_STDS = (0.229, 0.224, 0.225)       # thor_image_features.py:42
TV_STEM_KROW, TV_STEM_K = 24, 176               # ec_stem7_pool's K layout


def pack_tv_resnet(sd: Dict[str, torch.Tensor]):
    """``torchvision.models.resnet50().state_dict()`` (``fc.*`` ignored: the reference drops avgpool and fc,
    thor_image_features.py:47) -> (cfg, stem_w bf16 [64,176], w bf16 flat, bias f32 flat) as ``ec_rn50tv_create`` documents."""
    sd = {k: v.detach().cpu() for k, v in sd.items()}
    width = sd["conv1.weight"].shape[0]
    assert tuple(sd["conv1.weight"].shape[1:]) == (3, 7, 7), "not a torchvision ResNet state dict"
    layers = _layers(sd)
    w1, b1 = _fold(sd["conv1.weight"], sd, "bn1")                       # [w, 3, 7, 7]
    rows = w1.permute(0, 2, 3, 1).reshape(width, 7, 21)                 # [(ky), (kx, ci)]
    stem = torch.zeros(width, TV_STEM_K)
    stem[:, :7 * TV_STEM_KROW].view(width, 7, TV_STEM_KROW)[:, :, :21] = rows
    ws: List[torch.Tensor] = []
    bs: List[torch.Tensor] = [b1]
    for li, n in enumerate(layers, start=1):
        for b in range(n):
            p = f"layer{li}.{b}"
            for conv, bn in ((".conv1", ".bn1"), (".conv2", ".bn2"), (".conv3", ".bn3"), (".downsample.0", ".downsample.1")):
                if (p + conv + ".weight") not in sd:
                    continue
                w, bb = _fold(sd[p + conv + ".weight"], sd, p + bn)
                bs.append(bb)
                ws.append(w.permute(0, 2, 3, 1).reshape(-1).to(torch.bfloat16))
    return (width, layers), stem.to(torch.bfloat16).contiguous(), torch.cat(ws), torch.cat(bs)


class ImageNetRN50Trunk(RN50Trunk):
    """Frozen torchvision ResNet-50 without avgpool / fc: the ``resnet_model`` of the feature scripts
    (primitive_probing/generate_data/thor_image_features.py:46-49, reachable_image_features.py:48-51).
    ``forward(rgb_nhwc_f32 ImageNet-normalised) -> bf16 [B,7,7,2048]`` == ``imagenet_conv`` (NHWC, before ``.float()``);
    ``spatial_mean`` of it == ``imagenet_avgpool`` (:51-54,105-106).  Same executor and handle type as the CLIP trunk."""

    _pack = staticmethod(pack_tv_resnet)

    def _create(self, width, layers, input_resolution):
        if width != 64:
            raise _lib.EcError("ec_rn50tv_create: the 7x7 stem kernel is built for 64 channels (torchvision resnet50/101/152)")
        h = C.c_void_p()
        arr = (C.c_int * 4)(*layers)
        _lib.check(self.lib.ec_rn50tv_create(C.byref(h), arr, input_resolution, self.stem_w.data_ptr(), self.w.data_ptr(),
                                             self.w.numel(), self.bias.data_ptr(), self.bias.numel()), "ec_rn50tv_create")
        return h

    def forward_u8(self, rgb_u8, out=None, mean=IMAGENET_RGB_MEANS, std=IMAGENET_RGB_STDS):
        """raw uint8 frames: Resize(224, BICUBIC) + CenterCrop (``resnet_preprocess``, thor_image_features.py:36-39, the
        same Pillow-exact kernel as the CLIP branch) when needed, ToTensor + Normalize(ImageNet) fused into the stem."""
        return super().forward_u8(rgb_u8, out=out, mean=mean, std=std)


class AttentionPool:
    """[U] CLIP ``AttentionPool2d`` on bf16 NHWC trunk features (the ``clip_pool`` the reference detaches at
    primitive_probing/generate_data/thor_image_features.py:62 and calls at :112)."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], device="cuda", num_heads: int = 32, prefix="attnpool.",
                 weights_from: Optional["AttentionPool"] = None):
        self.lib = _lib.load()
        self.device = torch.device(device)
        if weights_from is not None:      # borrow the other instance's device tensors (own workspace): see RN50Trunk
            o = weights_from
            assert o.device == self.device
            self.pos, self.wq, self.bq, self.wkv, self.bkv, self.wc, self.bc = o.pos, o.wq, o.bq, o.wkv, o.bkv, o.wc, o.bc
            self.C, self.out_dim, self.heads, self._ws = o.C, o.out_dim, o.heads, None
            return
        sd = {k[len("visual."):] if k.startswith("visual.") else k: v for k, v in state_dict.items()}
        g = lambda n: sd[prefix + n].detach().float()
        bf = lambda t: t.to(torch.bfloat16).contiguous().to(self.device)
        f32 = lambda t: t.float().contiguous().to(self.device)
        self.pos = f32(g("positional_embedding"))
        self.wq, self.bq = bf(g("q_proj.weight")), f32(g("q_proj.bias"))
        self.wkv = bf(torch.cat([g("k_proj.weight"), g("v_proj.weight")], 0))
        self.bkv = f32(torch.cat([g("k_proj.bias"), g("v_proj.bias")], 0))
        self.wc, self.bc = bf(g("c_proj.weight")), f32(g("c_proj.bias"))
        self.C = self.wq.shape[0]
        self.out_dim = self.wc.shape[0]
        self.heads = num_heads
        self._ws = None

    @_lib.on_device
    def forward(self, feat: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """feat bf16 [B,S,S,C] -> fp32 [B,out_dim] (written into ``out`` when given, e.g. a rollout-buffer slice)."""
        B = feat.shape[0]
        HW = feat.numel() // (B * self.C)
        need = self.lib.ec_attnpool_workspace_bytes(B, HW, self.C)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        if out is None:
            out = torch.empty((B, self.out_dim), dtype=torch.float32, device=self.device)
        assert out.is_contiguous() and out.dtype == torch.float32 and out.numel() == B * self.out_dim
        _lib.check(self.lib.ec_attnpool_forward(feat.data_ptr(), B, HW, self.C, self.heads, self.out_dim,
                                                self.pos.data_ptr(), self.wq.data_ptr(), self.bq.data_ptr(),
                                                self.wkv.data_ptr(), self.bkv.data_ptr(), self.wc.data_ptr(),
                                                self.bc.data_ptr(), self._ws.data_ptr(), self._ws.numel(),
                                                out.data_ptr(), _lib.stream_ptr()), "ec_attnpool_forward")
        return out


def pack_vit(sd: Dict[str, torch.Tensor], drop_last: int = 1):
    """OpenAI-CLIP ``visual.state_dict()`` of a VisionTransformer -> (cfg, w bf16 flat, params f32 flat) in the
    order ``ec_vit_create`` documents.  ``drop_last=1`` == ClipViTEmbedder (``resblocks[:-1]``)."""
    sd = {k[len("visual."):] if k.startswith("visual.") else k: v.detach().cpu().float() for k, v in sd.items()}
    D = sd["class_embedding"].numel()
    patch = sd["conv1.weight"].shape[-1]
    L = sd["positional_embedding"].shape[0]
    n = 0
    while f"transformer.resblocks.{n}.ln_1.weight" in sd:
        n += 1
    run = n - drop_last
    ws = [sd["conv1.weight"].permute(0, 2, 3, 1).reshape(D, -1)]
    fs = [sd["class_embedding"], sd["positional_embedding"].reshape(-1), sd["ln_pre.weight"], sd["ln_pre.bias"]]
    for i in range(run):
        p = f"transformer.resblocks.{i}."
        ws += [sd[p + "attn.in_proj_weight"], sd[p + "attn.out_proj.weight"], sd[p + "mlp.c_fc.weight"],
               sd[p + "mlp.c_proj.weight"]]
        fs += [sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], sd[p + "attn.in_proj_bias"], sd[p + "attn.out_proj.bias"],
               sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], sd[p + "mlp.c_fc.bias"], sd[p + "mlp.c_proj.bias"]]
    w = torch.cat([t.reshape(-1) for t in ws]).to(torch.bfloat16)
    f = torch.cat([t.reshape(-1) for t in fs]).float()
    return dict(width=D, layers_run=run, patch=patch, tokens=L), w, f


class ViTEmbedder:
    """Frozen CLIP VisionTransformer run as [U] ``ClipViTEmbedder`` does: tokens after all but the last block."""

    def __init__(self, state_dict, device="cuda", heads: int = 12, input_resolution: int = 224, drop_last: int = 1,
                 weights_from: Optional["ViTEmbedder"] = None):
        """``weights_from``: another embedder on the same device whose packed weights this handle borrows (see RN50Trunk)."""
        self.lib = _lib.load()
        self.device = torch.device(device)
        if weights_from is not None:
            assert weights_from.device == self.device
            cfg, self.w, self.f = weights_from._cfg, weights_from.w, weights_from.f
        else:
            cfg, w, f = pack_vit(state_dict, drop_last)
            self.w, self.f = w.to(self.device), f.to(self.device)
        self._cfg = cfg
        self.D, self.input_resolution = cfg["width"], input_resolution
        h = C.c_void_p()
        _lib.check(self.lib.ec_vit_create(C.byref(h), cfg["width"], cfg["layers_run"], heads, cfg["patch"],
                                          input_resolution, self.w.data_ptr(), self.w.numel(), self.f.data_ptr(),
                                          self.f.numel()), "ec_vit_create")
        self.h = h
        self.L = self.lib.ec_vit_tokens(h)
        self._ws = None

    def plan_hash(self) -> str:
        """Hex hash of what fixes the launch plan + library version (keys the PMC summaries under profiles/)."""
        return f"{self.lib.ec_vit_plan_hash(self.h):016x}"

    def set_conv8_min_tiles(self, n: int) -> None:
        """Dispatch threshold of THIS handle's GEMM launches (0 = default; see ``ec_vit_set_conv8_min_tiles``)."""
        _lib.check(self.lib.ec_vit_set_conv8_min_tiles(self.h, int(n)), "ec_vit_set_conv8_min_tiles")

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.ec_vit_destroy(self.h)
                self.h = None
        except Exception:
            pass

    @_lib.on_device
    def forward(self, rgb: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """rgb device fp32 [B,R,R,3] -> tokens bf16 [B,L,D]."""
        assert rgb.is_cuda and rgb.dtype == torch.float32 and rgb.is_contiguous()
        B = rgb.shape[0]
        need = self.lib.ec_vit_workspace_bytes(self.h, B)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        if out is None:
            out = torch.empty((B, self.L, self.D), dtype=torch.bfloat16, device=self.device)
        _lib.check(self.lib.ec_vit_forward(self.h, rgb.data_ptr(), B, self._ws.data_ptr(), self._ws.numel(),
                                           out.data_ptr(), _lib.stream_ptr()), "ec_vit_forward")
        return out

    @_lib.on_device
    def to_f32(self, tokens: torch.Tensor, class_emb_only: bool = False) -> torch.Tensor:
        B = tokens.shape[0]
        if class_emb_only:
            o = torch.empty((B, self.D), dtype=torch.float32, device=self.device)
            _lib.check(self.lib.ec_bf16_to_f32(tokens.data_ptr(), o.data_ptr(), B, self.D, self.L * self.D,
                                               _lib.stream_ptr()))
        else:
            o = torch.empty((B, self.L, self.D), dtype=torch.float32, device=self.device)
            _lib.check(self.lib.ec_bf16_to_f32(tokens.data_ptr(), o.data_ptr(), B, self.L * self.D, self.L * self.D,
                                               _lib.stream_ptr()))
        return o


# ---- thin op-level wrappers (used by the parity tests) -------------------------------------

def _guard_first(fn):
    """Run a free op wrapper on the GPU that owns its first tensor argument."""
    import functools

    @functools.wraps(fn)
    def wrapped(x, *a, **kw):
        with _lib.tensor_guard(x):
            return fn(x, *a, **kw)
    return wrapped


@_guard_first
def conv_bf16(x, w, bias, res=None, ksize=1, pool=False, act=1, out=None):
    """x bf16 [B,H,W,Cin]; w bf16 [Cout, k*k*Cin]; bias f32 [Cout] -> bf16 [B,H',W',Cout].
    ``out`` may be a COLUMN BLOCK ``wide[..., c0:c0 + Cout]`` of a wider contiguous NHWC tensor (``ec_conv_bf16_ld``)."""
    lib = _lib.load()
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    if out is None:
        out = torch.empty((B, Ho, Wo, Cout), dtype=torch.bfloat16, device=x.device)
    ld = _row_stride(out, (B, Ho, Wo, Cout))
    _lib.check(lib.ec_conv_bf16_ld(x.data_ptr(), w.data_ptr(), _lib.ptr(bias), _lib.ptr(res), out.data_ptr(), B, H, W,
                                   Cin, Cout, ksize, int(pool), act, ld, _lib.stream_ptr()), "ec_conv_bf16_ld")
    return out


def _row_stride(out, shape) -> int:
    """Row stride (elements) of ``out`` [B,H,W,C]: C for a contiguous tensor, the parent's channel count for a column block."""
    assert tuple(out.shape) == tuple(shape) and out.dtype == torch.bfloat16, (out.shape, shape, out.dtype)
    B, H, W, C = shape
    ld = out.stride(2)
    assert out.stride(3) == 1 and ld >= C and out.stride(1) == W * ld and (B == 1 or out.stride(0) == H * W * ld), out.stride()
    return ld


@_guard_first
def avgpool2_bf16(x, out=None):
    """AvgPool2d(2) on bf16 NHWC (``ec_avgpool2_bf16_ld``); ``out`` may be a column block of a wider NHWC tensor."""
    lib = _lib.load()
    B, H, W, C = x.shape
    if out is None:
        out = torch.empty((B, H // 2, W // 2, C), dtype=torch.bfloat16, device=x.device)
    ld = _row_stride(out, (B, H // 2, W // 2, C))
    _lib.check(lib.ec_avgpool2_bf16_ld(x.data_ptr(), out.data_ptr(), B, H, W, C, ld, _lib.stream_ptr()), "ec_avgpool2_bf16_ld")
    return out


@_guard_first
def conv_bf16_s2(x, w, bias, res=None, ksize=1, act=1, out=None):
    """Stride-2 conv (torchvision Bottleneck conv2 / downsample conv, ``ec_conv_bf16_s2``): x bf16 [B,H,W,Cin],
    w bf16 [Cout, k*k*Cin], res bf16 [B,H/2,W/2,Cout] -> bf16 [B,H/2,W/2,Cout]."""
    lib = _lib.load()
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    if out is None:
        out = torch.empty((B, H // 2, W // 2, Cout), dtype=torch.bfloat16, device=x.device)
    _lib.check(lib.ec_conv_bf16_s2(x.data_ptr(), w.data_ptr(), _lib.ptr(bias), _lib.ptr(res), out.data_ptr(), B, H, W, Cin, Cout,
                                   ksize, act, _lib.stream_ptr()), "ec_conv_bf16_s2")
    return out


@_guard_first
def stem7_pool(rgb, stem_w, bias, mean=None, std=None):
    """torchvision stem in one launch (``ec_stem7_pool``): rgb fp32 NHWC (normalised) or uint8 NHWC (then ``mean`` / ``std``
    are fused) -> bf16 [B,H/4,W/4,64]."""
    lib = _lib.load()
    B, H, W, _ = rgb.shape
    u8 = rgb.dtype == torch.uint8
    out = torch.empty((B, H // 4, W // 4, 64), dtype=torch.bfloat16, device=rgb.device)
    m3 = (C.c_float * 3)(*(mean or (0.0, 0.0, 0.0)))
    s3 = (C.c_float * 3)(*(std or (1.0, 1.0, 1.0)))
    _lib.check(lib.ec_stem7_pool(rgb.data_ptr(), int(u8), m3, s3, stem_w.data_ptr(), bias.data_ptr(), out.data_ptr(), B, H, W,
                                 _lib.stream_ptr()), "ec_stem7_pool")
    return out


@_guard_first
def bneck_conv23_bf16(c1, w2, b2, w3, b3, x, out=None):
    """Fused conv2 (3x3) + conv3 (1x1) + identity + ReLU of a stride-1 Bottleneck (``ec_bneck_conv23_bf16``):
    c1 bf16 [B,14,14,C], w2 bf16 [C, 9C], w3 bf16 [4C, C], x bf16 [B,14,14,4C] -> bf16 [B,14,14,4C]."""
    lib = _lib.load()
    B, H, W, C = c1.shape
    if out is None:
        out = torch.empty_like(x)
    packed = bneck_pack_weights(w2, w3)
    _lib.check(lib.ec_bneck_conv23_bf16(c1.data_ptr(), packed.data_ptr(), b2.data_ptr(), b3.data_ptr(), x.data_ptr(),
                                        out.data_ptr(), B, H, W, C, _lib.stream_ptr()), "ec_bneck_conv23_bf16")
    return out


@_guard_first
def conv3x3_img_bf16(x, w, bias, out=None, pool=False):
    """relu(conv3x3(x) + bias) on the image-resident small-launch kernel (``ec_conv3x3_img_bf16``): x bf16 [B,14,14,256] or
    [B,7,7,512], w bf16 [C, 9C]; ``pool``: [B,14,14,512] -> AvgPool2d(2) of the result, [B,7,7,512]."""
    lib = _lib.load()
    B, H, W, C = x.shape
    packed = _packed_lookup("img", (w,))
    if packed is None:
        packed = torch.empty_like(w)
        _lib.check(lib.ec_conv3x3_img_pack(w.data_ptr(), packed.data_ptr(), C, _lib.stream_ptr()), "ec_conv3x3_img_pack")
        _packed_store("img", (w,), packed)
    if out is None:
        out = torch.empty((B, H // 2, W // 2, C), dtype=x.dtype, device=x.device) if pool else torch.empty_like(x)
    ld = _row_stride(out, (B, H // 2, W // 2, C) if pool else (B, H, W, C))      # (a column block only with pool=True)
    _lib.check(lib.ec_conv3x3_img_bf16_ld(x.data_ptr(), packed.data_ptr(), bias.data_ptr(), out.data_ptr(), B, H, W, C, int(pool),
                                          ld, _lib.stream_ptr()), "ec_conv3x3_img_bf16_ld")
    return out


@_guard_first
def bneck_conv123_bf16(x, w1, b1, w2, b2, w3, b3, out=None):
    """The whole stride-1 Bottleneck in one launch (``ec_bneck_conv123_bf16``): x bf16 [B,14,14,4C], w1 bf16 [C,4C], w2 bf16
    [C,9C], w3 bf16 [4C,C] -> bf16 [B,14,14,4C]."""
    lib = _lib.load()
    B, H, W, C4 = x.shape
    C = C4 // 4
    packed = _packed_lookup("bneck3", (w1, w2, w3))
    if packed is None:
        packed = torch.empty(lib.ec_bneck3_packed_elems(C), dtype=torch.bfloat16, device=x.device)
        _lib.check(lib.ec_bneck3_pack_weights(w1.data_ptr(), w2.data_ptr(), w3.data_ptr(), packed.data_ptr(), C, _lib.stream_ptr()),
                   "ec_bneck3_pack_weights")
        _packed_store("bneck3", (w1, w2, w3), packed)
    if out is None:
        out = torch.empty_like(x)
    _lib.check(lib.ec_bneck_conv123_bf16(x.data_ptr(), packed.data_ptr(), b1.data_ptr(), b2.data_ptr(), b3.data_ptr(), out.data_ptr(),
                                         B, H, W, C, _lib.stream_ptr()), "ec_bneck_conv123_bf16")
    return out


_BNECK_PACKED = {}


def _packed_lookup(tag, tensors):
    """Packed-weight cache keyed by the IDENTITY of the source tensors (weak references: a freed tensor's address can be
    handed to another one) and their version counters."""
    import weakref  # noqa: F401
    hit = _BNECK_PACKED.get((tag,) + tuple(id(t) for t in tensors))
    if hit is None:
        return None
    refs, versions, packed = hit
    if all(r() is t for r, t in zip(refs, tensors)) and versions == tuple(t._version for t in tensors):
        return packed
    return None


def _packed_store(tag, tensors, packed):
    import weakref
    if len(_BNECK_PACKED) > 32:
        _BNECK_PACKED.clear()
    _BNECK_PACKED[(tag,) + tuple(id(t) for t in tensors)] = (tuple(weakref.ref(t) for t in tensors),
                                                           tuple(t._version for t in tensors), packed)


def bneck_pack_weights(w2, w3):
    """The two weight matrices in the fused kernel's streaming order (``ec_bneck_pack_weights``); cached per tensor pair."""
    hit = _packed_lookup("bneck", (w2, w3))
    if hit is not None:
        return hit
    lib = _lib.load()
    C = w3.shape[1]
    packed = torch.empty(lib.ec_bneck_packed_elems(C), dtype=torch.bfloat16, device=w2.device)
    _lib.check(lib.ec_bneck_pack_weights(w2.data_ptr(), w3.data_ptr(), packed.data_ptr(), C, _lib.stream_ptr()), "ec_bneck_pack_weights")
    _packed_store("bneck", (w2, w3), packed)
    return packed


@_guard_first
def gemm_bf16(a, w, bias=None, res=None, act=0):
    """a bf16 [M,K]; w bf16 [N,K] -> bf16 [M,N] = act(a @ w.T + bias (+res))."""
    lib = _lib.load()
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    _lib.check(lib.ec_gemm_bf16(a.data_ptr(), w.data_ptr(), _lib.ptr(bias), _lib.ptr(res), out.data_ptr(), M, N, K,
                                act, _lib.stream_ptr()), "ec_gemm_bf16")
    return out


@_guard_first
def conv1x1_pair_bf16(a0, w0, b0, w2, b2, a1=None, w1=None, b1=None, res=None):
    """y = relu(a0 @ w0.T + b0 [+ a1 @ w1.T + b1] [+ res]) bf16 [M,256];  z = relu(y @ w2.T + b2) bf16 [M,N2].
    a0/a1 bf16 [M,64], w0/w1 bf16 [256,64], res bf16 [M,256], w2 bf16 [N2,256] (layer-1 Bottleneck boundary)."""
    lib = _lib.load()
    M, K0 = a0.shape
    N, N2 = w0.shape[0], w2.shape[0]
    y = torch.empty((M, N), dtype=torch.bfloat16, device=a0.device)
    z = torch.empty((M, N2), dtype=torch.bfloat16, device=a0.device)
    _lib.check(lib.ec_conv1x1_pair_bf16(a0.data_ptr(), w0.data_ptr(), b0.data_ptr(), _lib.ptr(a1), _lib.ptr(w1),
                                        _lib.ptr(b1), _lib.ptr(res), y.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                        z.data_ptr(), M, K0, N, N2, _lib.stream_ptr()), "ec_conv1x1_pair_bf16")
    return y, z


def pack_text(sd: Dict[str, torch.Tensor]):
    """OpenAI-CLIP ``state_dict()`` entries of the text tower (``token_embedding.weight``, ``positional_embedding``,
    ``transformer.resblocks.*``, ``ln_final.*``, ``text_projection``) -> (cfg, w bf16 flat, params f32 flat) in the
    order ``ec_text_create`` documents."""
    sd = {k: v.detach().cpu().float() for k, v in sd.items()}
    vocab, D = sd["token_embedding.weight"].shape
    ctx = sd["positional_embedding"].shape[0]
    E = sd["text_projection"].shape[1]
    n = 0
    while f"transformer.resblocks.{n}.ln_1.weight" in sd:
        n += 1
    ws, fs = [], [sd["token_embedding.weight"].reshape(-1), sd["positional_embedding"].reshape(-1)]
    for i in range(n):
        p = f"transformer.resblocks.{i}."
        ws += [sd[p + "attn.in_proj_weight"], sd[p + "attn.out_proj.weight"], sd[p + "mlp.c_fc.weight"],
               sd[p + "mlp.c_proj.weight"]]
        fs += [sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], sd[p + "attn.in_proj_bias"], sd[p + "attn.out_proj.bias"],
               sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], sd[p + "mlp.c_fc.bias"], sd[p + "mlp.c_proj.bias"]]
    ws.append(sd["text_projection"].t().contiguous())
    fs += [sd["ln_final.weight"], sd["ln_final.bias"]]
    w = torch.cat([t.reshape(-1) for t in ws]).to(torch.bfloat16)
    f = torch.cat([t.reshape(-1) for t in fs]).float()
    return dict(width=D, layers=n, context_length=ctx, vocab_size=vocab, embed_dim=E), w, f


class ClipTextEncoder:
    """Frozen CLIP text tower on one MI355X: ``encode_text(tokens int [B, ctx]) -> fp32 [B, embed_dim]``
    (== ``clip_model.encode_text``).  In the zero-shot ObjectNav variant it is run once over the goal strings to
    build the goal-embedding table that replaces ``nn.Embedding`` (readme_files/zeroshot_objectnav.md)."""

    def __init__(self, state_dict, device="cuda", heads: Optional[int] = None):
        self.lib = _lib.load()
        self.device = torch.device(device)
        cfg, w, f = pack_text(state_dict)
        self.cfg = cfg
        self.w, self.f = w.to(self.device), f.to(self.device)
        heads = cfg["width"] // 64 if heads is None else heads        # CLIP: transformer_heads = transformer_width // 64
        h = C.c_void_p()
        _lib.check(self.lib.ec_text_create(C.byref(h), cfg["width"], cfg["layers"], heads, cfg["context_length"],
                                           cfg["vocab_size"], cfg["embed_dim"], self.w.data_ptr(), self.w.numel(),
                                           self.f.data_ptr(), self.f.numel()), "ec_text_create")
        self.h = h
        self._ws = None

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.ec_text_destroy(self.h)
                self.h = None
        except Exception:
            pass

    @_lib.on_device
    @torch.no_grad()
    def encode_text(self, tokens: torch.Tensor) -> torch.Tensor:
        assert tokens.dim() == 2 and tokens.shape[1] == self.cfg["context_length"], tokens.shape
        t = tokens.to(self.device, torch.int32).contiguous()
        B = t.shape[0]
        need = self.lib.ec_text_workspace_bytes(self.h, B)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        out = torch.empty((B, self.cfg["embed_dim"]), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.ec_text_forward(self.h, t.data_ptr(), B, self._ws.data_ptr(), self._ws.numel(),
                                            out.data_ptr(), _lib.stream_ptr()), "ec_text_forward")
        return out

    def goal_table(self, goal_tokens: torch.Tensor, normalize: bool = True) -> torch.Tensor:
        """[num_goals, ctx] token ids -> [num_goals, embed_dim] (L2-normalised like CLIP's zero-shot classifier)."""
        e = self.encode_text(goal_tokens)
        return e / e.norm(dim=-1, keepdim=True) if normalize else e


@_guard_first
def conv1x1_pair_pool_bf16(a0, w0, b0, res, w2, b2):
    """Layer-1 -> layer-2 boundary with the pooled copy: a0 bf16 [B,H,W,64], res bf16 [B,H,W,256] ->
    (y [B,H,W,256], AvgPool2d(2)(y) [B,H/2,W/2,256], z = relu(y @ w2.T + b2) [B,H,W,128])."""
    lib = _lib.load()
    B, H, W, K0 = a0.shape
    N, N2 = w0.shape[0], w2.shape[0]
    y = torch.empty((B, H, W, N), dtype=torch.bfloat16, device=a0.device)
    yp = torch.empty((B, H // 2, W // 2, N), dtype=torch.bfloat16, device=a0.device)
    z = torch.empty((B, H, W, N2), dtype=torch.bfloat16, device=a0.device)
    _lib.check(lib.ec_conv1x1_pair_pool_bf16(a0.data_ptr(), w0.data_ptr(), b0.data_ptr(), res.data_ptr(), y.data_ptr(),
                                             yp.data_ptr(), w2.data_ptr(), b2.data_ptr(), z.data_ptr(), B, H, W, K0, N,
                                             N2, _lib.stream_ptr()), "ec_conv1x1_pair_pool_bf16")
    return y, yp, z
