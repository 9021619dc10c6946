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

    w1 = add("conv1", "bn1", keep_f32=True)          # [Cout, 3, 3, 3]
    stem_w = w1.reshape(w1.shape[0], 27).t().contiguous()   # [(ky,kx,ci), Cout]
    add("conv2", "bn2")
    add("conv3", "bn3")
    for li, n in enumerate(layers, start=1):
        for b in range(n):
            p = f"layer{li}.{b}"
            add(p + ".conv1", p + ".bn1")
            add(p + ".conv2", p + ".bn2")
            add(p + ".conv3", p + ".bn3")
            if (p + ".downsample.0.weight") in sd:
                add(p + ".downsample.0", p + ".downsample.1")
    return (width, layers), stem_w, torch.cat(ws), torch.cat(bs)


class RN50Trunk:
    """Frozen CLIP ModifiedResNet trunk on one MI355X.

    ``forward(rgb_nhwc_f32) -> feat bf16 [B, S, S, C]`` (NHWC).  The handle
    borrows the packed device weights held by this object.
    """

    def __init__(self, state_dict: Dict[str, torch.Tensor], device="cuda", input_resolution: int = 224,
                 chunk: int = 0):
        self.lib = _lib.load()
        self.device = torch.device(device)
        (width, layers), stem_w, w, bias = pack_rn50(state_dict)
        self.stem_w = stem_w.to(self.device)
        self.w = w.to(self.device)
        self.bias = bias.to(self.device)
        self.input_resolution = input_resolution
        self.chunk = chunk
        h = C.c_void_p()
        arr = (C.c_int * 4)(*layers)
        _lib.check(self.lib.ec_rn50_create(C.byref(h), width, arr, input_resolution, self.stem_w.data_ptr(),
                                           self.w.data_ptr(), self.w.numel(), self.bias.data_ptr(),
                                           self.bias.numel()), "ec_rn50_create")
        self.h = h
        self.out_channels = self.lib.ec_rn50_out_channels(h)
        self.out_spatial = self.lib.ec_rn50_out_spatial(h)
        self._ws: Optional[torch.Tensor] = None

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.ec_rn50_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _workspace(self, n: int) -> torch.Tensor:
        need = self.lib.ec_rn50_workspace_bytes(self.h, n)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

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

    def to_nchw_f32(self, feat: torch.Tensor) -> torch.Tensor:
        B = feat.shape[0]
        S, Cc = self.out_spatial, self.out_channels
        o = torch.empty((B, Cc, S, S), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.ec_nhwc_bf16_to_nchw_f32(feat.data_ptr(), o.data_ptr(), B, S * S, Cc, _lib.stream_ptr()),
                   "ec_nhwc_bf16_to_nchw_f32")
        return o

    def spatial_mean(self, feat: torch.Tensor) -> torch.Tensor:
        B = feat.shape[0]
        S, Cc = self.out_spatial, self.out_channels
        o = torch.empty((B, Cc), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.ec_spatial_mean_bf16(feat.data_ptr(), o.data_ptr(), B, S * S, Cc, _lib.stream_ptr()),
                   "ec_spatial_mean_bf16")
        return o


# ---- thin op-level wrappers (used by the parity tests) -------------------------------------

def conv_bf16(x, w, bias, res=None, ksize=1, pool=False, act=1):
    """x bf16 [B,H,W,Cin]; w bf16 [Cout, k*k*Cin]; bias f32 [Cout] -> bf16 [B,H',W',Cout]."""
    lib = _lib.load()
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    out = torch.empty((B, Ho, Wo, Cout), dtype=torch.bfloat16, device=x.device)
    _lib.check(lib.ec_conv_bf16(x.data_ptr(), w.data_ptr(), _lib.ptr(bias), _lib.ptr(res), out.data_ptr(), B, H, W,
                                Cin, Cout, ksize, int(pool), act, _lib.stream_ptr()), "ec_conv_bf16")
    return out


def gemm_bf16(a, w, bias=None, res=None, act=0):
    """a bf16 [M,K]; w bf16 [N,K] -> bf16 [M,N] = act(a @ w.T + bias (+res))."""
    lib = _lib.load()
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    _lib.check(lib.ec_gemm_bf16(a.data_ptr(), w.data_ptr(), _lib.ptr(bias), _lib.ptr(res), out.data_ptr(), M, N, K,
                                act, _lib.stream_ptr()), "ec_gemm_bf16")
    return out
