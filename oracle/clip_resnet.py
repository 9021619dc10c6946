"""Oracle: CLIP ModifiedResNet trunk, AttentionPool2d, avgpool head.

Restates openai/CLIP ``clip/model.py`` (pinned at 40f5484c by
``primitive_probing/environment.yml:22``) as driven by the reference's call
sites:

  * ``primitive_probing/generate_data/thor_image_features.py:26-33``
    ``freeze_model``: BN momentum 0 + eval  => BN is the running-stat affine.
  * ``thor_image_features.py:57-68``: ``clip_model.visual`` with
    ``attnpool`` detached (``:62``) and replaced by Identity (``:67``),
    avgpool head = AdaptiveAvgPool2d(1)+Flatten (``:63-66``).
  * ``thor_image_features.py:108-113``: trunk(frame) -> conv features,
    attnpool(features), avgpool(features.float()).

Functional (state-dict in, tensors out) so the same code serves random and
real weights.  ``emulate_bf16=True`` rounds folded weights and every layer
output to bf16 (fp32 accumulate), mirroring where the HIP path rounds; that
mode exists so GPU parity can be asserted tightly (indexing bugs hide under
a loose bf16-vs-fp32 tolerance).  TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # nn.BatchNorm2d default, used by CLIP's ModifiedResNet


def _r(x: torch.Tensor, emulate: bool) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32) if emulate else x


def fold_bn(w: torch.Tensor, sd: Dict[str, torch.Tensor], bn: str) -> Tuple[torch.Tensor, torch.Tensor]:
    """freeze_model contract (thor_image_features.py:26-33): eval-mode BN is
    y = (x - mu)/sqrt(var+eps)*gamma + beta, foldable into the conv:
    w' = w*gamma/sqrt(var+eps), b' = beta - mu*gamma/sqrt(var+eps)."""
    g = sd[bn + ".weight"].float()
    b = sd[bn + ".bias"].float()
    mu = sd[bn + ".running_mean"].float()
    var = sd[bn + ".running_var"].float()
    s = g / torch.sqrt(var + BN_EPS)
    return w.float() * s.view(-1, 1, 1, 1), b - mu * s


def _conv_bn(x, sd, conv, bn, stride=1, padding=0, emulate=False, fold=True):
    w = sd[conv + ".weight"].float()
    if fold:
        wf, bf = fold_bn(w, sd, bn)
        return F.conv2d(x, _r(wf, emulate), bf, stride=stride, padding=padding)
    y = F.conv2d(x, w, None, stride=stride, padding=padding)
    return F.batch_norm(y, sd[bn + ".running_mean"].float(), sd[bn + ".running_var"].float(),
                        sd[bn + ".weight"].float(), sd[bn + ".bias"].float(), False, 0.0, BN_EPS)


def _layer_cfg(sd: Dict[str, torch.Tensor]) -> List[int]:
    layers = []
    for li in range(1, 5):
        n = 0
        while f"layer{li}.{n}.conv1.weight" in sd:
            n += 1
        layers.append(n)
    return layers


def bottleneck(x, sd, p, stride, emulate=False, fold=True):
    """CLIP ``Bottleneck.forward``: conv1(1x1)-bn-relu, conv2(3x3,p1,s1)-bn-relu,
    AvgPool2d(stride) if stride>1 (anti-aliased stride), conv3(1x1)-bn;
    identity = AvgPool2d(stride)->conv1x1->bn when present; relu(out+identity)."""
    out = _r(F.relu(_conv_bn(x, sd, p + ".conv1", p + ".bn1", emulate=emulate, fold=fold)), emulate)
    out = F.relu(_conv_bn(out, sd, p + ".conv2", p + ".bn2", padding=1, emulate=emulate, fold=fold))
    if stride > 1:
        out = F.avg_pool2d(out, stride)
    out = _r(out, emulate)
    out = _conv_bn(out, sd, p + ".conv3", p + ".bn3", emulate=emulate, fold=fold)
    if (p + ".downsample.0.weight") in sd:
        idt = x
        if stride > 1:
            idt = _r(F.avg_pool2d(idt, stride), emulate)
        idt = _r(_conv_bn(idt, sd, p + ".downsample.0", p + ".downsample.1", emulate=emulate, fold=fold), emulate)
    else:
        idt = x
    return _r(F.relu(out + idt), emulate)


def rn50_trunk(x_nchw: torch.Tensor, sd: Dict[str, torch.Tensor], emulate_bf16: bool = False,
               fold: bool = True, return_stages: bool = False):
    """``ModifiedResNet.forward`` with ``attnpool = Identity``
    (thor_image_features.py:67,109): 3-conv stem (3->w/2 s2, w/2->w/2, w/2->w,
    each conv-bn-relu) + AvgPool2d(2), then layer1..4.
    x: fp32 [B,3,R,R] (CLIP-normalised).  Returns fp32 [B, 32*w, R/32, R/32]."""
    e = emulate_bf16
    stages = {}
    # the HIP stem conv1 runs on the bf16 MFMA like every other conv (round 3): in emulation mode the frame and the
    # folded weights are rounded to bf16, the accumulation stays fp32 and the output is rounded to bf16.
    x = _r(x_nchw.float(), e)
    x = _r(F.relu(_conv_bn(x, sd, "conv1", "bn1", stride=2, padding=1, emulate=e, fold=fold)), e)
    stages["stem1"] = x
    x = _r(F.relu(_conv_bn(x, sd, "conv2", "bn2", padding=1, emulate=e, fold=fold)), e)
    stages["stem2"] = x
    x = F.relu(_conv_bn(x, sd, "conv3", "bn3", padding=1, emulate=e, fold=fold))
    x = _r(F.avg_pool2d(x, 2), e)
    stages["stem"] = x
    for li, nblocks in enumerate(_layer_cfg(sd), start=1):
        for b in range(nblocks):
            stride = 2 if (b == 0 and li > 1) else 1
            x = bottleneck(x, sd, f"layer{li}.{b}", stride, emulate=e, fold=fold)
        stages[f"layer{li}"] = x
    if return_stages:
        return x, stages
    return x


def attnpool(feat_nchw: torch.Tensor, sd: Dict[str, torch.Tensor], num_heads: int = 32,
             prefix: str = "attnpool.") -> torch.Tensor:
    """CLIP ``AttentionPool2d.forward`` (called detached at
    thor_image_features.py:62,112): NCHW->(HW)NC, prepend mean token, add
    positional embedding, ``F.multi_head_attention_forward`` with separate
    q/k/v projections and c_proj as out-proj; returns token 0 -> [B, out]."""
    x = feat_nchw.float()
    B, C, H, W = x.shape
    x = x.reshape(B, C, H * W).permute(2, 0, 1)
    x = torch.cat([x.mean(dim=0, keepdim=True), x], dim=0)
    x = x + sd[prefix + "positional_embedding"].float()[:, None, :]
    out, _ = F.multi_head_attention_forward(
        query=x, key=x, value=x, embed_dim_to_check=C, num_heads=num_heads,
        q_proj_weight=sd[prefix + "q_proj.weight"].float(), k_proj_weight=sd[prefix + "k_proj.weight"].float(),
        v_proj_weight=sd[prefix + "v_proj.weight"].float(), in_proj_weight=None,
        in_proj_bias=torch.cat([sd[prefix + "q_proj.bias"], sd[prefix + "k_proj.bias"],
                                sd[prefix + "v_proj.bias"]]).float(),
        bias_k=None, bias_v=None, add_zero_attn=False, dropout_p=0.0,
        out_proj_weight=sd[prefix + "c_proj.weight"].float(), out_proj_bias=sd[prefix + "c_proj.bias"].float(),
        use_separate_proj_weight=True, training=False, need_weights=False)
    return out[0]


def avgpool_head(feat_nchw: torch.Tensor) -> torch.Tensor:
    """thor_image_features.py:63-66,113: AdaptiveAvgPool2d(1)+Flatten on the
    fp32-cast conv features."""
    return F.adaptive_avg_pool2d(feat_nchw.float(), 1).flatten(1)


def clip_resnet_preprocessor(rgb_nhwc: torch.Tensor, sd: Dict[str, torch.Tensor], pool: bool = False,
                             emulate_bf16: bool = False) -> torch.Tensor:
    """AllenAct ``ClipResNetPreprocessor.process`` / ``ClipResNetEmbedder``
    (SURVEY.md §8a a8): NHWC fp32 -> permute(0,3,1,2) -> stem, layer1..4 under
    no_grad -> optional adaptive_avg_pool2d(1)+flatten -> .float()."""
    with torch.no_grad():
        x = rgb_nhwc.permute(0, 3, 1, 2)
        if x.shape[1] == 1:  # depth input is repeated to 3 channels
            x = x.repeat(1, 3, 1, 1)
        f = rn50_trunk(x, sd, emulate_bf16=emulate_bf16)
        if pool:
            f = F.adaptive_avg_pool2d(f, 1).flatten(1)
        return f.float()


def param_count(sd: Dict[str, torch.Tensor]) -> int:
    return sum(v.numel() for k, v in sd.items()
               if not k.endswith(("running_mean", "running_var", "num_batches_tracked")))
