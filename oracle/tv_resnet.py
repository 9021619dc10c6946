"""Oracle: torchvision ResNet-50 (v1.5) trunk -- the ImageNet half of the reference's feature scripts.

TEST INFRASTRUCTURE ONLY (imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product
never imports it).

Restates ``torchvision.models.resnet50`` (torchvision 0.8.2, pinned by ``primitive_probing/environment.yml``) as the
reference drives it:

  * ``primitive_probing/generate_data/thor_image_features.py:46-49`` /
    ``reachable_image_features.py:48-51``: ``Sequential(*list(resnet50(pretrained=True).children())[:-2])`` =
    conv1 (7x7 s2 p3) - bn1 - relu - maxpool (3x3 s2 p1) - layer1..4, run in fp32 under ``freeze_model``
    (eval-mode BatchNorm, ``:26-33``);
  * ``:51-54``, ``:102-106``: ``imagenet_conv`` = the [2048,7,7] map, ``imagenet_avgpool`` =
    AdaptiveAvgPool2d(1)+Flatten of it;
  * ``:36-44``: Resize(224, BICUBIC) + CenterCrop(224) + ToTensor + Normalize(ImageNet mean / std).

torchvision's Bottleneck (v1.5): conv1 1x1 - bn - relu, conv2 3x3 **stride s** pad 1 - bn - relu, conv3 1x1 - bn;
identity = conv 1x1 **stride s** - bn when the shape changes; relu(out + identity).  (CLIP's ModifiedResNet differs in
the stem and in how it strides: oracle/clip_resnet.py.)

PINNED: ``tests/test_oracle_tv_resnet.py`` checks this restatement against HuggingFace ``transformers``'
``ResNetModel`` (an independent implementation of the same v1.5 network, installed in the image) on shared random
weights, whole network, to fp32 rounding.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

from .clip_resnet import _conv_bn, _layer_cfg, _r

IMAGENET_RGB_MEANS = (0.485, 0.456, 0.406)     # thor_image_features.py:41
IMAGENET_RGB_STDS = (0.229, 0.224, 0.225)      # thor_image_features.py:42


def tv_bottleneck(x, sd, p, stride, emulate=False, fold=True):
    """torchvision ``Bottleneck.forward`` (stride on the 3x3 conv: ResNet v1.5)."""
    out = _r(F.relu(_conv_bn(x, sd, p + ".conv1", p + ".bn1", emulate=emulate, fold=fold)), emulate)
    out = _r(F.relu(_conv_bn(out, sd, p + ".conv2", p + ".bn2", stride=stride, padding=1, emulate=emulate, fold=fold)), emulate)
    out = _conv_bn(out, sd, p + ".conv3", p + ".bn3", emulate=emulate, fold=fold)
    if (p + ".downsample.0.weight") in sd:
        idt = _r(_conv_bn(x, sd, p + ".downsample.0", p + ".downsample.1", stride=stride, emulate=emulate, fold=fold), emulate)
    else:
        idt = x
    return _r(F.relu(out + idt), emulate)


def tv_resnet_trunk(x_nchw: torch.Tensor, sd: Dict[str, torch.Tensor], emulate_bf16: bool = False, fold: bool = True,
                    return_stages: bool = False):
    """``Sequential(*list(resnet50.children())[:-2])`` (thor_image_features.py:47): fp32 [B,3,R,R] ImageNet-normalised
    -> fp32 [B, 32*w, R/32, R/32].  ``emulate_bf16`` rounds the frame, the folded weights and every layer output to
    bf16 (fp32 accumulation) where the HIP path rounds."""
    e = emulate_bf16
    stages = {}
    x = _r(x_nchw.float(), e)
    x = _r(F.relu(_conv_bn(x, sd, "conv1", "bn1", stride=2, padding=3, emulate=e, fold=fold)), e)
    stages["conv1"] = x
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    stages["stem"] = x
    for li, nblocks in enumerate(_layer_cfg(sd), start=1):
        for b in range(nblocks):
            stride = 2 if (b == 0 and li > 1) else 1
            x = tv_bottleneck(x, sd, f"layer{li}.{b}", stride, emulate=e, fold=fold)
        stages[f"layer{li}"] = x
    if return_stages:
        return x, stages
    return x


def imagenet_features(frames_nhwc_f32: torch.Tensor, sd: Dict[str, torch.Tensor], emulate_bf16: bool = False):
    """(imagenet_conv [B,2048,7,7], imagenet_avgpool [B,2048]) of thor_image_features.py:102-106 for frames that are
    already resized, cropped and normalised (fp32 NHWC)."""
    with torch.no_grad():
        f = tv_resnet_trunk(frames_nhwc_f32.permute(0, 3, 1, 2), sd, emulate_bf16=emulate_bf16)
        return f, F.adaptive_avg_pool2d(f, 1).flatten(1)


def hf_resnet_to_torchvision_keys(hf_sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """HuggingFace ``ResNetModel.state_dict()`` -> torchvision ``resnet50`` key layout (test helper for the pin)."""
    out: Dict[str, torch.Tensor] = {}
    bn = {"weight": "weight", "bias": "bias", "running_mean": "running_mean", "running_var": "running_var",
          "num_batches_tracked": "num_batches_tracked"}
    for k, v in hf_sd.items():
        parts = k.split(".")
        if parts[0] == "embedder":
            kind, leaf = parts[2], parts[3]
            out[("conv1." if kind == "convolution" else "bn1.") + (leaf if kind == "convolution" else bn[leaf])] = v
            continue
        assert parts[0] == "encoder" and parts[1] == "stages", k
        li, b = int(parts[2]) + 1, int(parts[4])
        p = f"layer{li}.{b}."
        if parts[5] == "shortcut":
            kind, leaf = parts[6], parts[7]
            out[p + ("downsample.0." if kind == "convolution" else "downsample.1.") + leaf] = v
        else:
            i, kind, leaf = int(parts[6]) + 1, parts[7], parts[8]
            out[p + (f"conv{i}." if kind == "convolution" else f"bn{i}.") + leaf] = v
    return out
