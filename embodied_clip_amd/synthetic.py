"""Portable, hash-seeded synthetic weights and inputs.

There is no network in the build/bench environment, so pretrained CLIP weights
(`clip.load('RN50')`, reference call site
`primitive_probing/generate_data/thor_image_features.py:57`) are unavailable.
Everything here is a counter-based generator (splitmix64 on ``seed, index``),
*not* ``torch.manual_seed`` streams, so the exact same tensors can be
regenerated on any box / any torch version, and only seeds + expected outputs
need to be stored as golden fixtures.

The state-dict KEY LAYOUTS are the real ones, so a user can drop in real
weights instead:

* RN50 visual tower: openai/CLIP ``ModifiedResNet`` keys (``conv1.weight``,
  ``bn1.running_mean`` ... ``layer4.2.bn3.bias``, ``attnpool.c_proj.bias``)
  -- i.e. ``clip_model.visual.state_dict()`` as used at
  ``thor_image_features.py:59``.
* ViT-B/32 visual tower: openai/CLIP ``VisionTransformer`` keys.
* Policy: AllenAct ``ResnetTensorObjectNavActorCritic`` parameter names
  (SURVEY.md §8b).

This module is data generation only (inputs), shared by tests, bench and
smoke. It contains no model arithmetic.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, Sequence, Tuple

import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)

CLIP_RGB_MEANS = (0.48145466, 0.4578275, 0.40821073)
CLIP_RGB_STDS = (0.26862954, 0.26130258, 0.27577711)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser over uint64 counters."""
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def hash_u64(seed: int, n: int, stream: int = 0) -> np.ndarray:
    base = _splitmix64(np.array([(seed * 0x100000001B3 + stream * 0x9E3779B1) & 0xFFFFFFFFFFFFFFFF],
                                dtype=np.uint64))[0]
    with np.errstate(over="ignore"):
        ctr = (np.arange(n, dtype=np.uint64) + base) & _MASK
    return _splitmix64(ctr)


def hash_uniform(seed: int, n: int, stream: int = 0) -> np.ndarray:
    """float64 uniform in [0, 1) with 53 random bits."""
    return (hash_u64(seed, n, stream) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def hash_normal(seed: int, n: int) -> np.ndarray:
    """float64 standard normal via Box-Muller on two hash streams."""
    u1 = hash_uniform(seed, n, stream=1)
    u2 = hash_uniform(seed, n, stream=2)
    u1 = np.maximum(u1, 1e-300)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def _key_seed(seed: int, key: str) -> int:
    return (seed * 1000003 + zlib.crc32(key.encode())) & 0x7FFFFFFFFFFF


def _normal(seed, key, shape, std) -> torch.Tensor:
    n = int(np.prod(shape))
    return torch.from_numpy((hash_normal(_key_seed(seed, key), n) * std).astype(np.float32)).reshape(shape)


def _uniform(seed, key, shape, lo, hi) -> torch.Tensor:
    n = int(np.prod(shape))
    u = hash_uniform(_key_seed(seed, key), n)
    return torch.from_numpy((lo + (hi - lo) * u).astype(np.float32)).reshape(shape)


# --------------------------------------------------------------------------
# CLIP ModifiedResNet (RN50) visual tower
# --------------------------------------------------------------------------

def _bn(sd, seed, prefix, c, gamma_scale=1.0):
    # gamma~U(0.5,1.5), beta,mu~N(0,0.1), var~U(0.5,1.5)
    sd[prefix + ".weight"] = _uniform(seed, prefix + ".weight", (c,), 0.5, 1.5) * gamma_scale
    sd[prefix + ".bias"] = _normal(seed, prefix + ".bias", (c,), 0.1)
    sd[prefix + ".running_mean"] = _normal(seed, prefix + ".running_mean", (c,), 0.1)
    sd[prefix + ".running_var"] = _uniform(seed, prefix + ".running_var", (c,), 0.5, 1.5)
    sd[prefix + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)


def _conv(sd, seed, key, cout, cin, k):
    fan_in = cin * k * k
    sd[key] = _normal(seed, key, (cout, cin, k, k), (2.0 / fan_in) ** 0.5)


def rn50_visual_state_dict(seed: int = 0, width: int = 64, layers: Sequence[int] = (3, 4, 6, 3),
                           output_dim: int = 1024, heads: int = 32,
                           input_resolution: int = 224) -> "OrderedDict[str, torch.Tensor]":
    """Synthetic ``clip_model.visual.state_dict()`` for ModifiedResNet.

    RN50: width 64, layers (3,4,6,3), embed 2048, 32 heads, out 1024 ->
    38,316,896 parameters.
    """
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    _conv(sd, seed, "conv1.weight", width // 2, 3, 3)
    _bn(sd, seed, "bn1", width // 2)
    _conv(sd, seed, "conv2.weight", width // 2, width // 2, 3)
    _bn(sd, seed, "bn2", width // 2)
    _conv(sd, seed, "conv3.weight", width, width // 2, 3)
    _bn(sd, seed, "bn3", width)
    inplanes = width
    for li, (nblocks, mult) in enumerate(zip(layers, (1, 2, 4, 8)), start=1):
        planes = width * mult
        for b in range(nblocks):
            stride = 2 if (b == 0 and li > 1) else 1
            p = f"layer{li}.{b}"
            _conv(sd, seed, p + ".conv1.weight", planes, inplanes, 1)
            _bn(sd, seed, p + ".bn1", planes)
            _conv(sd, seed, p + ".conv2.weight", planes, planes, 3)
            _bn(sd, seed, p + ".bn2", planes)
            _conv(sd, seed, p + ".conv3.weight", planes * 4, planes, 1)
            # damp the residual branch (and the projection shortcut) so the 16 stacked blocks keep
            # activations O(1) like the real CLIP tower; otherwise attention logits of AttentionPool2d
            # reach the hundreds and the softmax becomes ill-conditioned under ANY reduced precision
            _bn(sd, seed, p + ".bn3", planes * 4, gamma_scale=0.3)
            if stride > 1 or inplanes != planes * 4:
                _conv(sd, seed, p + ".downsample.0.weight", planes * 4, inplanes, 1)
                _bn(sd, seed, p + ".downsample.1", planes * 4, gamma_scale=0.7)
            inplanes = planes * 4
    embed = width * 32
    sp = input_resolution // 32
    sd["attnpool.positional_embedding"] = _normal(seed, "attnpool.positional_embedding",
                                                  (sp * sp + 1, embed), embed ** -0.5)
    for name, od in (("k_proj", embed), ("q_proj", embed), ("v_proj", embed), ("c_proj", output_dim)):
        sd[f"attnpool.{name}.weight"] = _normal(seed, f"attnpool.{name}.weight", (od, embed), embed ** -0.5)
        sd[f"attnpool.{name}.bias"] = _normal(seed, f"attnpool.{name}.bias", (od,), 0.02)
    return sd


# --------------------------------------------------------------------------
# torchvision ResNet-50 (the ImageNet branch of the feature scripts)
# --------------------------------------------------------------------------

IMAGENET_RGB_MEANS = (0.485, 0.456, 0.406)
IMAGENET_RGB_STDS = (0.229, 0.224, 0.225)


def tv_resnet_state_dict(seed: int = 0, width: int = 64, layers: Sequence[int] = (3, 4, 6, 3),
                         with_fc: bool = False) -> "OrderedDict[str, torch.Tensor]":
    """Synthetic ``torchvision.models.resnet50().state_dict()`` (the network behind ``imagenet_conv`` /
    ``imagenet_avgpool``: primitive_probing/generate_data/thor_image_features.py:46-49).  Key layout: ``conv1.weight``
    [w,3,7,7], ``bn1.*``, ``layer{1..4}.{b}.conv{1,2,3}.weight`` / ``bn{1,2,3}.*`` / ``downsample.{0,1}.*`` (+ ``fc.*``
    with ``with_fc``; the reference drops avgpool and fc).  ResNet-50: 23,508,032 parameters without fc."""
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    _conv(sd, seed, "tv.conv1.weight", width, 3, 7)
    sd["conv1.weight"] = sd.pop("tv.conv1.weight")
    _bn(sd, seed, "bn1", width)
    inplanes = width
    for li, (nblocks, mult) in enumerate(zip(layers, (1, 2, 4, 8)), start=1):
        planes = width * mult
        for b in range(nblocks):
            stride = 2 if (b == 0 and li > 1) else 1
            p = f"layer{li}.{b}"
            _conv(sd, seed, p + ".conv1.weight", planes, inplanes, 1)
            _bn(sd, seed, p + ".bn1", planes)
            _conv(sd, seed, p + ".conv2.weight", planes, planes, 3)
            _bn(sd, seed, p + ".bn2", planes)
            _conv(sd, seed, p + ".conv3.weight", planes * 4, planes, 1)
            _bn(sd, seed, p + ".bn3", planes * 4, gamma_scale=0.3)      # damped residual branch, as in rn50_visual_state_dict
            if stride > 1 or inplanes != planes * 4:
                _conv(sd, seed, p + ".downsample.0.weight", planes * 4, inplanes, 1)
                _bn(sd, seed, p + ".downsample.1", planes * 4, gamma_scale=0.7)
            inplanes = planes * 4
    if with_fc:
        sd["fc.weight"] = _normal(seed, "fc.weight", (1000, inplanes), inplanes ** -0.5)
        sd["fc.bias"] = _normal(seed, "fc.bias", (1000,), 0.01)
    return sd


def normalize_rgb_imagenet(u8: torch.Tensor) -> torch.Tensor:
    """ToTensor + Normalize(ImageNet mean / std) of ``resnet_preprocess`` (thor_image_features.py:36-44), NHWC."""
    mean = torch.tensor(IMAGENET_RGB_MEANS, dtype=torch.float32, device=u8.device)
    std = torch.tensor(IMAGENET_RGB_STDS, dtype=torch.float32, device=u8.device)
    return (u8.to(torch.float32) / 255.0 - mean) / std


# --------------------------------------------------------------------------
# CLIP VisionTransformer (ViT-B/32) visual tower
# --------------------------------------------------------------------------

def vit_visual_state_dict(seed: int = 0, width: int = 768, layers: int = 12, heads: int = 12,
                          patch_size: int = 32, input_resolution: int = 224,
                          output_dim: int = 512) -> "OrderedDict[str, torch.Tensor]":
    """Synthetic ``clip_model.visual.state_dict()`` for VisionTransformer.

    ViT-B/32 -> 87,849,216 parameters.
    """
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    scale = width ** -0.5
    grid = input_resolution // patch_size
    sd["class_embedding"] = _normal(seed, "class_embedding", (width,), scale)
    sd["positional_embedding"] = _normal(seed, "positional_embedding", (grid * grid + 1, width), scale)
    sd["proj"] = _normal(seed, "proj", (width, output_dim), scale)
    sd["conv1.weight"] = _normal(seed, "conv1.weight", (width, 3, patch_size, patch_size),
                                 (3 * patch_size * patch_size) ** -0.5)
    for ln in ("ln_pre", "ln_post"):
        sd[ln + ".weight"] = _uniform(seed, ln + ".weight", (width,), 0.8, 1.2)
        sd[ln + ".bias"] = _normal(seed, ln + ".bias", (width,), 0.05)
    proj_std = scale * ((2 * layers) ** -0.5)
    fc_std = (2 * width) ** -0.5
    for i in range(layers):
        p = f"transformer.resblocks.{i}"
        sd[p + ".attn.in_proj_weight"] = _normal(seed, p + ".attn.in_proj_weight", (3 * width, width), scale)
        sd[p + ".attn.in_proj_bias"] = _normal(seed, p + ".attn.in_proj_bias", (3 * width,), 0.02)
        sd[p + ".attn.out_proj.weight"] = _normal(seed, p + ".attn.out_proj.weight", (width, width), proj_std)
        sd[p + ".attn.out_proj.bias"] = _normal(seed, p + ".attn.out_proj.bias", (width,), 0.02)
        sd[p + ".ln_1.weight"] = _uniform(seed, p + ".ln_1.weight", (width,), 0.8, 1.2)
        sd[p + ".ln_1.bias"] = _normal(seed, p + ".ln_1.bias", (width,), 0.05)
        sd[p + ".mlp.c_fc.weight"] = _normal(seed, p + ".mlp.c_fc.weight", (4 * width, width), fc_std)
        sd[p + ".mlp.c_fc.bias"] = _normal(seed, p + ".mlp.c_fc.bias", (4 * width,), 0.02)
        sd[p + ".mlp.c_proj.weight"] = _normal(seed, p + ".mlp.c_proj.weight", (width, 4 * width), proj_std)
        sd[p + ".mlp.c_proj.bias"] = _normal(seed, p + ".mlp.c_proj.bias", (width,), 0.02)
        sd[p + ".ln_2.weight"] = _uniform(seed, p + ".ln_2.weight", (width,), 0.8, 1.2)
        sd[p + ".ln_2.bias"] = _normal(seed, p + ".ln_2.bias", (width,), 0.05)
    return sd


def text_state_dict(seed: int = 0, width: int = 512, layers: int = 12, heads: int = 8, context_length: int = 77,
                    vocab_size: int = 49408, embed_dim: int = 1024) -> "OrderedDict[str, torch.Tensor]":
    """Synthetic CLIP text tower (``token_embedding`` .. ``text_projection``) with CLIP's init scales
    (``CLIP.initialize_parameters``).  RN50 CLIP: width 512, 12 layers, 8 heads, ctx 77, vocab 49,408, out 1024."""
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    sd["token_embedding.weight"] = _normal(seed, "token_embedding.weight", (vocab_size, width), 0.02)
    sd["positional_embedding"] = _normal(seed, "text.positional_embedding", (context_length, width), 0.01)
    attn_std = width ** -0.5
    proj_std = attn_std * ((2 * layers) ** -0.5)
    fc_std = (2 * width) ** -0.5
    for i in range(layers):
        p = f"transformer.resblocks.{i}"
        sd[p + ".attn.in_proj_weight"] = _normal(seed, "text." + p + ".attn.in_proj_weight", (3 * width, width), attn_std)
        sd[p + ".attn.in_proj_bias"] = _normal(seed, "text." + p + ".attn.in_proj_bias", (3 * width,), 0.02)
        sd[p + ".attn.out_proj.weight"] = _normal(seed, "text." + p + ".attn.out_proj.weight", (width, width), proj_std)
        sd[p + ".attn.out_proj.bias"] = _normal(seed, "text." + p + ".attn.out_proj.bias", (width,), 0.02)
        sd[p + ".ln_1.weight"] = _uniform(seed, "text." + p + ".ln_1.weight", (width,), 0.8, 1.2)
        sd[p + ".ln_1.bias"] = _normal(seed, "text." + p + ".ln_1.bias", (width,), 0.05)
        sd[p + ".mlp.c_fc.weight"] = _normal(seed, "text." + p + ".mlp.c_fc.weight", (4 * width, width), fc_std)
        sd[p + ".mlp.c_fc.bias"] = _normal(seed, "text." + p + ".mlp.c_fc.bias", (4 * width,), 0.02)
        sd[p + ".mlp.c_proj.weight"] = _normal(seed, "text." + p + ".mlp.c_proj.weight", (width, 4 * width), proj_std)
        sd[p + ".mlp.c_proj.bias"] = _normal(seed, "text." + p + ".mlp.c_proj.bias", (width,), 0.02)
        sd[p + ".ln_2.weight"] = _uniform(seed, "text." + p + ".ln_2.weight", (width,), 0.8, 1.2)
        sd[p + ".ln_2.bias"] = _normal(seed, "text." + p + ".ln_2.bias", (width,), 0.05)
    sd["ln_final.weight"] = _uniform(seed, "ln_final.weight", (width,), 0.8, 1.2)
    sd["ln_final.bias"] = _normal(seed, "ln_final.bias", (width,), 0.05)
    sd["text_projection"] = _normal(seed, "text_projection", (width, embed_dim), attn_std)
    return sd


def synthetic_tokens(seed: int, n: int, context_length: int = 77, vocab_size: int = 49408) -> torch.Tensor:
    """CLIP-tokenizer-shaped ids: <SOT> = vocab-2, 1..8 word tokens, <EOT> = vocab-1 (the arg-max), zero padding."""
    k = hash_u64(seed, n * (context_length + 1), stream=9).reshape(n, context_length + 1)
    out = np.zeros((n, context_length), dtype=np.int64)
    for i in range(n):
        nw = 1 + int(k[i, 0] % np.uint64(8))
        out[i, 0] = vocab_size - 2
        out[i, 1:1 + nw] = (k[i, 1:1 + nw] % np.uint64(vocab_size - 2)).astype(np.int64)
        out[i, 1 + nw] = vocab_size - 1
    return torch.from_numpy(out)


# --------------------------------------------------------------------------
# AllenAct ResnetTensorObjectNavActorCritic policy
# --------------------------------------------------------------------------

POLICY_PARAM_ORDER: Tuple[str, ...] = (
    "goal_visual_encoder.embed_class.weight",
    "goal_visual_encoder.resnet_compressor.0.weight",
    "goal_visual_encoder.resnet_compressor.0.bias",
    "goal_visual_encoder.resnet_compressor.2.weight",
    "goal_visual_encoder.resnet_compressor.2.bias",
    "goal_visual_encoder.target_obs_combiner.0.weight",
    "goal_visual_encoder.target_obs_combiner.0.bias",
    "goal_visual_encoder.target_obs_combiner.2.weight",
    "goal_visual_encoder.target_obs_combiner.2.bias",
    "state_encoder.rnn.weight_ih_l0",
    "state_encoder.rnn.weight_hh_l0",
    "state_encoder.rnn.bias_ih_l0",
    "state_encoder.rnn.bias_hh_l0",
    "actor.linear.weight",
    "actor.linear.bias",
    "critic.fc.weight",
    "critic.fc.bias",
)


# [U] ResnetDualTensorGoalEncoder (RGB + depth, ``dual=1``): the same flat layout with the RGB stream's compressor /
# combiner in the single encoder's slots and the depth stream's eight tensors appended after the heads
_DUAL_RGB = {k: k.replace("resnet_compressor", "rgb_resnet_compressor").replace("target_obs_combiner", "rgb_target_obs_combiner")
             for k in POLICY_PARAM_ORDER[1:9]}
POLICY_PARAM_ORDER_DUAL: Tuple[str, ...] = tuple(_DUAL_RGB.get(k, k) for k in POLICY_PARAM_ORDER) + tuple(
    k.replace("resnet_compressor", "depth_resnet_compressor").replace("target_obs_combiner", "depth_target_obs_combiner")
    for k in POLICY_PARAM_ORDER[1:9])


def policy_param_order(dual: int = 0) -> Tuple[str, ...]:
    return POLICY_PARAM_ORDER_DUAL if dual else POLICY_PARAM_ORDER


def policy_param_shapes(in_channels: int = 2048, spatial: int = 7, hidden: int = 512, goal_dims: int = 32,
                        num_goals: int = 12, num_actions: int = 6, compress_hid: int = 128,
                        compress_out: int = 32, comb_hid: int = 128, comb_out: int = 32, fusion: int = 0, dual: int = 0):
    """``fusion=1`` (zero-shot dual-encoder policy): the goal-embedding / compressor / combiner tensors have zero
    elements and the GRU reads the ``in_channels``-wide fused embedding.  ``dual=1``: RGB + depth streams (25 tensors,
    GRU input 2 * comb_out * spatial^2)."""
    if dual:
        assert not fusion
        one = policy_param_shapes(in_channels, spatial, hidden, goal_dims, num_goals, num_actions, compress_hid, compress_out,
                                  comb_hid, comb_out)
        vals = list(one.values())
        out = OrderedDict(zip(POLICY_PARAM_ORDER_DUAL[:17], vals))
        out["state_encoder.rnn.weight_ih_l0"] = (3 * hidden, 2 * comb_out * spatial * spatial)
        for k, v in zip(POLICY_PARAM_ORDER_DUAL[17:], vals[1:9]):
            out[k] = v
        return out
    flat = comb_out * spatial * spatial
    if fusion:
        z = (0,)
        return OrderedDict([(k, z) for k in POLICY_PARAM_ORDER[:9]] + [
            ("state_encoder.rnn.weight_ih_l0", (3 * hidden, in_channels)),
            ("state_encoder.rnn.weight_hh_l0", (3 * hidden, hidden)),
            ("state_encoder.rnn.bias_ih_l0", (3 * hidden,)),
            ("state_encoder.rnn.bias_hh_l0", (3 * hidden,)),
            ("actor.linear.weight", (num_actions, hidden)),
            ("actor.linear.bias", (num_actions,)),
            ("critic.fc.weight", (1, hidden)),
            ("critic.fc.bias", (1,)),
        ])
    return OrderedDict([
        ("goal_visual_encoder.embed_class.weight", (num_goals, goal_dims)),
        ("goal_visual_encoder.resnet_compressor.0.weight", (compress_hid, in_channels, 1, 1)),
        ("goal_visual_encoder.resnet_compressor.0.bias", (compress_hid,)),
        ("goal_visual_encoder.resnet_compressor.2.weight", (compress_out, compress_hid, 1, 1)),
        ("goal_visual_encoder.resnet_compressor.2.bias", (compress_out,)),
        ("goal_visual_encoder.target_obs_combiner.0.weight", (comb_hid, compress_out + goal_dims, 1, 1)),
        ("goal_visual_encoder.target_obs_combiner.0.bias", (comb_hid,)),
        ("goal_visual_encoder.target_obs_combiner.2.weight", (comb_out, comb_hid, 1, 1)),
        ("goal_visual_encoder.target_obs_combiner.2.bias", (comb_out,)),
        ("state_encoder.rnn.weight_ih_l0", (3 * hidden, flat)),
        ("state_encoder.rnn.weight_hh_l0", (3 * hidden, hidden)),
        ("state_encoder.rnn.bias_ih_l0", (3 * hidden,)),
        ("state_encoder.rnn.bias_hh_l0", (3 * hidden,)),
        ("actor.linear.weight", (num_actions, hidden)),
        ("actor.linear.bias", (num_actions,)),
        ("critic.fc.weight", (1, hidden)),
        ("critic.fc.bias", (1,)),
    ])


def policy_state_dict(seed: int = 0, **kw) -> "OrderedDict[str, torch.Tensor]":
    """Synthetic policy parameters; RoboTHOR default = 3,480,775 params."""
    shapes = policy_param_shapes(**kw)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, shp in shapes.items():
        if int(np.prod(shp)) == 0:
            sd[k] = torch.zeros(shp)
        elif k.endswith("bias") or "bias_" in k:
            sd[k] = _normal(seed, k, shp, 0.02)
        elif "embed_class" in k:
            sd[k] = _normal(seed, k, shp, 1.0)
        else:
            fan_in = int(np.prod(shp[1:]))
            std = (1.0 / fan_in) ** 0.5
            if k.startswith("actor."):
                std *= 0.5
            sd[k] = _normal(seed, k, shp, std)
    return sd


# --------------------------------------------------------------------------
# Synthetic rollout inputs (SURVEY.md §8d "Synthetic inputs")
# --------------------------------------------------------------------------

def synthetic_rgb_u8(seed: int, n: int, res: int = 224) -> torch.Tensor:
    """uint8 ~ U{0..255}, [n, res, res, 3] (HWC, as AI2-THOR frames are)."""
    v = (hash_u64(seed, n * res * res * 3, stream=7) >> np.uint64(56)).astype(np.uint8)
    return torch.from_numpy(v).reshape(n, res, res, 3)


def normalize_rgb(u8: torch.Tensor) -> torch.Tensor:
    """The sensor's wire form: fp32 NHWC, (u8/255 - mean)/std with CLIP consts."""
    mean = torch.tensor(CLIP_RGB_MEANS, dtype=torch.float32, device=u8.device)
    std = torch.tensor(CLIP_RGB_STDS, dtype=torch.float32, device=u8.device)
    return (u8.to(torch.float32) / 255.0 - mean) / std


def synthetic_rgb(seed: int, n: int, res: int = 224) -> torch.Tensor:
    return normalize_rgb(synthetic_rgb_u8(seed, n, res))


def synthetic_goals(seed: int, shape, num_goals: int = 12) -> torch.Tensor:
    n = int(np.prod(shape))
    return torch.from_numpy((hash_u64(seed, n, stream=11) % np.uint64(num_goals)).astype(np.int64)).reshape(shape)


def synthetic_masks(seed: int, T: int, N: int, p_reset: float = 0.01) -> torch.Tensor:
    """masks[t,n,0] = 0 with prob p_reset (episode reset), else 1. fp32 [T,N,1]."""
    u = hash_uniform(seed, T * N, stream=13)
    return torch.from_numpy((u >= p_reset).astype(np.float32)).reshape(T, N, 1)


def synthetic_rewards(seed: int, masks_next: torch.Tensor) -> torch.Tensor:
    """-0.01 step penalty; +10 on the step before a reset w.p. 0.3. [T,N,1]."""
    T, N, _ = masks_next.shape
    u = torch.from_numpy(hash_uniform(seed, T * N, stream=17).astype(np.float32)).reshape(T, N, 1)
    r = torch.full((T, N, 1), -0.01, dtype=torch.float32)
    r = r + 10.0 * ((masks_next == 0) & (u < 0.3)).to(torch.float32)
    return r
