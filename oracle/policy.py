"""Oracle: AllenAct ``ResnetTensorObjectNavActorCritic`` forward (autograd-able).

Restates allenai/allenact (~v0.5.0, the ``allenact`` branch the reference
points at: ``readme_files/baselines_robothor_objectnav.md:6,25,51``):

  * ``projects/objectnav_baselines/models/object_nav_models.py``
    ``ResnetTensorObjectNavActorCritic`` / ``ResnetTensorGoalEncoder``
  * ``allenact/embodiedai/models/basic_models.py`` ``RNNStateEncoder``,
    ``LinearActorHead``, ``LinearCriticHead``
  * ``allenact/base_abstractions/distributions.py`` ``CategoricalDistr``

  None of it is under /root/reference; the spec is
restated from the published sources and composed from torch-CPU ops.  The
GRU recurrence is written out explicitly and checked against ``torch.nn.GRU``
in tests/test_oracle_policy.py.  Gradients come from torch autograd over this
forward.  TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.nn.functional as F

P = "goal_visual_encoder."


def goal_encoder(feat: torch.Tensor, goal: torch.Tensor, sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    """``ResnetTensorGoalEncoder.forward``: compress resnet tensor
    (1x1 conv C->128, ReLU, 1x1 conv 128->32, ReLU), embed goal id (12x32),
    broadcast over the 7x7 grid, concat on channels (64), combine
    (1x1 conv 64->128, ReLU, 1x1 conv 128->32), flatten C-major -> 1568.
    feat: [B, C, H, W] fp32; goal: [B] int64 -> [B, 32*H*W]."""
    x = F.relu(F.conv2d(feat, sd[P + "resnet_compressor.0.weight"], sd[P + "resnet_compressor.0.bias"]))
    x = F.relu(F.conv2d(x, sd[P + "resnet_compressor.2.weight"], sd[P + "resnet_compressor.2.bias"]))
    emb = F.embedding(goal, sd[P + "embed_class.weight"])  # [B, 32]
    emb = emb.view(emb.shape[0], emb.shape[1], 1, 1).expand(-1, -1, x.shape[-2], x.shape[-1])
    x = torch.cat([x, emb], dim=1)
    x = F.relu(F.conv2d(x, sd[P + "target_obs_combiner.0.weight"], sd[P + "target_obs_combiner.0.bias"]))
    x = F.conv2d(x, sd[P + "target_obs_combiner.2.weight"], sd[P + "target_obs_combiner.2.bias"])
    return x.reshape(x.shape[0], -1)


def dual_goal_encoder(rgb: torch.Tensor, depth: torch.Tensor, goal: torch.Tensor, sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    """``ResnetDualTensorGoalEncoder.forward`` (same upstream file; the RGB-D variant the reference's Habitat readme
    names: ``readme_files/baselines_habitat.md:75``): each stream runs its OWN compressor (``rgb_resnet_compressor`` /
    ``depth_resnet_compressor``) and combiner (``rgb_target_obs_combiner`` / ``depth_target_obs_combiner``) on
    ``cat([compressed, distribute_target(goal)], dim=1)`` with ONE shared ``embed_class``; the result is
    ``cat([rgb_x, depth_x], dim=1)`` flattened C-major -> 2 * 32 * H * W.  Restated from the published source like the
    single-tower encoder (parity unpinned).  rgb, depth: [B, C, H, W]; goal: [B] -> [B, 64*H*W]."""
    emb = F.embedding(goal, sd[P + "embed_class.weight"])
    outs = []
    for tag, feat in (("rgb_", rgb), ("depth_", depth)):
        x = F.relu(F.conv2d(feat, sd[P + tag + "resnet_compressor.0.weight"], sd[P + tag + "resnet_compressor.0.bias"]))
        x = F.relu(F.conv2d(x, sd[P + tag + "resnet_compressor.2.weight"], sd[P + tag + "resnet_compressor.2.bias"]))
        e = emb.view(emb.shape[0], emb.shape[1], 1, 1).expand(-1, -1, x.shape[-2], x.shape[-1])
        x = torch.cat([x, e], dim=1)
        x = F.relu(F.conv2d(x, sd[P + tag + "target_obs_combiner.0.weight"], sd[P + tag + "target_obs_combiner.0.bias"]))
        outs.append(F.conv2d(x, sd[P + tag + "target_obs_combiner.2.weight"], sd[P + tag + "target_obs_combiner.2.bias"]))
    x = torch.cat(outs, dim=1)
    return x.reshape(x.shape[0], -1)


def gru_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    """torch.nn.GRU cell, gate order (r, z, n):
    r = s(W_ir x + b_ir + W_hr h + b_hr); z likewise;
    n = tanh(W_in x + b_in + r*(W_hn h + b_hn)); h' = (1-z)*n + z*h."""
    H = h.shape[-1]
    gi = F.linear(x, w_ih, b_ih)
    gh = F.linear(h, w_hh, b_hh)
    r = torch.sigmoid(gi[..., :H] + gh[..., :H])
    z = torch.sigmoid(gi[..., H:2 * H] + gh[..., H:2 * H])
    n = torch.tanh(gi[..., 2 * H:] + r * gh[..., 2 * H:])
    return (1 - z) * n + z * h


def rnn_state_encoder(x: torch.Tensor, h0: torch.Tensor, masks: torch.Tensor,
                      sd: Dict[str, torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """``RNNStateEncoder.forward`` (1-layer GRU): the hidden state is
    multiplied by ``masks[t]`` before every step (episode reset); the
    sequence form that splits at mask zeros is arithmetically identical.
    x: [T, N, I]; h0: [1, N, H]; masks: [T, N, 1] -> ([T, N, H], [1, N, H])."""
    w_ih, w_hh = sd["state_encoder.rnn.weight_ih_l0"], sd["state_encoder.rnn.weight_hh_l0"]
    b_ih, b_hh = sd["state_encoder.rnn.bias_ih_l0"], sd["state_encoder.rnn.bias_hh_l0"]
    h = h0[0]
    outs = []
    for t in range(x.shape[0]):
        h = gru_cell(x[t], h * masks[t], w_ih, w_hh, b_ih, b_hh)
        outs.append(h)
    return torch.stack(outs, 0), h.unsqueeze(0)


def actor_critic_forward(feat: torch.Tensor, goal: torch.Tensor, h0: torch.Tensor, masks: torch.Tensor,
                         sd: Dict[str, torch.Tensor]):
    """``ResnetTensorObjectNavActorCritic.forward``.
    feat: [T, N, C, H, W] fp32; goal: [T, N] int64; h0: [1, N, hidden];
    masks: [T, N, 1].  Returns (logits [T,N,A], values [T,N,1], h [1,N,hidden])."""
    if isinstance(feat, (tuple, list)):           # RGB-D: (rgb features, depth features) -> ResnetDualTensorGoalEncoder
        rgb, depth = feat
        T, N = rgb.shape[:2]
        x = dual_goal_encoder(rgb.reshape(T * N, *rgb.shape[2:]), depth.reshape(T * N, *depth.shape[2:]),
                              goal.reshape(T * N), sd).view(T, N, -1)
    else:
        T, N = feat.shape[:2]
        x = goal_encoder(feat.reshape(T * N, *feat.shape[2:]), goal.reshape(T * N), sd).view(T, N, -1)
    out, h = rnn_state_encoder(x, h0, masks, sd)
    logits = F.linear(out, sd["actor.linear.weight"], sd["actor.linear.bias"])
    values = F.linear(out, sd["critic.fc.weight"], sd["critic.fc.bias"])
    return logits, values, h


def zeroshot_actor_critic_forward(img_emb: torch.Tensor, goal: torch.Tensor, h0: torch.Tensor, masks: torch.Tensor,
                                  sd: Dict[str, torch.Tensor], goal_table: torch.Tensor):
    """Zero-shot ObjectNav policy (BASELINE config 5; ``readme_files/zeroshot_objectnav.md:3-8``: "CLIP text encoder
    for goal embedding").  The model code lives on the unmounted ``zeroshot-objectnav`` branch, so the FUSION OP IS
    BUILDER-DEFINED AND PARITY-UNPINNED: the L2-normalised CLIP image embedding (AttentionPool2d output, 1024-d)
    is multiplied element-wise with the goal's frozen, L2-normalised CLIP text embedding (the per-dimension
    terms of CLIP's own image-text cosine score) and fed to the same 1-layer GRU + linear actor/critic heads
    as the RoboTHOR policy; there is no trainable goal embedding, compressor or combiner.
    img_emb: [T, N, E] fp32; goal: [T, N] int64; goal_table: [num_goals, E]; h0: [1, N, hidden]; masks: [T, N, 1].
    Returns (logits [T,N,A], values [T,N,1], h [1,N,hidden])."""
    x = F.normalize(img_emb, dim=-1, eps=1e-12) * goal_table[goal]
    out, h = rnn_state_encoder(x, h0, masks, sd)
    logits = F.linear(out, sd["actor.linear.weight"], sd["actor.linear.bias"])
    values = F.linear(out, sd["critic.fc.weight"], sd["critic.fc.bias"])
    return logits, values, h


def categorical_log_prob(logits: torch.Tensor, actions: torch.Tensor) -> torch.Tensor:
    """``CategoricalDistr.log_prob``: log_softmax(logits)[a].  [T,N,A],[T,N] -> [T,N]."""
    return torch.log_softmax(logits, dim=-1).gather(-1, actions.unsqueeze(-1)).squeeze(-1)


def categorical_entropy(logits: torch.Tensor) -> torch.Tensor:
    """``CategoricalDistr.entropy``: -sum p log p.  -> [T,N]."""
    lp = torch.log_softmax(logits, dim=-1)
    return -(lp.exp() * lp).sum(-1)


def param_count(sd) -> int:
    return sum(v.numel() for v in sd.values())
