"""Oracle: the linear probe of ``primitive_probing/train.py`` (BASELINE config 1).

Follows ``/root/reference/primitive_probing/train.py:14-113`` line by line
(LinearEncoder) without pytorch-lightning / torchmetrics, which are not
installed here:

  * model heads ``train.py:19-49`` (Linear+Sigmoid / Linear+Softmax(dim=1) /
    AdaptiveAvgPool2d(3,3)+Conv1x1+Flatten(2)+Sigmoid)
  * ``compute_loss`` ``train.py:56-81`` incl. the reference's *double
    softmax* for ``free_space`` (``F.cross_entropy`` applied to Softmax
    output, ``train.py:35,78``) -- reproduced, not "fixed"
  * label clamp ``y[y > 10] = 10`` (``train.py:65``; ``constants.py:3``)
  * metrics ``train.py:84-90`` (micro-F1 at threshold 0.5 standing in for the
    unpinned ``torchmetrics.functional.f1``)
  * Adam lr 1e-3 (``train.py:111-113,137``)

TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

NUM_TARGET_OBJECTS = 52   # len(constants.target_objects), constants.py:1
MAX_FORWARD_STEPS = 10    # constants.py:3


def head_dims(embedding_type: str, prediction_type: str):
    """train.py:19-35."""
    if prediction_type in ("object_presence", "reachability", "free_space"):
        assert embedding_type in ("imagenet_avgpool", "clip_avgpool", "clip_attnpool")
        in_dim = 1024 if embedding_type == "clip_attnpool" else 2048
        out_dim = {"object_presence": NUM_TARGET_OBJECTS, "reachability": 110,
                   "free_space": MAX_FORWARD_STEPS + 1}[prediction_type]
        return in_dim, out_dim
    if prediction_type == "object_localization":
        assert embedding_type in ("imagenet_avgpool", "clip_avgpool")
        return 2048, NUM_TARGET_OBJECTS
    raise NotImplementedError()


def forward(x, weight, bias, prediction_type: str):
    """train.py:37-49,53-54."""
    if prediction_type == "object_localization":
        y = F.adaptive_avg_pool2d(x, (3, 3))
        y = F.conv2d(y, weight.view(weight.shape[0], -1, 1, 1), bias)
        return torch.sigmoid(y.flatten(2))
    y = F.linear(x, weight, bias)
    if prediction_type == "free_space":
        return torch.softmax(y, dim=1)
    return torch.sigmoid(y)


def compute_loss(x, y, weight, bias, prediction_type: str, eval: bool = False):
    """train.py:56-92."""
    if prediction_type == "object_localization":
        y = y.flatten(1)
    elif prediction_type == "reachability":
        obj_idx, y = y
        obj_idx = obj_idx.tolist()
    elif prediction_type == "free_space":
        y = y.clone()
        y[y > MAX_FORWARD_STEPS] = MAX_FORWARD_STEPS
    y_pred = forward(x, weight, bias, prediction_type)
    if prediction_type == "object_localization":
        y_pred = y_pred.permute(0, 2, 1).flatten(1)
    elif prediction_type == "reachability":
        y_pred = y_pred[range(len(obj_idx)), obj_idx]
    if prediction_type in ("object_presence", "object_localization", "reachability"):
        loss = F.binary_cross_entropy(y_pred, y.float())
    else:
        loss = F.cross_entropy(y_pred, y)  # softmax applied twice, as in the reference
    if not eval:
        return loss
    if prediction_type in ("object_presence", "object_localization"):
        p = (y_pred > 0.5)
        t = y.bool()
        tp = (p & t).sum().float()
        acc = 2 * tp / (p.sum() + t.sum()).clamp(min=1).float()
    elif prediction_type == "reachability":
        acc = ((y_pred > 0.5) == y).float().mean()
    else:
        acc = (torch.argmax(y_pred, dim=1) == y).float().mean()
    return loss, {"accuracy": acc}
