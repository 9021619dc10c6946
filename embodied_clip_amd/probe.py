"""Linear probe on cached CLIP features (BASELINE config 1) -- the HIP side of
``primitive_probing/train.py``'s ``LinearEncoder`` (train.py:14-113).

Same constructor arguments, attribute names and step methods as the reference's LightningModule so a parity test
reads like a test of the reference; pytorch-lightning itself (trainer loop, checkpointing, TensorBoard) is control
plane and not rebuilt.  Everything runs on the MI355X through the C-ABI:

  forward          ec_gemm_f32 (+bias)  [+ ec_probe_pool3 for object_localization]   train.py:37-54
  compute_loss     ec_probe_head (activation, loss, metric counts, d loss/d logits)   train.py:56-92
  training_step    + dW = dlogits^T x (ec_gemm_f32, TN), db, Adam (ec_clip_adam_step, clipping off)
                                                                                      train.py:94-97,111-113

There is no CPU fallback: constructing a ``LinearEncoder`` without the HIP library raises.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, Optional, Tuple

import torch

from . import _lib
from . import synthetic as syn

NUM_TARGET_OBJECTS = 52   # len(constants.target_objects), primitive_probing/constants.py:1
MAX_FORWARD_STEPS = 10    # primitive_probing/constants.py:3

_MODE = {"object_presence": 0, "object_localization": 0, "reachability": 1, "free_space": 2}


def head_dims(embedding_type: str, prediction_type: str) -> Tuple[int, int]:
    """train.py:19-49: (input_dim, output_dim) with the reference's asserts."""
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


class LinearEncoder:
    """``LinearEncoder(embedding_type, prediction_type, batch_size, lr)`` (train.py:14-17).

    Parameters live in one flat fp32 bucket ``[weight (C*K) | bias (C)]``; ``state_dict()`` uses the reference's
    names (``model.0.weight`` / ``model.0.bias``; ``model.1.*`` for the localization Conv2d, whose weight is
    ``[52, 2048, 1, 1]``)."""

    def __init__(self, embedding_type: str, prediction_type: str, batch_size: int, lr: float, device="cuda:0",
                 seed: int = 1, state_dict: Optional[Dict[str, torch.Tensor]] = None):
        self.lib = _lib.load()
        self.hparams = SimpleNamespace(embedding_type=embedding_type, prediction_type=prediction_type,
                                       batch_size=batch_size, lr=lr)
        self.device = torch.device(device)
        K, Cc = head_dims(embedding_type, prediction_type)
        self.K, self.C = K, Cc
        self._loc = prediction_type == "object_localization"
        self._prefix = "model.1" if self._loc else "model.0"
        self.flat = torch.empty(Cc * K + Cc, dtype=torch.float32, device=self.device)
        self.weight = self.flat[:Cc * K].view(Cc, K)
        self.bias = self.flat[Cc * K:]
        if state_dict is None:   # nn.Linear / nn.Conv2d default init: U(-1/sqrt(fan_in), 1/sqrt(fan_in)), portable seed
            b = 1.0 / math.sqrt(K)
            self.weight.copy_(syn._uniform(seed, "probe.weight", (Cc, K), -b, b))
            self.bias.copy_(syn._uniform(seed, "probe.bias", (Cc,), -b, b))
        else:
            self.load_state_dict(state_dict)
        self.grads = torch.zeros_like(self.flat)
        self._dW = self.grads[:Cc * K].view(Cc, K)
        self._db = self.grads[Cc * K:]
        self._out5 = torch.zeros(5, dtype=torch.float64, device=self.device)
        self._opt = None
        self.logged: Dict[str, float] = {}

    # ---- nn.Module-ish surface -----------------------------------------------------------------
    def state_dict(self) -> Dict[str, torch.Tensor]:
        w = self.weight.detach().clone()
        return {f"{self._prefix}.weight": w.view(self.C, self.K, 1, 1) if self._loc else w,
                f"{self._prefix}.bias": self.bias.detach().clone()}

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        self.weight.copy_(sd[f"{self._prefix}.weight"].reshape(self.C, self.K).to(self.device, torch.float32))
        self.bias.copy_(sd[f"{self._prefix}.bias"].to(self.device, torch.float32))

    def parameters(self):
        return [self.weight, self.bias]

    # ---- forward ---------------------------------------------------------------------------------
    def _rows(self, x: torch.Tensor) -> torch.Tensor:
        x = x.to(self.device, torch.float32).contiguous()
        if not self._loc:
            assert x.dim() == 2 and x.shape[1] == self.K, x.shape
            return x
        assert x.dim() == 4 and x.shape[1] == self.K, x.shape          # cached clip_conv [B, 2048, 7, 7]
        B, _, H, W = x.shape
        rows = torch.empty(B * 9, self.K, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.ec_probe_pool3(x.data_ptr(), rows.data_ptr(), B, self.K, H, W, _lib.stream_ptr()),
                   "ec_probe_pool3")
        return rows

    def _logits(self, rows: torch.Tensor) -> torch.Tensor:
        R = rows.shape[0]
        z = torch.empty(R, self.C, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.ec_gemm_f32(rows.data_ptr(), self.weight.data_ptr(), z.data_ptr(), R, self.C, self.K,
                                        self.K, 1, 1, self.K, self.C, 0, self.bias.data_ptr(), None, None, 0, None,
                                        None, 1, _lib.stream_ptr()), "ec_gemm_f32(probe fwd)")
        return z

    def _head(self, z, labels, idx, pred, dz, db):
        R = z.shape[0]
        _lib.check(self.lib.ec_probe_head(_MODE[self.hparams.prediction_type], z.data_ptr(), labels.data_ptr(),
                                          _lib.ptr(idx), R, self.C,
                                          MAX_FORWARD_STEPS if self.hparams.prediction_type == "free_space" else -1,
                                          _lib.ptr(pred), _lib.ptr(dz), _lib.ptr(db), self._out5.data_ptr(),
                                          _lib.stream_ptr()), "ec_probe_head")

    @_lib.on_device
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Probabilities in the reference's layout: [B, C], or [B, 52, 9] for object_localization."""
        rows = self._rows(x)
        z = self._logits(rows)
        pred = torch.empty_like(z)
        dummy = torch.zeros(z.shape if _MODE[self.hparams.prediction_type] == 0 else z.shape[:1], dtype=torch.int64,
                            device=self.device)
        idx = torch.zeros(z.shape[0], dtype=torch.int64, device=self.device)
        self._head(z, dummy, idx, pred, None, None)
        return pred.view(-1, 9, self.C).permute(0, 2, 1) if self._loc else pred

    __call__ = forward

    # ---- loss ------------------------------------------------------------------------------------
    def _labels(self, y):
        pt = self.hparams.prediction_type
        idx = None
        if pt == "reachability":
            idx, y = y
            idx = torch.as_tensor(idx, dtype=torch.int64).to(self.device).contiguous()
        y = torch.as_tensor(y).to(self.device, torch.int64).contiguous()
        if pt == "object_localization":
            y = y.reshape(-1, self.C)                 # [B, 9, 52] -> rows (b, cell) == y.flatten(1) element order
        return y, idx

    @_lib.on_device
    def compute_loss(self, batch, eval: bool = False, _backward: bool = False):
        """train.py:56-92.  Returns the mean loss as a 0-d device tensor (and the metrics dict when ``eval``)."""
        x, y = batch
        labels, idx = self._labels(y)
        rows = self._rows(x)
        z = self._logits(rows)
        dz = torch.empty_like(z) if _backward else None
        self._head(z, labels, idx, None, dz, self._db if _backward else None)
        count = z.numel() if _MODE[self.hparams.prediction_type] == 0 else z.shape[0]
        loss = (self._out5[0] / count).to(torch.float32)
        if _backward:   # dW[c, k] = sum_r dz[r, c] rows[r, k]  (TN GEMM)
            R = z.shape[0]
            _lib.check(self.lib.ec_gemm_f32(dz.data_ptr(), rows.data_ptr(), self._dW.data_ptr(), self.C, self.K, R,
                                            1, self.C, self.K, 1, self.K, 0, None, None, None, 0, None, None, 1,
                                            _lib.stream_ptr()), "ec_gemm_f32(probe dW)")
        if not eval:
            return loss
        o = self._out5
        if self.hparams.prediction_type in ("object_presence", "object_localization"):
            acc = 2.0 * o[1] / torch.clamp(o[2] + o[3], min=1.0)         # micro-F1 at 0.5 (MF.f1, train.py:86)
        else:
            acc = o[4] / count                                            # train.py:88,90
        return loss, {"accuracy": acc.to(torch.float32)}

    # ---- LightningModule step surface -------------------------------------------------------------
    def configure_optimizers(self):
        from .ppo import FlatAdam
        if self._opt is None:    # torch.optim.Adam(self.parameters(), lr) (train.py:111-113); no grad clipping
            self._opt = FlatAdam(self.flat, lr=self.hparams.lr, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=0.0)
        return self._opt

    def log(self, name: str, value):
        self.logged[name] = value

    def training_step(self, batch, batch_idx: int = 0):
        """forward + loss + backward + Adam step (what the Lightning trainer does around train.py:94-97)."""
        loss = self.compute_loss(batch, _backward=True)
        self.configure_optimizers().step(self.grads)
        self.log("train_loss", loss)
        return loss

    def validation_step(self, batch, batch_idx: int = 0):
        loss, metrics = self.compute_loss(batch, eval=True)
        self.log("val_loss", loss)
        self.log("val_acc", metrics["accuracy"])
        return loss

    def test_step(self, batch, batch_idx: int = 0):
        loss, metrics = self.compute_loss(batch, eval=True)
        self.log("test_loss", loss)
        self.log("test_acc", metrics["accuracy"])
        return loss
