"""Oracle: CLIP's image preprocessing ``clip_preprocess`` on uint8 frames.

Restates what the reference runs per frame at
primitive_probing/generate_data/thor_image_features.py:108 (``clip_preprocess(frame)``; the ImageNet analogue is
spelled out at :36-44) on the 300x300 frames of thor_frames.py:33-34:

  [U] openai/CLIP clip/clip.py ``_transform(n_px)``: Resize(n_px, interpolation=BICUBIC) -> CenterCrop(n_px) ->
  ToTensor (/255, CHW) -> Normalize(CLIP mean/std)
  [U] torchvision 0.8.2 ``F.resize`` (smaller edge -> n_px, the other edge ``int(n_px * long / short)``) and
  ``F.center_crop`` (``int(round((H - th) / 2.))``), both on PIL images
  [U] Pillow ``Image.resize(size, BICUBIC)`` == libImaging/Resample.c ``ImagingResample``: separable, ANTIALIASED
  (filter support scaled by the down-scale factor), bicubic a = -0.5, coefficients normalised in double precision
  and quantised to 22-bit fixed point, horizontal pass first with a uint8 intermediate, 32-bit integer accumulation.

PARITY PINNED: Pillow IS installed in the build image (it is the reference's own dependency), so this restatement
is checked BIT-EXACTLY against ``PIL.Image.resize`` in tests/test_oracle_preprocess.py and the committed fixtures
under tests/golden/ were produced by PIL itself (tests/golden/make_preprocess_golden.py).
Pure numpy integer / float64 arithmetic in PIL's operation order.  TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def bicubic_filter(x: float) -> float:
    """Resample.c ``bicubic_filter`` (a = -0.5, support 2)."""
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """Resample.c ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` for the full-image box.
    -> (bounds int32 [out, 2] = (xmin, count), kk int32 [out, ksize], ksize)."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [bicubic_filter((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        if ww != 0.0:
            k = [w / ww for w in k]
        for x, w in enumerate(k):
            kk[xx, x] = int(-0.5 + w * (1 << PRECISION_BITS)) if w < 0 else int(0.5 + w * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _resample_axis1(img: np.ndarray, out_size: int) -> np.ndarray:
    """One 8-bit pass along axis 1 of [R, S, C] uint8 (``ImagingResampleHorizontal_8bpc`` / ``Vertical``)."""
    bounds, kk, _ = precompute_coeffs(img.shape[1], out_size)
    src = img.astype(np.int64)
    out = np.empty((img.shape[0], out_size, img.shape[2]), dtype=np.uint8)
    for xx in range(out_size):
        xmin, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(src[:, xmin:xmin + n, :], kk[xx, :n].astype(np.int64), axes=([1], [0]))
        out[:, xx, :] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)      # clip8
    return out


def pil_bicubic_resize_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """``Image.fromarray(img).resize((out_w, out_h), Image.BICUBIC)`` for uint8 [H, W, C]: horizontal pass, then
    vertical pass on the uint8 intermediate; a pass whose size does not change is skipped (as ImagingResample does)."""
    assert img.dtype == np.uint8 and img.ndim == 3
    x = img
    if out_w != x.shape[1]:
        x = _resample_axis1(x, out_w)
    if out_h != x.shape[0]:
        x = _resample_axis1(x.transpose(1, 0, 2), out_h).transpose(1, 0, 2)
    return np.ascontiguousarray(x)


def resize_geometry(h: int, w: int, n_px: int = 224) -> Tuple[int, int, int, int]:
    """torchvision 0.8.2 ``F.resize(img, n_px)`` + ``F.center_crop(img, n_px)`` -> (oh, ow, crop_top, crop_left)."""
    if (w <= h and w == n_px) or (h <= w and h == n_px):
        oh, ow = h, w
    elif w < h:
        ow, oh = n_px, int(n_px * h / w)
    else:
        oh, ow = n_px, int(n_px * w / h)
    return oh, ow, int(round((oh - n_px) / 2.0)), int(round((ow - n_px) / 2.0))


def clip_resize_crop_u8(frame: np.ndarray, n_px: int = 224) -> np.ndarray:
    """Resize(n_px, BICUBIC) + CenterCrop(n_px) on a uint8 RGB frame [H, W, 3] -> uint8 [n_px, n_px, 3]."""
    oh, ow, top, left = resize_geometry(frame.shape[0], frame.shape[1], n_px)
    r = pil_bicubic_resize_u8(frame, oh, ow)
    return np.ascontiguousarray(r[top:top + n_px, left:left + n_px])


def clip_preprocess(frame: np.ndarray, n_px: int = 224) -> np.ndarray:
    """The whole ``clip_preprocess``: uint8 [H, W, 3] -> float32 [3, n_px, n_px] normalised (ToTensor + Normalize)."""
    x = clip_resize_crop_u8(frame, n_px).astype(np.float32) / np.float32(255.0)
    x = (x - np.asarray(CLIP_MEAN, np.float32)) / np.asarray(CLIP_STD, np.float32)
    return np.ascontiguousarray(x.transpose(2, 0, 1))
