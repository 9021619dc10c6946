"""CPU: the preprocessing oracle is PINNED against the reference's own dependency.  ``clip_preprocess``
(thor_image_features.py:108) resizes with Pillow's antialiased bicubic; Pillow is installed here, so the numpy
restatement must reproduce ``Image.resize(..., BICUBIC)`` bit for bit, and the committed fixture (made by PIL:
tests/golden/make_preprocess_golden.py) pins it where PIL may be absent."""
import os

import numpy as np
import pytest
import torch

from embodied_clip_amd import synthetic as syn
from oracle import preprocess as opre

GOLD = os.path.join(os.path.dirname(__file__), "golden", "preprocess_golden.npz")


def _frames(seed, n, h, w):
    u = syn.hash_u64(seed, n * h * w * 3, stream=3)
    # smooth-ish content plus noise: exercises rounding in both passes and the clip8 saturation at 0 / 255
    noise = (u % np.uint64(256)).astype(np.int64).reshape(n, h, w, 3)
    yy, xx = np.mgrid[0:h, 0:w]
    base = (127 + 120 * np.sin(xx / 9.0 + seed) * np.cos(yy / 7.0))[None, :, :, None]
    img = np.where(noise > 200, 255, np.where(noise < 50, 0, base + (noise - 128) // 4))
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("h,w,oh,ow", [(300, 300, 224, 224), (400, 300, 298, 224), (300, 533, 224, 397),
                                       (200, 200, 224, 224), (224, 300, 224, 224), (37, 53, 11, 29)])
def test_restatement_equals_pillow_bit_for_bit(h, w, oh, ow):
    Image = pytest.importorskip("PIL.Image")
    for f in _frames(h + w, 2, h, w):
        ref = np.asarray(Image.fromarray(f).resize((ow, oh), Image.BICUBIC))
        got = opre.pil_bicubic_resize_u8(f, oh, ow)
        assert got.shape == ref.shape and np.array_equal(got, ref), int(np.abs(got.astype(int) - ref.astype(int)).max())


def test_clip_transform_geometry_and_values_equal_pillow_pipeline():
    """Resize(224) + CenterCrop(224) + ToTensor + Normalize exactly as torchvision 0.8.2 composes them on PIL images."""
    Image = pytest.importorskip("PIL.Image")
    for (h, w) in ((300, 300), (300, 400), (451, 300)):
        f = _frames(7, 1, h, w)[0]
        oh, ow, top, left = opre.resize_geometry(h, w)
        assert min(oh, ow) == 224 and (oh, ow) == ((224, int(224 * w / h)) if w >= h else (int(224 * h / w), 224))
        img = Image.fromarray(f).resize((ow, oh), Image.BICUBIC)
        img = img.crop((left, top, left + 224, top + 224))
        ref = (np.asarray(img).astype(np.float32) / np.float32(255) - np.asarray(opre.CLIP_MEAN, np.float32)) / \
            np.asarray(opre.CLIP_STD, np.float32)
        got = opre.clip_preprocess(f)
        assert got.shape == (3, 224, 224) and np.array_equal(got, ref.transpose(2, 0, 1))


def test_antialiased_float_bicubic_is_within_one_level():
    """Independent implementation: torch's antialiased bicubic (a = -0.5, float math, no uint8 intermediate)."""
    # smooth content away from 0 / 255: Pillow clips the uint8 intermediate after the horizontal pass, a float
    # implementation does not, so saturating overshoots are excluded from this cross-check
    yy, xx = np.mgrid[0:300, 0:300]
    f = np.stack([128 + 70 * np.sin(xx / (6.0 + c)) * np.cos(yy / (5.0 + 2 * c)) + 9 * np.sin(xx * yy / 997.0)
                  for c in range(3)], -1).round().astype(np.uint8)
    got = opre.pil_bicubic_resize_u8(f, 224, 224).astype(np.float32)
    t = torch.from_numpy(f).permute(2, 0, 1)[None].float()
    ref = torch.nn.functional.interpolate(t, size=(224, 224), mode="bicubic", antialias=True, align_corners=False)
    ref = ref.clamp(0, 255)[0].permute(1, 2, 0).numpy()
    assert np.abs(got - ref).max() <= 1.25 and np.abs(got - ref).mean() < 0.35    # two uint8 roundings vs none


def test_committed_pillow_fixture():
    g = np.load(GOLD)
    frames = _frames(int(g["seed"]), int(g["n"]), int(g["h"]), int(g["w"]))
    assert frames[:, ::17, ::13].tobytes() == g["input_slice"].tobytes()        # the generator has not drifted
    out = np.stack([opre.clip_resize_crop_u8(f) for f in frames])
    assert np.array_equal(out[:, ::4, ::4], g["resized_slice"])
    assert [int(x) for x in out.reshape(len(out), -1).astype(np.int64).sum(1)] == [int(x) for x in g["resized_sum"]]


def test_host_coefficient_tables_of_the_library_equal_the_oracle():
    """``ec_clip_resize_table`` is host-only arithmetic (no GPU call): torchvision geometry + Pillow's fixed-point
    tables, cropped to the 224 kept columns / rows, must equal the pinned oracle's for every frame geometry."""
    from embodied_clip_amd import _lib
    lib = _lib.load()
    for (H, W) in [(300, 300), (400, 300), (300, 533), (224, 300), (200, 260), (1080, 1920), (225, 224)]:
        n = lib.ec_clip_resize_table_ints(H, W, 224)
        t = torch.empty(n, dtype=torch.int32)
        _lib.check(lib.ec_clip_resize_table(H, W, 224, t.data_ptr(), n))
        t = t.numpy()
        oh, ow, top, left = opre.resize_geometry(H, W)
        bx, kx, ksx = opre.precompute_coeffs(W, ow)
        by, ky, ksy = opre.precompute_coeffs(H, oh)
        assert t[:6].tolist() == [oh, ow, top, left, ksx, ksy]
        o = 16
        assert np.array_equal(t[o:o + 448].reshape(224, 2), bx[left:left + 224]); o += 448
        assert np.array_equal(t[o:o + 224 * ksx].reshape(224, ksx), kx[left:left + 224]); o += 224 * ksx
        assert np.array_equal(t[o:o + 448].reshape(224, 2), by[top:top + 224]); o += 448
        assert np.array_equal(t[o:o + 224 * ksy].reshape(224, ksy), ky[top:top + 224])
    assert lib.ec_clip_resize_table_ints(100, 300, 224) > 0               # up-scaling the short edge is a valid geometry
    assert lib.ec_clip_resize_table_ints(0, 300, 224) == 0
