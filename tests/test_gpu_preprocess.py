"""GPU: CLIP's Resize(224, BICUBIC) + CenterCrop(224) on raw uint8 frames (``ec_clip_resize_crop_u8``) is BIT-EXACT
with Pillow -- checked against the committed Pillow-made fixture and against the Pillow-pinned numpy oracle -- and the
fused pipeline uint8 300x300 -> resize -> crop -> normalise -> stem -> trunk matches the oracle
(``clip_preprocess`` + trunk; primitive_probing/generate_data/thor_image_features.py:108-109)."""
import os

import numpy as np
import pytest
import torch

from embodied_clip_amd import synthetic as syn
from oracle import clip_resnet as ocr
from oracle import preprocess as opre
from test_oracle_preprocess import GOLD, _frames

pytestmark = pytest.mark.gpu


def test_resize_crop_is_bit_exact_with_the_pillow_fixture():
    from embodied_clip_amd.encoder import ClipResizeCrop
    g = np.load(GOLD)
    frames = _frames(int(g["seed"]), int(g["n"]), int(g["h"]), int(g["w"]))
    out = ClipResizeCrop("cuda:0")(torch.from_numpy(frames).cuda()).cpu().numpy()
    assert out.shape == (3, 224, 224, 3) and out.dtype == np.uint8
    assert np.array_equal(out[:, ::4, ::4], g["resized_slice"])
    assert [int(x) for x in out.reshape(3, -1).astype(np.int64).sum(1)] == [int(x) for x in g["resized_sum"]]


@pytest.mark.parametrize("h,w", [(300, 300), (400, 300), (300, 533), (224, 300), (260, 225), (720, 1280)])
def test_resize_crop_matches_the_pinned_oracle_everywhere(h, w):
    from embodied_clip_amd.encoder import ClipResizeCrop
    frames = _frames(h * 3 + w, 2, h, w)
    rz = ClipResizeCrop("cuda:0")
    out = rz(torch.from_numpy(frames).cuda()).cpu().numpy()
    ref = np.stack([opre.clip_resize_crop_u8(f) for f in frames])
    assert np.array_equal(out, ref), int(np.abs(out.astype(int) - ref.astype(int)).max())
    again = rz(torch.from_numpy(frames[::-1].copy()).cuda()).cpu().numpy()          # cached table, second launch
    assert np.array_equal(again, ref[::-1])


def test_fused_input_pipeline_300x300_uint8_to_features():
    """f2: raw 300x300 uint8 frame -> (resize, crop) -> (/255, mean/std fused into the stem) -> trunk."""
    from embodied_clip_amd.clip_preprocessors import ClipResNetPreprocessor
    sd = syn.rn50_visual_state_dict(0)
    frames = _frames(5, 2, 300, 300)
    pre = ClipResNetPreprocessor("rgb", "RN50", pool=False, device="cuda:0", state_dict=sd)
    out = pre.process({"rgb": torch.from_numpy(frames)}).cpu()
    x = torch.from_numpy(np.stack([opre.clip_preprocess(f) for f in frames]))        # [2, 3, 224, 224] normalised
    ref = ocr.rn50_trunk(x, sd)
    rel = ((out - ref).norm() / ref.norm()).item()
    assert out.shape == (2, 2048, 7, 7) and rel < 2e-2, rel
