"""Generates tests/golden/tvresnet_golden.pt (run from the repo root: python tests/golden/make_tvresnet_golden.py).

The ImageNet tower of the feature scripts (torchvision ResNet-50 minus avgpool / fc: primitive_probing/generate_data/
thor_image_features.py:46-54,102-106) on portable hash-seeded weights and frames, computed by ``oracle/tv_resnet.py`` -- the
restatement that tests/test_oracle_tv_resnet.py pins against HuggingFace ``ResNetModel``.  Only seeds and expected outputs are
stored: a strided slice of ``imagenet_conv``, the full ``imagenet_avgpool`` and per-frame norms."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from embodied_clip_amd import synthetic as syn  # noqa: E402
from oracle import tv_resnet as otv  # noqa: E402

SEED_W, SEED_RGB, N = 13, 2025, 2


def compute():
    sd = syn.tv_resnet_state_dict(SEED_W)
    x = syn.normalize_rgb_imagenet(syn.synthetic_rgb_u8(SEED_RGB, N))
    conv, avg = otv.imagenet_features(x, sd)
    return {"seed_weights": SEED_W, "seed_rgb": SEED_RGB, "n": N, "conv_slice": conv[:, ::32].clone(), "avgpool": avg.clone(),
            "norm": conv.flatten(1).norm(dim=1)}


if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tvresnet_golden.pt")
    torch.save(compute(), out)
    print("wrote", out, os.path.getsize(out), "bytes")
