"""GPU parity of the ImageNet (torchvision ResNet-50) branch of the feature scripts -- SURVEY.md 8f-4,
primitive_probing/generate_data/thor_image_features.py:36-54,102-106 -- against ``oracle/tv_resnet.py``, which is itself
pinned against HuggingFace ``ResNetModel`` (tests/test_oracle_tv_resnet.py).

Tolerances as for the CLIP trunk: vs the oracle's bf16-rounding emulation (same rounding points) rel-L2 <= 4e-3 * sqrt(1 +
#blocks); vs the fp32 oracle rel-L2 <= 2e-2 and cosine >= 0.999."""
import math

import pytest
import torch
import torch.nn.functional as F

from embodied_clip_amd import synthetic as syn
from oracle import tv_resnet as otv

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item()


def _bf(x):
    return x.to(torch.bfloat16)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")


@pytest.mark.parametrize("B,H,W,Cin,Cout,ks,res,act", [
    (2, 56, 56, 128, 128, 3, False, 1),     # layer2.0 conv2: 128x128 tiles
    (3, 28, 28, 256, 256, 3, False, 1),     # layer3.0 conv2: few tiles -> 64x64 ring tiles, borders on every side
    (2, 14, 14, 512, 512, 3, False, 1),     # layer4.0 conv2
    (2, 56, 56, 256, 512, 1, False, 0),     # layer2.0 downsample (no activation)
    (5, 14, 14, 1024, 2048, 1, False, 0),   # layer4.0 downsample
    (40, 28, 28, 64, 128, 3, False, 1),     # many tiles (persistent workgroups walk several), K = 576
    (3, 12, 20, 32, 64, 3, True, 1),        # Cout = 64 (256x64 tiles), residual, non-square, ragged last tile
    (2, 6, 10, 64, 192, 1, True, 0),        # Cout = 192 -> 64-wide tiles
    # the 8-wave ping-pong kernel's stride-2 instances (>= 150 tiles: what a >= 128-frame launch of the tower takes)
    (200, 28, 28, 128, 128, 3, False, 1),   # 128-wide tiles, 3x3 (layer2.0 conv2 geometry at 28x28: 154 tiles, ragged last tile)
    (200, 28, 28, 256, 256, 3, False, 1),   # 256-wide tiles, 3x3
    (200, 14, 14, 512, 1024, 1, False, 0),  # 256-wide tiles, 1x1 downsample conv, no activation
    (160, 14, 14, 1024, 2048, 1, True, 1),  # 256-wide tiles, 1x1 with a residual
    (128, 14, 14, 512, 512, 3, False, 1),   # layer4.0 conv2 at 128 frames: 196 tiles -> 128 x 128 ring tiles on 8 waves (3 stages)
    (40, 14, 14, 1024, 512, 1, False, 0),   # 1x1, 64 tiles < 150 -> 64 x 64 ring tiles
])
def test_stride2_conv_matches_torch(dev, B, H, W, Cin, Cout, ks, res, act):
    from embodied_clip_amd import encoder as enc
    g = torch.Generator().manual_seed(B * 1000 + Cin + Cout + ks)
    x = _bf(torch.randn(B, H, W, Cin, generator=g))
    w = _bf(torch.randn(Cout, ks, ks, Cin, generator=g) * (ks * ks * Cin) ** -0.5)
    b = torch.randn(Cout, generator=g) * 0.1
    r = _bf(torch.randn(B, H // 2, W // 2, Cout, generator=g)) if res else None
    y = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b, stride=2, padding=ks // 2)
    if r is not None:
        y = y + r.float().permute(0, 3, 1, 2)
    if act:
        y = F.relu(y)
    y = y.permute(0, 2, 3, 1)
    got = enc.conv_bf16_s2(x.to(dev), w.reshape(Cout, -1).to(dev), b.to(dev), None if r is None else r.to(dev), ksize=ks, act=act)
    torch.cuda.synchronize()
    got = got.cpu().float()
    assert got.shape == y.shape
    assert _rel(got, y) < 4e-3, _rel(got, y)
    assert (got - y).abs().max() <= 2e-2 * y.abs().max() + 1e-3


@pytest.mark.parametrize("B,R", [(2, 224), (3, 64), (1, 96), (2, 40)])
@pytest.mark.parametrize("u8", [False, True])
def test_stem7_pool_matches_oracle(dev, B, R, u8):
    """conv1 7x7 s2 + bn1 + relu + maxpool 3x3 s2 in one launch vs F.conv2d / F.max_pool2d on the same bf16-rounded operands
    (tile seams, ragged tiles at 64 / 96 / 40, every frame border)."""
    from embodied_clip_amd import encoder as enc
    sd = syn.tv_resnet_state_dict(5, layers=(1, 1, 1, 1))
    (_w, _l), stem_w, _wf, bias = enc.pack_tv_resnet(sd)
    raw = syn.synthetic_rgb_u8(31 + R, B, R)
    x = syn.normalize_rgb_imagenet(raw)
    wf = stem_w.float()[:, :168].reshape(64, 7, 24)[:, :, :21].reshape(64, 7, 7, 3).permute(0, 3, 1, 2)   # back to [64,3,7,7]
    conv = F.relu(F.conv2d(_bf(x).float().permute(0, 3, 1, 2), wf, bias[:64], stride=2, padding=3))
    ref = F.max_pool2d(_bf(conv).float(), 3, 2, 1).permute(0, 2, 3, 1)
    if u8:
        got = enc.stem7_pool(raw.to(dev), stem_w.to(dev), bias[:64].contiguous().to(dev), mean=syn.IMAGENET_RGB_MEANS,
                             std=syn.IMAGENET_RGB_STDS)
    else:
        got = enc.stem7_pool(x.contiguous().to(dev), stem_w.to(dev), bias[:64].contiguous().to(dev))
    torch.cuda.synchronize()
    got = got.cpu().float()
    assert got.shape == ref.shape == (B, R // 4, R // 4, 64)
    assert _rel(got, ref) < 4e-3, _rel(got, ref)
    assert (got - ref).abs().max() <= 2e-2 * ref.abs().max() + 1e-3


@pytest.mark.parametrize("layers,R,B", [((1, 1, 1, 1), 64, 3), ((2, 2, 2, 2), 96, 2), ((3, 4, 6, 3), 224, 2)])
def test_tv_resnet_trunk_matches_oracle(dev, layers, R, B):
    from embodied_clip_amd.encoder import ImageNetRN50Trunk
    sd = syn.tv_resnet_state_dict(7, layers=layers)
    x = syn.normalize_rgb_imagenet(syn.synthetic_rgb_u8(17, B, R))
    trunk = ImageNetRN50Trunk(sd, device=dev, input_resolution=R)
    assert trunk.out_channels == 2048 and trunk.out_spatial == R // 32
    feat = trunk.forward(x.contiguous().to(dev))
    nchw = trunk.to_nchw_f32(feat).cpu()
    avg = trunk.spatial_mean(feat).cpu()
    ref = otv.tv_resnet_trunk(x.permute(0, 3, 1, 2), sd)
    emu = otv.tv_resnet_trunk(x.permute(0, 3, 1, 2), sd, emulate_bf16=True)
    assert nchw.shape == ref.shape == (B, 2048, R // 32, R // 32)
    nb = sum(layers)
    assert _rel(nchw, emu) < 4e-3 * math.sqrt(1 + nb), _rel(nchw, emu)
    assert _rel(nchw, ref) < 2e-2, _rel(nchw, ref)
    assert F.cosine_similarity(nchw.flatten(1), ref.flatten(1)).min() > 0.999
    # imagenet_avgpool (thor_image_features.py:51-54,106) of the stored features
    assert torch.allclose(avg, nchw.mean(dim=(2, 3)), rtol=1e-5, atol=1e-6)
    assert _rel(avg, F.adaptive_avg_pool2d(ref, 1).flatten(1)) < 2e-2


def test_tv_resnet_u8_path_and_large_launch(dev):
    """raw 300x300 uint8 frames (thor_frames.py:33-34) -> Pillow-exact resize/crop -> fused ImageNet normalisation -> trunk ==
    the fp32 path on the same resized frames; and a 130-frame launch (fused whole-bottleneck launches in layer 3) gives the
    frames the features a 2-frame launch gives them (up to the across-launch-shape bound of DESIGN.md section 2)."""
    from embodied_clip_amd.encoder import ClipResizeCrop, ImageNetRN50Trunk
    sd = syn.tv_resnet_state_dict(9)
    trunk = ImageNetRN50Trunk(sd, device=dev)
    raw = syn.synthetic_rgb_u8(3, 2, 300).to(dev)
    f_u8 = trunk.forward_u8(raw).float().cpu()
    resized = ClipResizeCrop(dev)(raw)
    f_f32 = trunk.forward(syn.normalize_rgb_imagenet(resized).contiguous()).float().cpu()
    assert _rel(f_u8, f_f32) < 7e-3, _rel(f_u8, f_f32)
    frames = syn.normalize_rgb_imagenet(syn.synthetic_rgb_u8(4, 5, 224)).to(dev)
    big = frames.repeat(26, 1, 1, 1).contiguous()           # 130 frames
    fb = trunk.forward(big).float().cpu()
    fs = trunk.forward(frames[:2].contiguous()).float().cpu()
    assert torch.equal(fb[:5], fb[125:130])                  # same frame, same launch -> bit-identical
    assert _rel(fb[:2], fs) < 7e-3, _rel(fb[:2], fs)
    ref = otv.tv_resnet_trunk(frames[:1].cpu().permute(0, 3, 1, 2), sd).permute(0, 2, 3, 1)
    assert _rel(fb[:1], ref) < 2e-2
