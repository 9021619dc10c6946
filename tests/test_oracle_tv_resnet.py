"""Pins for the torchvision-ResNet-50 oracle (the ImageNet branch of the feature scripts,
primitive_probing/generate_data/thor_image_features.py:36-54,102-106).

Unlike the CLIP ModifiedResNet, this network HAS an independent implementation installed in the image: HuggingFace
``transformers.ResNetModel`` (ResNet v1.5: stride on the 3x3 conv, 1x1-conv/BN shortcut, 7x7-s2 conv + 3x3-s2 max-pool
embedder).  The restatement is checked against it end to end on shared random weights."""
import pytest
import torch

from embodied_clip_amd import synthetic as syn
from oracle import tv_resnet as otv
from oracle import clip_resnet as ocr


def _randomise(model, g):
    with torch.no_grad():
        for mod in model.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.copy_(torch.rand(mod.weight.shape, generator=g) + 0.5)
                mod.bias.copy_(torch.randn(mod.bias.shape, generator=g) * 0.1)
                mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=g) * 0.2)
                mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) + 0.5)
            elif isinstance(mod, torch.nn.Conv2d):
                mod.weight.copy_(torch.randn(mod.weight.shape, generator=g) * (mod.weight[0].numel() ** -0.5))


@pytest.mark.parametrize("depths,width,res", [((1, 1, 1, 1), 16, 64), ((2, 2, 2, 1), 8, 96)])
def test_trunk_matches_hf_resnet_model(depths, width, res):
    """whole network vs transformers.ResNetModel (last_hidden_state == children()[:-2] output; pooler_output ==
    AdaptiveAvgPool2d(1)), folded and unfolded BatchNorm."""
    tr = pytest.importorskip("transformers")
    cfg = tr.ResNetConfig(num_channels=3, embedding_size=width, hidden_sizes=[width * 4, width * 8, width * 16, width * 32],
                          depths=list(depths), layer_type="bottleneck", hidden_act="relu", downsample_in_first_stage=False,
                          downsample_in_bottleneck=False)
    model = tr.ResNetModel(cfg).eval()
    g = torch.Generator().manual_seed(11)
    _randomise(model, g)
    sd = otv.hf_resnet_to_torchvision_keys(model.state_dict())
    assert sd["conv1.weight"].shape == (width, 3, 7, 7) and "layer2.0.downsample.0.weight" in sd
    x = torch.randn(2, 3, res, res, generator=g)
    with torch.no_grad():
        out = model(x.clone())
    for fold in (True, False):
        got = otv.tv_resnet_trunk(x, sd, fold=fold)
        assert got.shape == out.last_hidden_state.shape == (2, width * 32, res // 32, res // 32)
        assert (got - out.last_hidden_state).abs().max() < 5e-5 * max(1.0, float(out.last_hidden_state.abs().max()))
    conv, avg = otv.imagenet_features(x.permute(0, 2, 3, 1).contiguous(), sd)
    assert torch.allclose(avg, out.pooler_output.flatten(1), rtol=1e-4, atol=1e-5)


def test_full_resnet50_geometry_and_param_count():
    sd = syn.tv_resnet_state_dict(0)
    # torchvision resnet50: 25,557,032 parameters with fc (1000 x 2048 + 1000 = 2,049,000)
    assert ocr.param_count(sd) == 25_557_032 - 2_049_000
    assert ocr.param_count(syn.tv_resnet_state_dict(0, with_fc=True)) == 25_557_032
    assert sd["conv1.weight"].shape == (64, 3, 7, 7) and sd["layer4.2.conv3.weight"].shape == (2048, 512, 1, 1)
    assert "layer1.0.downsample.0.weight" in sd and "layer1.1.downsample.0.weight" not in sd


def test_bf16_emulation_close_to_fp32_and_portable():
    sd = syn.tv_resnet_state_dict(3, width=16, layers=(1, 1, 1, 1))
    sd2 = syn.tv_resnet_state_dict(3, width=16, layers=(1, 1, 1, 1))
    assert all(torch.equal(sd[k], sd2[k]) for k in sd)
    x = syn.normalize_rgb_imagenet(syn.synthetic_rgb_u8(4, 2, 64)).permute(0, 3, 1, 2)
    a = otv.tv_resnet_trunk(x, sd)
    b = otv.tv_resnet_trunk(x, sd, emulate_bf16=True)
    assert a.shape == (2, 512, 2, 2)
    assert torch.nn.functional.cosine_similarity(a.flatten(1), b.flatten(1)).min() > 0.999
