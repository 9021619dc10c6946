"""Oracle pins for the CLIP-RN50 tower (no reference tests exist: SURVEY.md §4)."""
import torch
import torch.nn as nn

from embodied_clip_amd import synthetic as syn
from oracle import clip_resnet as ocr


def test_rn50_param_checksum():
    sd = syn.rn50_visual_state_dict(0)
    assert ocr.param_count(sd) == 38_316_896
    trunk = {k: v for k, v in sd.items() if not k.startswith("attnpool.")}
    assert ocr.param_count(trunk) == 23_527_264
    assert ocr.param_count(sd) - ocr.param_count(trunk) == 14_789_632


def test_synthetic_is_portable():
    a = syn.rn50_visual_state_dict(3, width=8, layers=(1, 1, 1, 1), output_dim=16, heads=4, input_resolution=64)
    b = syn.rn50_visual_state_dict(3, width=8, layers=(1, 1, 1, 1), output_dim=16, heads=4, input_resolution=64)
    for k in a:
        assert torch.equal(a[k], b[k])
    # known-answer values of the hash generator: a change here breaks every golden fixture
    u = syn.hash_uniform(1, 4)
    assert abs(u[0] - 0.16737875674524771) < 1e-15 and abs(u[3] - 0.5307842665235228) < 1e-15
    assert syn.synthetic_rgb_u8(5, 1, 8).sum().item() == 23637


def test_bn_fold_equals_eval_batchnorm():
    """freeze_model contract (thor_image_features.py:26-33): eval BN == folded affine."""
    sd = syn.rn50_visual_state_dict(1, width=16, layers=(1, 1, 1, 1), output_dim=32, heads=4, input_resolution=64)
    x = syn.synthetic_rgb(2, 2, 64).permute(0, 3, 1, 2)
    a = ocr.rn50_trunk(x, sd, fold=True)
    b = ocr.rn50_trunk(x, sd, fold=False)
    assert a.shape == (2, 512, 2, 2)
    assert torch.allclose(a, b, rtol=1e-4, atol=1e-4)


def test_trunk_matches_nn_module_composition():
    """Independent composition from nn.Modules (Conv2d/BatchNorm2d/AvgPool2d) loaded
    with the same state dict, structured like CLIP's ModifiedResNet."""
    w, layers = 16, (2, 1, 1, 1)
    sd = syn.rn50_visual_state_dict(4, width=w, layers=layers, output_dim=32, heads=4, input_resolution=64)

    class Bott(nn.Module):
        def __init__(s, inp, planes, stride):
            super().__init__()
            s.conv1 = nn.Conv2d(inp, planes, 1, bias=False); s.bn1 = nn.BatchNorm2d(planes)
            s.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False); s.bn2 = nn.BatchNorm2d(planes)
            s.avgpool = nn.AvgPool2d(stride) if stride > 1 else nn.Identity()
            s.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False); s.bn3 = nn.BatchNorm2d(planes * 4)
            s.relu = nn.ReLU()
            s.downsample = None
            if stride > 1 or inp != planes * 4:
                from collections import OrderedDict
                s.downsample = nn.Sequential(OrderedDict([("-1", nn.AvgPool2d(stride)),
                                                          ("0", nn.Conv2d(inp, planes * 4, 1, bias=False)),
                                                          ("1", nn.BatchNorm2d(planes * 4))]))

        def forward(s, x):
            idt = x
            o = s.relu(s.bn1(s.conv1(x))); o = s.relu(s.bn2(s.conv2(o))); o = s.avgpool(o); o = s.bn3(s.conv3(o))
            if s.downsample is not None:
                idt = s.downsample(x)
            return s.relu(o + idt)

    class Net(nn.Module):
        def __init__(s):
            super().__init__()
            s.conv1 = nn.Conv2d(3, w // 2, 3, 2, 1, bias=False); s.bn1 = nn.BatchNorm2d(w // 2)
            s.conv2 = nn.Conv2d(w // 2, w // 2, 3, padding=1, bias=False); s.bn2 = nn.BatchNorm2d(w // 2)
            s.conv3 = nn.Conv2d(w // 2, w, 3, padding=1, bias=False); s.bn3 = nn.BatchNorm2d(w)
            s.avgpool = nn.AvgPool2d(2); s.relu = nn.ReLU()
            inp = w
            for li, (n, m) in enumerate(zip(layers, (1, 2, 4, 8)), 1):
                blocks = []
                for b in range(n):
                    blocks.append(Bott(inp, w * m, 2 if (b == 0 and li > 1) else 1)); inp = w * m * 4
                setattr(s, f"layer{li}", nn.Sequential(*blocks))

        def forward(s, x):
            for c, b in ((s.conv1, s.bn1), (s.conv2, s.bn2), (s.conv3, s.bn3)):
                x = s.relu(b(c(x)))
            x = s.avgpool(x)
            return s.layer4(s.layer3(s.layer2(s.layer1(x))))

    net = Net()
    net.load_state_dict({k: v for k, v in sd.items() if not k.startswith("attnpool.")})
    net.eval()
    x = syn.synthetic_rgb(9, 2, 64).permute(0, 3, 1, 2)
    with torch.no_grad():
        ref = net(x)
    got = ocr.rn50_trunk(x, sd)
    assert torch.allclose(got, ref, rtol=1e-4, atol=1e-4)


def test_attnpool_matches_manual_cls_query():
    """AttentionPool2d returns token 0 only, so a CLS-only query gives the same result."""
    sd = syn.rn50_visual_state_dict(2, width=16, layers=(1, 1, 1, 1), output_dim=32, heads=4, input_resolution=64)
    f = torch.randn(3, 512, 2, 2, generator=torch.Generator().manual_seed(0))
    ref = ocr.attnpool(f, sd, num_heads=4)
    B, C = 3, 512
    x = f.reshape(B, C, 4).permute(2, 0, 1)
    x = torch.cat([x.mean(0, keepdim=True), x], 0) + sd["attnpool.positional_embedding"][:, None, :]
    q = (x[:1] @ sd["attnpool.q_proj.weight"].T + sd["attnpool.q_proj.bias"]) * (C // 4) ** -0.5
    k = x @ sd["attnpool.k_proj.weight"].T + sd["attnpool.k_proj.bias"]
    v = x @ sd["attnpool.v_proj.weight"].T + sd["attnpool.v_proj.bias"]
    q = q.view(1, B, 4, C // 4); k = k.view(5, B, 4, C // 4); v = v.view(5, B, 4, C // 4)
    s = torch.einsum("qbhd,kbhd->bhqk", q, k).softmax(-1)
    o = torch.einsum("bhqk,kbhd->qbhd", s, v).reshape(1, B, C)
    out = o[0] @ sd["attnpool.c_proj.weight"].T + sd["attnpool.c_proj.bias"]
    assert ref.shape == (3, 32)
    assert torch.allclose(ref, out, rtol=1e-4, atol=1e-5)


def test_emulated_bf16_close_to_fp32():
    sd = syn.rn50_visual_state_dict(1, width=16, layers=(1, 1, 1, 1), output_dim=32, heads=4, input_resolution=64)
    x = syn.synthetic_rgb(2, 2, 64).permute(0, 3, 1, 2)
    a = ocr.rn50_trunk(x, sd)
    b = ocr.rn50_trunk(x, sd, emulate_bf16=True)
    cos = torch.nn.functional.cosine_similarity(a.flatten(1), b.flatten(1)).min()
    assert cos > 0.999


def test_bottleneck_matches_hf_resnet_bottleneck_layer():
    """Independent pin for the bottleneck arithmetic (row a5): HuggingFace transformers' ResNetBottleNeckLayer is a
    third-party implementation of conv1x1-BN-ReLU -> conv3x3-BN-ReLU -> conv1x1-BN (+ 1x1-conv/BN shortcut) -> add ->
    ReLU with eval-mode BatchNorm.  For stride 1 that is exactly CLIP's Bottleneck (whose AvgPool2d(stride) is the
    identity at stride 1), with and without the projection shortcut; the oracle's folded and unfolded forms must both
    reproduce it.  (The anti-aliased stride-2 blocks and the 3-conv stem have no third-party counterpart installed.)"""
    pytest = __import__("pytest")
    m = pytest.importorskip("transformers.models.resnet.modeling_resnet")
    import torch
    from oracle import clip_resnet as R
    g = torch.Generator().manual_seed(5)
    for cin, cout in ((64, 64), (32, 64)):
        layer = m.ResNetBottleNeckLayer(cin, cout, stride=1).eval()
        with torch.no_grad():
            for mod in layer.modules():
                if isinstance(mod, torch.nn.BatchNorm2d):
                    mod.weight.copy_(torch.rand(mod.weight.shape, generator=g) + 0.5)
                    mod.bias.copy_(torch.randn(mod.bias.shape, generator=g) * 0.1)
                    mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=g) * 0.2)
                    mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) + 0.5)
                elif isinstance(mod, torch.nn.Conv2d):
                    mod.weight.copy_(torch.randn(mod.weight.shape, generator=g) * (mod.weight[0].numel() ** -0.5))
        sd = {}
        for i, name in enumerate(("1", "2", "3")):
            sd[f"blk.conv{name}.weight"] = layer.layer[i].convolution.weight.detach()
            for k in ("weight", "bias", "running_mean", "running_var"):
                sd[f"blk.bn{name}.{k}"] = getattr(layer.layer[i].normalization, k).detach()
        if cin != cout:
            sd["blk.downsample.0.weight"] = layer.shortcut.convolution.weight.detach()
            for k in ("weight", "bias", "running_mean", "running_var"):
                sd[f"blk.downsample.1.{k}"] = getattr(layer.shortcut.normalization, k).detach()
        x = torch.randn(2, cin, 9, 11, generator=g)
        with torch.no_grad():
            ref = layer(x.clone())
        for fold in (True, False):
            out = R.bottleneck(x, sd, "blk", 1, emulate=False, fold=fold)
            assert (out - ref).abs().max() < 2e-5, (cin, cout, fold, float((out - ref).abs().max()))


def test_stride2_bottleneck_equals_one_gemm_over_the_concatenated_k_axis():
    """What the trunk plan does for layer3.0 / layer4.0 (EC_RN50_DSCAT): with the folded weights,
    relu(bn3(conv3(avgpool(c2))) + bn_d(conv_d(avgpool(x)))) == relu([avgpool(c2) | avgpool(x)] . [W3 | Wd]^T + (b3 + bd)) -- the
    oracle's Bottleneck (the restatement of [U] clip/model.py Bottleneck.forward) against that single GEMM, and against the
    2 x 2 stride-2 convolution form of the anti-aliased stride (AvgPool2d(2) then a 1 x 1 conv == a 2 x 2 s2 conv with W / 4 taps)."""
    import torch.nn.functional as F
    sd = syn.rn50_visual_state_dict(4, width=16, layers=(1, 1, 1, 1), output_dim=32, heads=4, input_resolution=64)
    p = "layer3.0"
    g = torch.Generator().manual_seed(3)
    cin = sd[p + ".conv1.weight"].shape[1]
    x = torch.randn(2, cin, 8, 8, generator=g).relu()
    ref = ocr.bottleneck(x, sd, p, stride=2)
    w1, b1 = ocr.fold_bn(sd[p + ".conv1.weight"], sd, p + ".bn1")
    w2, b2 = ocr.fold_bn(sd[p + ".conv2.weight"], sd, p + ".bn2")
    w3, b3 = ocr.fold_bn(sd[p + ".conv3.weight"], sd, p + ".bn3")
    wd, bd = ocr.fold_bn(sd[p + ".downsample.0.weight"], sd, p + ".downsample.1")
    c2 = F.relu(F.conv2d(F.relu(F.conv2d(x, w1, b1)), w2, b2, padding=1))
    cat = torch.cat([F.avg_pool2d(c2, 2), F.avg_pool2d(x, 2)], 1)
    one = F.relu(F.conv2d(cat, torch.cat([w3, wd], 1), b3 + bd))
    assert one.shape == ref.shape and torch.allclose(one, ref, rtol=1e-4, atol=1e-5)
    two = F.relu(F.conv2d(c2, (w3 / 4).repeat(1, 1, 2, 2), b3, stride=2) + F.conv2d(x, (wd / 4).repeat(1, 1, 2, 2), bd, stride=2))
    assert torch.allclose(two, ref, rtol=1e-4, atol=1e-5)
