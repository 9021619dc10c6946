"""GPU parity: HIP encoder kernels (through the C-ABI) vs the CPU oracle.

Tolerances (floating point, bf16 storage + fp32 MFMA accumulate):
  * vs the oracle's bf16-emulation mode (same rounding points): rel-L2 <= 4e-3
    -- the tight check that catches indexing / layout bugs;
  * vs the pure fp32 oracle: rel-L2 <= 2e-2 and cosine >= 0.999.
"""
import pytest
import torch
import torch.nn.functional as F

from embodied_clip_amd import synthetic as syn
from oracle import clip_resnet as ocr

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item()


def _bf(x):
    return x.to(torch.bfloat16)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")


def _conv_case(dev, B, H, W, Cin, Cout, ks, pool, res, act, seed):
    from embodied_clip_amd import encoder as enc
    g = torch.Generator().manual_seed(seed)
    x = _bf(torch.randn(B, H, W, Cin, generator=g))
    w = _bf(torch.randn(Cout, ks, ks, Cin, generator=g) * (ks * ks * Cin) ** -0.5)
    b = torch.randn(Cout, generator=g) * 0.1
    r = _bf(torch.randn(B, H, W, Cout, generator=g)) if res else None
    # oracle (fp32 math on the bf16-rounded operands; asymmetric random data => transposes are caught)
    y = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b, padding=ks // 2)
    if r is not None:
        y = y + r.float().permute(0, 3, 1, 2)
    if act == 1:
        y = F.relu(y)
    elif act == 2:
        y = y * torch.sigmoid(1.702 * y)
    if pool:
        y = F.avg_pool2d(y, 2)
    y = y.permute(0, 2, 3, 1)
    out = enc.conv_bf16(x.to(dev), w.reshape(Cout, -1).to(dev), b.to(dev), None if r is None else r.to(dev),
                        ksize=ks, pool=pool, act=act)
    torch.cuda.synchronize()
    return out.cpu().float(), y


@pytest.mark.parametrize("B,H,W,Cin,Cout,ks,pool,res,act", [
    (2, 8, 8, 64, 128, 1, False, False, 1),      # 1x1, 128x128 tile
    (1, 7, 7, 512, 2048, 1, False, True, 1),     # layer4 conv3 + residual, M=49 tail
    (2, 12, 12, 32, 32, 3, False, False, 1),     # stem conv2: Cin=32 (tap split inside a K tile), BN=32 tile
    (2, 12, 12, 32, 64, 3, True, False, 1),      # stem conv3 + fused avgpool, BN=64 tile
    (3, 14, 14, 128, 128, 3, True, False, 1),    # bottleneck conv2 + fused avgpool
    (2, 14, 14, 256, 256, 3, False, False, 1),   # 3x3 K=2304
    (5, 7, 7, 64, 256, 1, False, True, 0),       # residual, no activation
    (1, 10, 6, 128, 384, 1, False, False, 2),    # QuickGELU epilogue, non-square
    (2, 6, 10, 16, 96, 3, False, False, 1),      # Cin=16, Cout=96 (32-wide tile)
    # narrow early 3x3 layers -> conv3x3_narrow.hip (M % 32 == 0); tiles straddle rows and frames, all borders hit
    (2, 12, 12, 64, 64, 3, False, False, 1),     # layer-1 conv2
    (2, 12, 12, 64, 64, 3, True, False, 1),
    (2, 12, 12, 32, 64, 3, False, False, 1),
    (4, 16, 8, 32, 32, 3, False, False, 1),      # non-square
    (1, 4, 8, 64, 64, 3, False, False, 1),       # a single 32-pixel tile
    (3, 20, 20, 32, 64, 3, True, False, 1),      # M = 1200 is not a multiple of 32 -> falls back to conv_igemm
    # row-tile kernel (W % 28 == 0 / POOL: W % 14 == 0): image borders, several segments per row, several frames
    (2, 6, 28, 32, 32, 3, False, False, 1),
    (3, 5, 56, 32, 64, 3, False, False, 1),
    (2, 8, 28, 32, 64, 3, True, False, 1),
    (1, 4, 14, 32, 64, 3, True, False, 1),
    (2, 7, 56, 64, 64, 3, False, False, 1),
    (3, 6, 28, 64, 64, 3, True, False, 1),
    # register-weight 1x1 kernel (conv_pair.hip: conv1x1_regw) -- needs >= 1024 32-pixel tiles
    (42, 28, 28, 256, 512, 1, False, False, 0),  # layer-2 downsample conv (no activation)
    (42, 28, 28, 512, 256, 1, False, False, 1),  # layer-3 first conv1
    (42, 28, 28, 128, 512, 1, False, True, 1),   # layer-2 last conv3 + residual
    (96, 14, 14, 256, 1024, 1, False, True, 1),  # layer-3 conv3 + residual: two channel groups
    # 256 frames of 14x14: 512 padded (196-of-224-row) tiles -> the quantisation-friendly tile config of conv_igemm
    (256, 14, 14, 1024, 256, 1, False, False, 1),
    (256, 14, 14, 128, 256, 3, False, False, 1),
])
def test_conv_bf16_matches_oracle(dev, B, H, W, Cin, Cout, ks, pool, res, act):
    got, ref = _conv_case(dev, B, H, W, Cin, Cout, ks, pool, res, act, seed=B * 1000 + Cin + Cout + ks)
    assert got.shape == ref.shape
    # fp32-accumulated result rounded once to bf16: half-ulp 2^-9 relative on each element
    assert _rel(got, ref) < 4e-3, _rel(got, ref)
    assert (got - ref).abs().max() <= 2e-2 * ref.abs().max() + 1e-3


@pytest.mark.parametrize("big,B,H,W,Cin,Cout,ks,pool,res,act", [
    # conv_igemm8 (8-wave ping-pong, 256 x 256 tiles) under its production rule (EC_CONV_BIG=1: 3x3, Cout % 256 == 0,
    # >= 150 tiles): tiles straddle frames, every border is hit, the last tile is ragged (M % 256 != 0)
    (1, 201, 14, 14, 64, 256, 3, False, False, 1),       # M = 39396 -> 154 tiles, K = 576 (9 K-tiles, one per tap)
    (1, 51, 28, 28, 64, 256, 3, True, False, 1),         # fused AvgPool2d(2): quad-ordered rows, M = 39984
    (1, 170, 7, 7, 128, 512, 3, False, False, 1),        # 7x7 maps, two N tiles, two K-tiles per tap
    # every instantiation (EC_CONV_BIG=4: wherever the preconditions hold): 1x1 with a residual, BN = 128, short nk
    (4, 45, 14, 14, 512, 256, 1, False, True, 1),
    (4, 45, 14, 14, 512, 128, 1, False, False, 0),
    (4, 43, 14, 14, 64, 128, 3, False, False, 1),
    (4, 12, 28, 28, 64, 128, 3, True, False, 1),
    (4, 9, 30, 30, 64, 256, 1, False, False, 2),         # K = 64 < 512: must fall back to the 4-wave kernel
    (1, 50, 14, 14, 768, 1024, 1, False, False, 2),      # ViT c_fc-like: Cin = 768 (not a power of two), QuickGELU, production rule
    (1, 200, 7, 7, 2048, 256, 1, False, True, 1),        # long-K 1x1 with a residual (ViT c_proj-like), production rule
])
def test_conv_igemm8_pingpong_matches_oracle(dev, monkeypatch, big, B, H, W, Cin, Cout, ks, pool, res, act):
    """The 8-wave kernel's cross-wave LDS hand-offs (counted waits + raw barriers) are exercised on many tiles per
    launch and repeated launches: a race shows up as rare wrong tiles, so every output element is compared, 3 times."""
    import subprocess, sys, os, json
    # EC_CONV_BIG is read once per process: run the case in a child process with the variable set
    code = f"""
import sys, torch
sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
sys.path.insert(0, {os.path.dirname(os.path.abspath(__file__))!r})
import test_gpu_encoder as T
dev = torch.device("cuda:0")
worst = 0.0
for rep in range(3):
    got, ref = T._conv_case(dev, {B}, {H}, {W}, {Cin}, {Cout}, {ks}, {pool}, {res}, {act}, seed=7 + rep)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    err = (got - ref).abs()
    tol = 2e-2 * ref.abs().clamp_min(1.0)
    bad = int((err > tol).sum())
    assert bad == 0, (rep, bad, float(err.max()))
    worst = max(worst, T._rel(got, ref))
print("REL", worst)
"""
    env = dict(os.environ, EC_CONV_BIG=str(big))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rel = float(r.stdout.strip().split("REL")[-1])
    assert rel < 4e-3, rel


def test_gemm_bf16_tail_shapes(dev):
    from embodied_clip_amd import encoder as enc
    g = torch.Generator().manual_seed(7)
    for (M, N, K) in [(50, 768, 768), (100, 2304, 768), (37, 64, 200), (300, 32, 3072)]:
        a = _bf(torch.randn(M, K, generator=g)); w = _bf(torch.randn(N, K, generator=g) * K ** -0.5)
        b = torch.randn(N, generator=g) * 0.1
        ref = a.float() @ w.float().t() + b
        got = enc.gemm_bf16(a.to(dev), w.to(dev), b.to(dev)).cpu().float()
        assert _rel(got, ref) < 4e-3, (M, N, K, _rel(got, ref))


def test_stem_pool_layout_kernels(dev):
    from embodied_clip_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    # stem conv1 vs F.conv2d
    x = torch.randn(2, 40, 40, 3, generator=g)
    w = torch.randn(32, 3, 3, 3, generator=g) * 0.2
    b = torch.randn(32, generator=g) * 0.1
    ref = F.relu(F.conv2d(x.permute(0, 3, 1, 2), w, b, stride=2, padding=1)).permute(0, 2, 3, 1)
    wk = w.permute(2, 3, 1, 0).reshape(27, 32).contiguous().to(dev)
    out = torch.empty(2, 20, 20, 32, dtype=torch.bfloat16, device=dev)
    xd, bd = x.to(dev), b.to(dev)    # keep device buffers alive: data_ptr() of a temporary dangles
    _lib.check(lib.ec_stem_conv1(xd.data_ptr(), wk.data_ptr(), bd.data_ptr(), out.data_ptr(), 2, 40, 40, 32, 0))
    torch.cuda.synchronize()
    assert _rel(out.cpu(), ref) < 4e-3
    # avgpool2
    a = _bf(torch.randn(3, 6, 10, 16, generator=g))
    o = torch.empty(3, 3, 5, 16, dtype=torch.bfloat16, device=dev)
    ad = a.to(dev)
    _lib.check(lib.ec_avgpool2_bf16(ad.data_ptr(), o.data_ptr(), 3, 6, 10, 16, 0))
    refp = F.avg_pool2d(a.float().permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    assert _rel(o.cpu(), refp) < 4e-3
    # nhwc bf16 -> nchw f32 is exact
    f = _bf(torch.randn(3, 49, 128, generator=g))
    o2 = torch.empty(3, 128, 49, dtype=torch.float32, device=dev)
    fd = f.to(dev)
    _lib.check(lib.ec_nhwc_bf16_to_nchw_f32(fd.data_ptr(), o2.data_ptr(), 3, 49, 128, 0))
    assert torch.equal(o2.cpu(), f.float().permute(0, 2, 1).contiguous())
    o3 = torch.empty(3, 128, dtype=torch.float32, device=dev)
    _lib.check(lib.ec_spatial_mean_bf16(fd.data_ptr(), o3.data_ptr(), 3, 49, 128, 0))
    assert torch.allclose(o3.cpu(), f.float().mean(1), atol=1e-5)


@pytest.mark.parametrize("width,layers,res", [(64, (1, 1, 1, 1), 64), (64, (3, 4, 6, 3), 224)])
def test_rn50_trunk_matches_oracle(dev, width, layers, res):
    from embodied_clip_amd.encoder import RN50Trunk
    sd = syn.rn50_visual_state_dict(11, width=width, layers=layers, output_dim=64, heads=4, input_resolution=res)
    trunk = RN50Trunk(sd, device=dev, input_resolution=res)
    B = 3
    rgb = syn.synthetic_rgb(1000, B, res)
    feat = trunk.forward(rgb.to(dev))
    got = trunk.to_nchw_f32(feat).cpu()
    x = rgb.permute(0, 3, 1, 2)
    ref_emul = ocr.rn50_trunk(x, sd, emulate_bf16=True)
    ref_fp32 = ocr.rn50_trunk(x, sd)
    assert got.shape == ref_fp32.shape
    assert _rel(got, ref_emul) < 4e-3 * (1 + sum(layers)) ** 0.5, _rel(got, ref_emul)
    assert _rel(got, ref_fp32) < 2e-2, _rel(got, ref_fp32)
    cos = F.cosine_similarity(got.flatten(1), ref_fp32.flatten(1)).min().item()
    assert cos > 0.999, cos
    # sub-batch chunking changes the launch shapes (3 frames vs 2 + 1): the same features up to fp32-accumulation
    # rounding (the K partition of the small layer-3/4 launches depends on the launch's tile count: tests/test_gpu_splitk.py)
    trunk.chunk = 2
    feat2 = trunk.forward(rgb.to(dev))
    assert _rel(feat2.cpu(), feat.cpu()) <= 7e-3       # across launch shapes: measured 4-5e-3
    # pool=True head
    pooled = trunk.spatial_mean(feat).cpu()
    assert torch.allclose(pooled, got.mean((2, 3)), atol=1e-3 * got.abs().max().item())


def test_preprocessor_api_surface(dev):
    from embodied_clip_amd.clip_preprocessors import ClipResNetPreprocessor
    sd = syn.rn50_visual_state_dict(0)
    pre = ClipResNetPreprocessor(rgb_input_uuid="rgb_lowres", clip_model_type="RN50", pool=False,
                                 output_uuid="rgb_clip_resnet", state_dict=sd, device=dev)
    assert pre.input_uuids == ["rgb_lowres"] and pre.uuid == "rgb_clip_resnet"
    assert pre.observation_space.shape == (2048, 7, 7)
    rgb = syn.synthetic_rgb(5, 2)            # CPU fp32 NHWC, as the sensor hands it over
    out = pre.to(dev).process({"rgb_lowres": rgb})
    assert out.shape == (2, 2048, 7, 7) and out.dtype == torch.float32 and out.is_cuda
    ref = ocr.clip_resnet_preprocessor(rgb, sd)
    assert _rel(out.cpu(), ref) < 2e-2
    pre_p = ClipResNetPreprocessor("rgb_lowres", "RN50", pool=True, state_dict=sd, device=dev)
    assert pre_p.observation_space.shape == (2048,)
    outp = pre_p.process({"rgb_lowres": rgb})
    assert outp.shape == (2, 2048)
    assert _rel(outp.cpu(), ocr.clip_resnet_preprocessor(rgb, sd, pool=True)) < 2e-2


def test_uint8_input_path_matches_normalised_fp32_path(dev):
    """raw uint8 frames in, CLIP normalisation fused into the stem kernel."""
    from embodied_clip_amd.clip_preprocessors import ClipResNetPreprocessor
    from embodied_clip_amd.encoder import RN50Trunk
    sd = syn.rn50_visual_state_dict(0)
    u8 = syn.synthetic_rgb_u8(77, 3)
    trunk = RN50Trunk(sd, device=dev)
    a = trunk.forward(syn.normalize_rgb(u8).to(dev)).cpu()
    b = trunk.forward_u8(u8.to(dev)).cpu()
    # same math up to fp32 rounding of the normalisation; 1-ulp input differences flip bf16 roundings that
    # propagate through 50 layers, so the bound is the bf16 noise floor of the trunk, not fp32
    assert _rel(b, a) < 1e-2, _rel(b, a)
    # stem conv1 alone (one bf16 rounding): tight
    import ctypes as C
    from embodied_clip_amd import _lib
    lib = _lib.load()
    w = torch.randn(27, 32, device=dev) * 0.2
    bias = torch.randn(32, device=dev) * 0.1
    xf = syn.normalize_rgb(u8).to(dev).contiguous()
    xu = u8.to(dev).contiguous()
    o1 = torch.empty(3, 112, 112, 32, dtype=torch.bfloat16, device=dev)
    o2 = torch.empty_like(o1)
    _lib.check(lib.ec_stem_conv1(xf.data_ptr(), w.data_ptr(), bias.data_ptr(), o1.data_ptr(), 3, 224, 224, 32,
                                 _lib.stream_ptr()), "stem")
    m3 = (C.c_float * 3)(*syn.CLIP_RGB_MEANS)
    s3 = (C.c_float * 3)(*syn.CLIP_RGB_STDS)
    _lib.check(lib.ec_stem_conv1_u8(xu.data_ptr(), m3, s3, w.data_ptr(), bias.data_ptr(), o2.data_ptr(), 3, 224, 224,
                                    32, _lib.stream_ptr()), "stem_u8")
    torch.cuda.synchronize()
    d = (o1.float() - o2.float()).abs()
    assert d.max().item() <= 2 ** -7 * o1.float().abs().max().item()      # at most one bf16 ulp
    assert (d > 0).float().mean().item() < 0.01
    ref = ocr.clip_resnet_preprocessor(syn.normalize_rgb(u8), sd)
    pre = ClipResNetPreprocessor("rgb", "RN50", pool=False, state_dict=sd, device=dev)
    assert _rel(pre.process({"rgb": u8}).cpu(), ref) < 2e-2


@pytest.mark.parametrize("two,res,N2,M", [(False, True, 64, 32 * 7), (True, False, 64, 32 * 1031), (False, True, 128, 32 * 300),
                                          (False, False, 64, 32), (True, True, 64, 32 * 130), (False, False, 128, 32 * 1025)])
def test_conv1x1_pair_matches_two_conv_launches_and_fp32(dev, two, res, N2, M):
    """Fused layer-1 block boundary (conv3 [+downsample] + identity + ReLU -> next conv1 + ReLU) vs the same math as
    separate ec_gemm_bf16 launches (same bf16 roundings; fp32 summation order differs) and vs an fp32 reference."""
    from embodied_clip_amd.encoder import conv1x1_pair_bf16, gemm_bf16
    g = torch.Generator().manual_seed(M + N2 + 2 * two + res)
    bf = lambda t: t.to(torch.bfloat16).to(dev)  # noqa: E731
    a0, a1 = bf(torch.randn(M, 64, generator=g).relu()), bf(torch.randn(M, 64, generator=g).relu())
    w0, w1 = bf(torch.randn(256, 64, generator=g) * 0.15), bf(torch.randn(256, 64, generator=g) * 0.15)
    w2 = bf(torch.randn(N2, 256, generator=g) * 0.08)
    b0, b1, b2 = (torch.randn(n, generator=g).mul(0.3).to(dev) for n in (256, 256, N2))
    r = bf(torch.randn(M, 256, generator=g).relu()) if res else None
    y, z = conv1x1_pair_bf16(a0, w0, b0, w2, b2, a1=a1 if two else None, w1=w1 if two else None,
                             b1=b1 if two else None, res=r)
    # fp32 reference on the same bf16 inputs
    yf = a0.float() @ w0.float().t() + b0
    if two:
        yf = yf + a1.float() @ w1.float().t() + b1
    if res:
        yf = yf + r.float()
    yf = yf.relu()
    assert _rel(y.float().cpu(), yf.cpu()) < 3e-3
    zf = (y.float() @ w2.float().t() + b2).relu()            # z from the kernel's OWN bf16 y: isolates GEMM 2
    assert _rel(z.float().cpu(), zf.cpu()) < 3e-3
    zz = z.float() - zf
    assert zz.abs().max().item() <= 2 ** -7 * max(1.0, zf.abs().max().item())       # <= 1 bf16 ulp of the largest value
    if not two:
        # bit-level agreement with the unfused launches up to rare 1-ulp rounding flips
        y2 = gemm_bf16(a0, w0, b0, res=r, act=1)
        z2 = gemm_bf16(y2, w2, b2, act=1)
        assert (y2 != y).float().mean().item() < 1e-3
        assert _rel(z.float().cpu(), z2.float().cpu()) < 2e-3


def test_conv1x1_pair_rejects_unsupported_shapes(dev):
    from embodied_clip_amd import _lib
    lib = _lib.load()
    t = torch.zeros(64, 256, dtype=torch.bfloat16, device=dev)
    f = torch.zeros(256, device=dev)
    args = lambda M, K0, N, N2: lib.ec_conv1x1_pair_bf16(t.data_ptr(), t.data_ptr(), f.data_ptr(), None, None, None, None,  # noqa: E731
                                                         t.data_ptr(), t.data_ptr(), f.data_ptr(), t.data_ptr(), M, K0, N, N2, 0)
    assert args(33, 64, 256, 64) == _lib.EC_ERR_SHAPE if hasattr(_lib, "EC_ERR_SHAPE") else args(33, 64, 256, 64) != 0
    assert args(32, 128, 256, 64) != 0 and args(32, 64, 512, 64) != 0 and args(32, 64, 256, 32) != 0


def test_rn50_fused_layer1_boundaries_match_unfused_plan(dev, tmp_path):
    """EC_RN50_FUSE=0 builds the plain one-launch-per-conv plan; the default plan fuses the three layer-1 block
    boundaries (conv_pair.hip).  Same bf16 roundings except the block-0 identity, which the fused plan keeps in fp32.
    The library reads its switches ONCE per process (ec_config), so the plain plan runs in a child process."""
    import os
    import subprocess
    import sys
    from embodied_clip_amd.encoder import RN50Trunk
    sd = syn.rn50_visual_state_dict(0)
    x = syn.synthetic_rgb(5, 3).to(dev)
    fused = RN50Trunk(sd, device=dev)
    out = str(tmp_path / "plain.pt")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, torch; sys.path.insert(0, %r)\n"
            "from embodied_clip_amd import synthetic as syn\n"
            "from embodied_clip_amd.encoder import RN50Trunk\n"
            "t = RN50Trunk(syn.rn50_visual_state_dict(0), device='cuda:0')\n"
            "x = syn.synthetic_rgb(5, 3).to('cuda:0')\n"
            "f = t.forward(x)\n"
            "torch.save({'feat': f.float().cpu(), 'nchw': t.to_nchw_f32(f).cpu(), 'ops': t.lib.ec_rn50_num_ops(t.h),"
            " 'hash': t.plan_hash()}, %r)\n") % (root, out)
    r = subprocess.run([sys.executable, "-c", code], env={**os.environ, "EC_RN50_FUSE": "0"}, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    plain = torch.load(out)
    # layer 1: 3 conv1 + 1 downsample + 1 avgpool (emitted by the layer-1 -> layer-2 boundary launch) launches gone;
    # layer 2: 3 conv1 launches gone
    assert fused.lib.ec_rn50_num_ops(fused.h) == plain["ops"] - 8
    assert fused.plan_hash() != plain["hash"]           # the switch is part of the plan hash
    a, b = fused.forward(x).float().cpu(), plain["feat"]
    assert _rel(a, b) < 1e-2, _rel(a, b)
    ref = ocr.clip_resnet_preprocessor(x.cpu(), sd)
    ra, rb = _rel(fused.to_nchw_f32(fused.forward(x)).cpu(), ref), _rel(plain["nchw"], ref)
    assert ra < 2e-2 and rb < 2e-2, (ra, rb)


def _conv_in_child(env, cases, tmp_path, tag):
    """Runs ec_conv_bf16 over `cases` in a child process with extra environment switches; returns the outputs."""
    import os
    import subprocess
    import sys
    out = str(tmp_path / f"conv_{tag}.pt")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, torch; sys.path.insert(0, %r)\n"
            "from embodied_clip_amd import encoder as enc\n"
            "res = []\n"
            "for (B, H, Cin, Cout, ks, pool) in %r:\n"
            "    g = torch.Generator().manual_seed(B * 1000 + H)\n"
            "    x = torch.randn(B, H, H, Cin, generator=g).to(torch.bfloat16).cuda()\n"
            "    w = (torch.randn(Cout, ks * ks * Cin, generator=g) * 0.05).to(torch.bfloat16).cuda()\n"
            "    b = torch.randn(Cout, generator=g).cuda()\n"
            "    res.append(enc.conv_bf16(x, w, b, None, ksize=ks, pool=bool(pool), act=1).cpu())\n"
            "torch.save(res, %r)\n") % (root, cases, out)
    r = subprocess.run([sys.executable, "-c", code], env={**os.environ, **env}, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return torch.load(out)


def test_ring_and_multirow_kernels_are_bit_identical_to_the_single_stage_kernels(dev, tmp_path):
    """Round-3 kernels that change the SCHEDULE only: the multi-stage ring pipeline of conv_igemm (small launches) and the
    multi-row tiles of the narrow 3x3 layers walk K in the same order as the kernels they replace, so their outputs are
    bit-identical (what keeps a frame's features independent of the batch it was encoded in)."""
    cases = [(2, 14, 256, 256, 3, 0),      # 64x64 ring tiles (4 stages)
             (8, 7, 512, 512, 3, 0),       # 64x64 ring, 72 K-tiles
             (16, 14, 1024, 256, 1, 0),    # 1x1, ring
             (40, 14, 256, 256, 3, 0),     # 128x128 ring tiles (3 stages)
             (3, 56, 64, 64, 3, 0),        # layer-1 conv2: multi-row tiles (RT = 2)
             (2, 112, 32, 32, 3, 0)]       # stem conv2: multi-row tiles (RT = 4)
    new = _conv_in_child({}, cases, tmp_path, "new")
    old = _conv_in_child({"EC_CONV_RING": "0", "EC_CONV_ROWSN": "0"}, cases, tmp_path, "old")
    for c, a, b in zip(cases, new, old):
        assert torch.equal(a, b), c


def test_rn50x16_style_width96_trunk_and_preprocessor(dev):
    """ClipResNetPreprocessor('RN50x16'): width 96 (non-power-of-two channels, 48-channel stem zero-padded to 64).
    A shallow width-96 tower (1 block per layer) keeps the CPU oracle fast; channel arithmetic is the x16 one."""
    from embodied_clip_amd.clip_preprocessors import ClipResNetPreprocessor
    sd = syn.rn50_visual_state_dict(11, width=96, layers=(1, 1, 1, 1), output_dim=768, heads=48)
    x = syn.synthetic_rgb(9, 2)
    ref = ocr.clip_resnet_preprocessor(x, sd)
    assert ref.shape == (2, 3072, 7, 7)
    pre = ClipResNetPreprocessor("rgb", "RN50x16", pool=False, state_dict=sd, device=dev)
    assert pre.observation_space.shape == (3072, 7, 7)
    got = pre.process({"rgb": x}).cpu()
    assert got.shape == ref.shape
    assert _rel(got, ref) < 2e-2, _rel(got, ref)
    cos = torch.nn.functional.cosine_similarity(got.flatten(1), ref.flatten(1)).min().item()
    assert cos > 0.999, cos
    pooled = ClipResNetPreprocessor("rgb", "RN50x16", pool=True, state_dict=sd, device=dev)
    assert pooled.observation_space.shape == (3072,)
    assert _rel(pooled.process({"rgb": x}).cpu(), ref.mean(dim=(2, 3))) < 2e-2


@pytest.mark.parametrize("B,H,W", [(1, 4, 8), (2, 8, 16), (3, 12, 24), (2, 56, 56)])
def test_conv1x1_pair_pool_matches_plain_pair_plus_avgpool(dev, B, H, W):
    """The pooled-output variant of the layer-1 -> layer-2 boundary: y and z bit-identical to the plain fused pair
    (same arithmetic, quad-ordered tiles), y_pooled == AvgPool2d(2) of the bf16 y (fp32 mean, one rounding)."""
    from embodied_clip_amd.encoder import conv1x1_pair_bf16, conv1x1_pair_pool_bf16
    g = torch.Generator().manual_seed(B * 100 + H)
    bf = lambda t: t.to(torch.bfloat16).to(dev)  # noqa: E731
    a0 = bf(torch.randn(B, H, W, 64, generator=g).relu())
    r = bf(torch.randn(B, H, W, 256, generator=g).relu())
    w0 = bf(torch.randn(256, 64, generator=g) * 0.15)
    w2 = bf(torch.randn(128, 256, generator=g) * 0.08)
    b0, b2 = torch.randn(256, generator=g).mul(0.3).to(dev), torch.randn(128, generator=g).mul(0.3).to(dev)
    y, yp, z = conv1x1_pair_pool_bf16(a0, w0, b0, r, w2, b2)
    M = B * H * W
    if M % 32 == 0:
        y2, z2 = conv1x1_pair_bf16(a0.reshape(M, 64), w0, b0, w2, b2, res=r.reshape(M, 256))
        assert torch.equal(y.reshape(M, 256), y2) and torch.equal(z.reshape(M, 128), z2)
    yf = (a0.float().reshape(M, 64) @ w0.float().t() + b0 + r.float().reshape(M, 256)).relu()
    assert _rel(y.float().reshape(M, 256).cpu(), yf.cpu()) < 3e-3
    ref_p = torch.nn.functional.avg_pool2d(y.float().permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1).to(torch.bfloat16)
    assert torch.equal(yp, ref_p)


@pytest.mark.parametrize("M", [32, 32 * 7, 32 * 1025, 256 * 784])
def test_conv1x1_pair_layer2_geometry_matches_two_launches(dev, M):
    """Layer-2 block boundary (128 -> 512 + residual + ReLU -> 128): weights in registers, y tile shared through LDS."""
    from embodied_clip_amd.encoder import conv1x1_pair_bf16, gemm_bf16
    g = torch.Generator().manual_seed(M % 9973)
    bf = lambda t: t.to(torch.bfloat16).to(dev)  # noqa: E731
    a0 = bf(torch.randn(M, 128, generator=g).relu())
    r = bf(torch.randn(M, 512, generator=g).relu())
    w0 = bf(torch.randn(512, 128, generator=g) * 0.1)
    w2 = bf(torch.randn(128, 512, generator=g) * 0.06)
    b0, b2 = torch.randn(512, generator=g).mul(0.3).to(dev), torch.randn(128, generator=g).mul(0.3).to(dev)
    y, z = conv1x1_pair_bf16(a0, w0, b0, w2, b2, res=r)
    y2 = gemm_bf16(a0, w0, b0, res=r, act=1)
    z2 = gemm_bf16(y2, w2, b2, act=1)
    torch.cuda.synchronize()
    assert (y2 != y).float().mean().item() < 1e-3                      # same roundings up to rare 1-ulp flips
    assert _rel(y.float().cpu(), y2.float().cpu()) < 1e-3
    assert _rel(z.float().cpu(), z2.float().cpu()) < 2e-3
    if M <= 32 * 1025:
        yf = (a0.float() @ w0.float().t() + b0 + r.float()).relu()
        assert _rel(y.float().cpu(), yf.cpu()) < 3e-3
        zf = (y.float() @ w2.float().t() + b2).relu()
        assert _rel(z.float().cpu(), zf.cpu()) < 3e-3


def test_preprocessor_loads_fp16_checkpoint_from_weights_dir(dev, tmp_path, monkeypatch):
    """a2: `ClipResNetPreprocessor('rgb', 'RN50', ...)` with no state_dict finds `$EC_CLIP_WEIGHTS_DIR/RN50.pt`, a
    whole-CLIP fp16 state_dict with `visual.`-prefixed keys (what `clip.load(...).state_dict()` holds), and matches the
    oracle run on the same fp16-rounded weights."""
    from embodied_clip_amd.clip_preprocessors import ClipResNetPreprocessor
    sd = syn.rn50_visual_state_dict(5)
    full = {"visual." + k: v.half() for k, v in sd.items()}
    full["logit_scale"] = torch.tensor(4.6).half()
    torch.save(full, tmp_path / "RN50.pt")
    monkeypatch.setenv("EC_CLIP_WEIGHTS_DIR", str(tmp_path))
    pre = ClipResNetPreprocessor("rgb", "RN50", pool=False, device=dev)
    rgb = syn.synthetic_rgb(11, 2)
    out = pre.process({"rgb": rgb})
    assert out.shape == (2, 2048, 7, 7) and out.dtype == torch.float32 and out.device.type == "cuda"
    ref = ocr.clip_resnet_preprocessor(rgb, {k: v.half().float() for k, v in sd.items()})
    assert _rel(out.cpu(), ref) < 2e-2


def test_real_clip_rn50_checkpoint_if_present(dev):
    """With a REAL OpenAI `RN50.pt` in $EC_CLIP_WEIGHTS_DIR (none ships with this repo: no network in the build image)
    the HIP embedding is compared with the fp32 oracle on the real weights; skipped otherwise."""
    import os
    d = os.environ.get("EC_CLIP_WEIGHTS_DIR", "")
    path = os.path.join(d, "RN50.pt")
    if not d or not os.path.exists(path) or os.path.getsize(path) < 100_000_000:
        pytest.skip("no real RN50.pt in $EC_CLIP_WEIGHTS_DIR")
    from embodied_clip_amd.clip_preprocessors import ClipResNetPreprocessor, _load_visual_state_dict
    sd = {k: v.float() for k, v in _load_visual_state_dict("RN50", None, path).items()}
    assert ocr.param_count(sd) == 38_316_896
    pre = ClipResNetPreprocessor("rgb", "RN50", pool=True, device=dev, weights_path=path)
    rgb = syn.synthetic_rgb(12, 4)
    out = pre.process({"rgb": rgb}).cpu()
    ref = ocr.clip_resnet_preprocessor(rgb, sd, pool=True)
    cos = F.cosine_similarity(out, ref, dim=1)
    print("real RN50.pt: cosine vs fp32 oracle", cos.tolist())
    assert cos.min() > 0.999 and _rel(out, ref) < 2e-2


def test_long_segment_variant_of_the_128_wide_8wave_tiles_is_bit_identical(dev, tmp_path):
    """EC_CONV8_LONGSEG (default 1, round 3): the 128-wide tiles of conv_igemm8 run two barrier-separated segments per
    K-tile (all fragments of a K-tile read at once, 16 MFMAs in one segment) out of three LDS stages instead of four
    segments out of two -- a schedule change only: same K walk, same accumulation order, bit-identical outputs."""
    cases = [(100, 28, 128, 128, 3, 0),    # layer-2 3x3 conv: 128-wide tiles, 18 K-tiles
             (104, 56, 128, 128, 3, 1)]    # ... with the fused 2x2 average pool (layer 2, block 0)
    new = _conv_in_child({}, cases, tmp_path, "ls1")
    old = _conv_in_child({"EC_CONV8_LONGSEG": "0"}, cases, tmp_path, "ls0")
    for c, a, b in zip(cases, new, old):
        assert torch.equal(a, b), c


def test_whole_bottleneck_launch_is_bit_identical_to_the_three_conv_launches(dev):
    """bneck23_kernel<.., F1> (round 4): conv1 + bn1 + ReLU in front of the fused conv2 / conv3 launch -- the whole stride-1
    Bottleneck ([U] clip/model.py Bottleneck.forward) in ONE launch, the block input streaming through a shared LDS ring,
    c1 and c2 never in HBM.  Same rounding points and K walks as the three conv_igemm launches: bit-identical; and against a
    torch fp32 reference with c1 / c2 rounded to bf16 where the kernels round them."""
    from embodied_clip_amd import encoder as enc
    C, H = 256, 14
    for B in (1, 3, 130):
        g = torch.Generator().manual_seed(300 + B)
        x = _bf(torch.randn(B, H, H, 4 * C, generator=g).relu())
        w1 = _bf(torch.randn(C, 4 * C, generator=g) * (4 * C) ** -0.5)
        w2 = _bf(torch.randn(C, 3, 3, C, generator=g) * (9 * C) ** -0.5)
        w3 = _bf(torch.randn(4 * C, C, generator=g) * C ** -0.5)
        b1, b2, b3 = (torch.randn(n, generator=g) * 0.1 for n in (C, C, 4 * C))
        d = lambda t: t.to(dev)
        got = enc.bneck_conv123_bf16(d(x), d(w1), d(b1), d(w2.reshape(C, -1)), d(b2), d(w3), d(b3))
        c1u = enc.conv_bf16(d(x), d(w1), d(b1), None, ksize=1, act=1)
        c2u = enc.conv_bf16(c1u, d(w2.reshape(C, -1)), d(b2), None, ksize=3, act=1)
        yu = enc.conv_bf16(c2u, d(w3), d(b3), d(x), ksize=1, act=1)
        torch.cuda.synchronize()
        assert torch.equal(got, yu), B
        xf = x.float().permute(0, 3, 1, 2)
        c1 = F.relu(F.conv2d(xf, w1.float()[:, :, None, None], b1)).to(torch.bfloat16).float()
        c2 = F.relu(F.conv2d(c1, w2.float().permute(0, 3, 1, 2), b2, padding=1)).to(torch.bfloat16).float()
        y = F.relu(F.conv2d(c2, w3.float()[:, :, None, None], b3) + xf).permute(0, 2, 3, 1)
        assert _rel(got.cpu(), y) < 6e-3, (B, _rel(got.cpu(), y))


def test_fused_bottleneck_launch_matches_reference_and_the_two_conv_launches(dev):
    """conv_bneck.hip (round 4): conv2 (3x3) + bn2 + ReLU and conv3 (1x1) + bn3 + identity + ReLU of a stride-1
    Bottleneck ([U] clip/model.py Bottleneck.forward) in one launch, one workgroup per image, the 14 x 14 x 256 map
    resident in LDS.  Against a torch fp32 reference of the same two ops on the same bf16 operands (with c2 rounded to
    bf16 where the kernel rounds it), and bit-identical to the two conv_igemm launches it replaces (same rounding
    points, same K walk per output element).  Frame counts: 1, a ragged 3, and 130 (> one workgroup per CU half)."""
    from embodied_clip_amd import encoder as enc
    C, H = 256, 14
    for B in (1, 3, 130):
        g = torch.Generator().manual_seed(100 + B)
        c1 = _bf(torch.randn(B, H, H, C, generator=g).relu())
        x = _bf(torch.randn(B, H, H, 4 * C, generator=g).relu())
        w2 = _bf(torch.randn(C, 3, 3, C, generator=g) * (9 * C) ** -0.5)
        w3 = _bf(torch.randn(4 * C, C, generator=g) * C ** -0.5)
        b2, b3 = torch.randn(C, generator=g) * 0.1, torch.randn(4 * C, generator=g) * 0.1
        got = enc.bneck_conv23_bf16(c1.to(dev), w2.reshape(C, -1).to(dev), b2.to(dev), w3.to(dev), b3.to(dev), x.to(dev))
        c2u = enc.conv_bf16(c1.to(dev), w2.reshape(C, -1).to(dev), b2.to(dev), None, ksize=3, act=1)
        yu = enc.conv_bf16(c2u, w3.to(dev), b3.to(dev), x.to(dev), ksize=1, act=1)
        torch.cuda.synchronize()
        assert torch.equal(got, yu), B
        c2 = F.relu(F.conv2d(c1.float().permute(0, 3, 1, 2), w2.float().permute(0, 3, 1, 2), b2, padding=1))
        c2 = c2.to(torch.bfloat16).float()                                     # the kernel's one rounding of conv2's output
        y = F.relu(F.conv2d(c2, w3.float()[:, :, None, None], b3) + x.float().permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
        assert _rel(got.cpu(), y) < 4e-3, (B, _rel(got.cpu(), y))
    # unsupported geometry: the caller falls back to the two convs
    lib = __import__("embodied_clip_amd._lib", fromlist=["load"]).load()
    assert lib.ec_bneck_conv23_bf16(1, 1, 1, 1, 1, 1, 2, 7, 7, 512, None) == -2


def test_trunk_with_fused_bottlenecks_is_bit_identical_to_the_unfused_plan(dev, tmp_path):
    """EC_RN50_BNECK (default 128: launches of >= 128 frames run layer3.1 .. layer3.5's conv2 + conv3 as one fused launch
    each): the same features, bit for bit, as the plan without them (child process, EC_RN50_BNECK=0), and fewer ops."""
    import os
    import subprocess
    import sys
    from embodied_clip_amd.encoder import RN50Trunk
    sd = syn.rn50_visual_state_dict(0)
    x = syn.synthetic_rgb(21, 8).repeat(16, 1, 1, 1).roll(2, dims=1)[:128].contiguous().to(dev)
    base = RN50Trunk(sd, device=dev)
    ref = base.forward(x).float().cpu()                     # 128 frames: the fused launches run
    small = base.forward(x[:5].contiguous()).float().cpu()  # 5 frames: the fallback inside the same plan
    out = str(tmp_path / "nobneck.pt")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, torch; sys.path.insert(0, %r)\n"
            "from embodied_clip_amd import synthetic as syn\n"
            "from embodied_clip_amd.encoder import RN50Trunk\n"
            "t = RN50Trunk(syn.rn50_visual_state_dict(0), device='cuda:0')\n"
            "x = syn.synthetic_rgb(21, 8).repeat(16, 1, 1, 1).roll(2, dims=1)[:128].contiguous().to('cuda:0')\n"
            "torch.save({'feat': t.forward(x).float().cpu(), 'small': t.forward(x[:5].contiguous()).float().cpu(),\n"
            "            'hash': t.plan_hash(), 'ops': t.lib.ec_rn50_num_ops(t.h)}, %r)\n") % (root, out)
    r = subprocess.run([sys.executable, "-c", code], env={**os.environ, "EC_RN50_BNECK": "0"}, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = torch.load(out)
    assert got["hash"] != base.plan_hash() and got["ops"] == base.lib.ec_rn50_num_ops(base.h) + 10    # (conv1 + conv2 + conv3 -> 1 op, five times)
    assert torch.equal(got["feat"], ref)
    # 5 frames: the default plan runs layer 3's conv2 on the image-resident K-split kernel (fixed fold order), the child's
    # unfused plan on conv_igemm: equal up to fp32-accumulation rounding amplified through the rest of the trunk
    assert _rel(got["small"], small) <= 7e-3, _rel(got["small"], small)


@pytest.mark.parametrize("H,C", [(14, 256), (7, 512)])
def test_small_launch_image_resident_3x3_kernel(dev, H, C):
    """conv3x3_img_kernel (round 4): the late 3x3 convs of SMALL launches (32-64 frames per GPU: strong scaling's operating
    points) -- one workgroup per (image, channel slice), the map resident in LDS, K split over the eight waves, partial
    tiles folded in a fixed order.  vs a torch fp32 reference of the op on the same bf16 operands; deterministic (two runs
    bit-identical); equal to the pixel-tiled conv_igemm launch up to fp32-accumulation rounding (rare 1-ulp bf16 flips)."""
    from embodied_clip_amd import encoder as enc
    for B in (1, 3, 33):
        g = torch.Generator().manual_seed(7 * B + H)
        x = _bf(torch.randn(B, H, H, C, generator=g).relu())
        w = _bf(torch.randn(C, 3, 3, C, generator=g) * (9 * C) ** -0.5)
        b = torch.randn(C, generator=g) * 0.1
        xd, wd, bd = x.to(dev), w.reshape(C, -1).to(dev), b.to(dev)
        a1 = enc.conv3x3_img_bf16(xd, wd, bd).cpu()
        a2 = enc.conv3x3_img_bf16(xd, wd, bd).cpu()
        plain = enc.conv_bf16(xd, wd, bd, None, ksize=3, act=1).cpu()
        ref = F.relu(F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b, padding=1)).permute(0, 2, 3, 1)
        assert torch.equal(a1, a2), B
        assert _rel(a1, ref) < 4e-3, (B, _rel(a1, ref))
        assert _rel(a1, plain) <= 1e-3 and (a1 != plain).float().mean().item() < 0.01, (B, _rel(a1, plain))
    lib = __import__("embodied_clip_amd._lib", fromlist=["load"]).load()
    assert lib.ec_conv3x3_img_bf16(1, 1, 1, 1, 2, 28, 28, 128, 0, None) == -2    # any other geometry: EC_ERR_SHAPE


def test_small_launch_image_resident_pooled_3x3_kernel(dev):
    """layer4.0's conv2 + CLIP's anti-aliased stride (ReLU, AvgPool2d(2)) at small launches: the 14 x 14 x 512 map (200 KB)
    is made resident in two channel chunks, each chunk's K-tiles split over the eight waves into the same accumulators,
    the fold's tiles pooled through an LDS image.  vs the torch reference, vs conv_igemm's fused-pool launch, deterministic."""
    from embodied_clip_amd import encoder as enc
    C, H = 512, 14
    for B in (1, 3, 33):
        g = torch.Generator().manual_seed(900 + B)
        x = _bf(torch.randn(B, H, H, C, generator=g).relu())
        w = _bf(torch.randn(C, 3, 3, C, generator=g) * (9 * C) ** -0.5)
        b = torch.randn(C, generator=g) * 0.1
        xd, wd, bd = x.to(dev), w.reshape(C, -1).to(dev), b.to(dev)
        a1 = enc.conv3x3_img_bf16(xd, wd, bd, pool=True).cpu()
        a2 = enc.conv3x3_img_bf16(xd, wd, bd, pool=True).cpu()
        plain = enc.conv_bf16(xd, wd, bd, None, ksize=3, pool=True, act=1).cpu()
        ref = F.avg_pool2d(F.relu(F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b, padding=1)), 2).permute(0, 2, 3, 1)
        assert a1.shape == (B, 7, 7, C) and torch.equal(a1, a2), B
        assert _rel(a1, ref) < 4e-3, (B, _rel(a1, ref))
        assert _rel(a1, plain) <= 1e-3, (B, _rel(a1, plain))


@pytest.mark.parametrize("B,H,C,ks,pool,kernel", [
    (64, 28, 256, 3, True, "8-wave"),        # layer3.0 conv2 + pool at a large launch (conv_igemm8 POOL epilogue)
    (3, 28, 256, 3, True, "4-wave"),         # ... at a small one (conv_igemm POOL epilogue)
    (40, 14, 512, 3, True, "4-wave ring"),   # layer4.0 conv2 + pool
    (5, 14, 256, 1, False, "1x1"),           # a plain 1x1 launch
])
def test_conv_writes_a_column_block_of_a_wider_tensor(dev, B, H, C, ks, pool, kernel):
    """``ec_conv_bf16_ld``: the same launch with an output row stride -- the block must equal the dense result bit for bit and
    the neighbouring columns must stay untouched (what the trunk relies on when it lays conv2's pooled output and the pooled
    block input side by side for the K-concatenated conv3 | downsample GEMM)."""
    from embodied_clip_amd import encoder as enc
    g = torch.Generator().manual_seed(77 + B)
    x = _bf(torch.randn(B, H, H, C, generator=g).relu()).to(dev)
    w = _bf(torch.randn(C, ks * ks * C, generator=g) * (ks * ks * C) ** -0.5).to(dev)
    b = (torch.randn(C, generator=g) * 0.1).to(dev)
    dense = enc.conv_bf16(x, w, b, None, ksize=ks, pool=pool, act=1)
    Ho = H // 2 if pool else H
    for c0, wide_c in ((0, 3 * C), (C, 3 * C), (2 * C, 3 * C + 64)):
        wide = torch.full((B, Ho, Ho, wide_c), 7.0, dtype=torch.bfloat16, device=dev)
        enc.conv_bf16(x, w, b, None, ksize=ks, pool=pool, act=1, out=wide[..., c0:c0 + C])
        torch.cuda.synchronize()
        assert torch.equal(wide[..., c0:c0 + C], dense), (kernel, c0)
        rest = torch.cat([wide[..., :c0], wide[..., c0 + C:]], -1)
        assert bool((rest == 7.0).all()), (kernel, c0)


def test_img_kernel_and_avgpool_write_column_blocks(dev):
    from embodied_clip_amd import encoder as enc
    g = torch.Generator().manual_seed(5)
    C = 512
    x = _bf(torch.randn(3, 14, 14, C, generator=g).relu()).to(dev)
    w = _bf(torch.randn(C, 9 * C, generator=g) * (9 * C) ** -0.5).to(dev)
    b = (torch.randn(C, generator=g) * 0.1).to(dev)
    dense = enc.conv3x3_img_bf16(x, w, b, pool=True)
    wide = torch.full((3, 7, 7, C + 1024), -3.0, dtype=torch.bfloat16, device=dev)
    enc.conv3x3_img_bf16(x, w, b, pool=True, out=wide[..., :C])
    xin = _bf(torch.randn(3, 14, 14, 1024, generator=g)).to(dev)
    pooled = enc.avgpool2_bf16(xin)
    enc.avgpool2_bf16(xin, out=wide[..., C:])
    torch.cuda.synchronize()
    assert torch.equal(wide[..., :C], dense) and torch.equal(wide[..., C:], pooled)
    ref = F.avg_pool2d(xin.float().cpu().permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    assert _rel(pooled.cpu(), ref) < 4e-3
    lib = enc._lib.load()
    assert lib.ec_conv3x3_img_bf16_ld(x.data_ptr(), w.data_ptr(), b.data_ptr(), wide.data_ptr(), 3, 14, 14, C, 0, C + 1024, 0) == -2   # EC_ERR_SHAPE: stride needs pool
    assert lib.ec_avgpool2_bf16_ld(xin.data_ptr(), wide.data_ptr(), 3, 14, 14, 1024, 1000, 0) == -2
    assert lib.ec_conv_bf16_ld(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, wide.data_ptr(), 3, 14, 14, C, C, 3, 1, 1, C - 8, 0) == -2


@pytest.mark.parametrize("B,R,K1", [(2, 14, 256), (33, 7, 512)])
def test_downsample_conv_folded_into_conv3_over_the_concatenated_k_axis(dev, B, R, K1):
    """The identity behind EC_RN50_DSCAT: relu(conv3(c2) + b3 + downsample(xp) + bd) == relu([c2 | xp] . [W3 | Wd]^T + (b3 + bd)),
    against the fp32 reference of [U] clip/model.py Bottleneck.forward's last line and against the two chained launches (which
    round the downsample output to bf16 first: the folded form is the closer one)."""
    from embodied_clip_amd import encoder as enc
    g = torch.Generator().manual_seed(31 + B)
    K2, Co = 2 * K1, 4 * K1
    c2 = _bf(torch.randn(B, R, R, K1, generator=g).relu())
    xp = _bf(torch.randn(B, R, R, K2, generator=g).relu())
    w3 = _bf(torch.randn(Co, K1, generator=g) * K1 ** -0.5)
    wd = _bf(torch.randn(Co, K2, generator=g) * K2 ** -0.5)
    b3, bd = torch.randn(Co, generator=g) * 0.1, torch.randn(Co, generator=g) * 0.1
    ref = F.relu(c2.float() @ w3.float().t() + b3 + xp.float() @ wd.float().t() + bd)
    dv = lambda t: t.to(dev)
    cat = torch.empty(B, R, R, K1 + K2, dtype=torch.bfloat16, device=dev)
    cat[..., :K1] = dv(c2); cat[..., K1:] = dv(xp)
    folded = enc.conv_bf16(cat, dv(torch.cat([w3, wd], 1).contiguous()), dv(b3 + bd), None, ksize=1, act=1)
    ds = enc.conv_bf16(dv(xp), dv(wd), dv(bd), None, ksize=1, act=0)
    chained = enc.conv_bf16(dv(c2), dv(w3), dv(b3), ds, ksize=1, act=1)
    torch.cuda.synchronize()
    rf, rc = _rel(folded.cpu(), ref), _rel(chained.cpu(), ref)
    assert rf < 3e-3 and rf <= rc + 1e-4, (rf, rc)
    assert _rel(folded.cpu(), chained.cpu()) < 7e-3


def test_trunk_with_folded_downsample_convs_matches_the_chained_plan(dev, tmp_path):
    """EC_RN50_DSCAT=0 (child process: the library reads its switches once) builds the round-4 plan -- AvgPool2d, downsample conv,
    conv3 + residual -- for layer3.0 / layer4.0; the default plan has one launch less per block and never writes the downsample
    output.  Both against the fp32 oracle; the folded plan rounds once where the chained one rounds twice."""
    import os
    import subprocess
    import sys
    from embodied_clip_amd.encoder import RN50Trunk
    sd = syn.rn50_visual_state_dict(0)
    x = syn.synthetic_rgb(5, 3).to(dev)
    folded = RN50Trunk(sd, device=dev)
    out = str(tmp_path / "chained.pt")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, torch; sys.path.insert(0, %r)\n"
            "from embodied_clip_amd import synthetic as syn\n"
            "from embodied_clip_amd.encoder import RN50Trunk\n"
            "t = RN50Trunk(syn.rn50_visual_state_dict(0), device='cuda:0')\n"
            "x = syn.synthetic_rgb(5, 3).to('cuda:0')\n"
            "f = t.forward(x)\n"
            "torch.save({'nchw': t.to_nchw_f32(f).cpu(), 'ops': t.lib.ec_rn50_num_ops(t.h), 'hash': t.plan_hash()}, %r)\n") % (root, out)
    r = subprocess.run([sys.executable, "-c", code], env={**os.environ, "EC_RN50_DSCAT": "0"}, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    chained = torch.load(out)
    assert folded.lib.ec_rn50_num_ops(folded.h) == chained["ops"] - 2 and folded.plan_hash() != chained["hash"]
    a = folded.to_nchw_f32(folded.forward(x)).cpu()
    ref = ocr.clip_resnet_preprocessor(x.cpu(), sd)
    ra, rb = _rel(a, ref), _rel(chained["nchw"], ref)
    assert ra < 2e-2 and rb < 2e-2 and ra <= rb + 5e-4, (ra, rb)
    assert _rel(a, chained["nchw"]) < 7e-3


def test_pooled_copy_from_layer2s_last_conv3_is_bit_identical_to_the_pooling_pass(dev, tmp_path):
    """EC_RN50_POOLOUT (default 1): layer 2's last conv3 (conv1x1_regw_kernel<.., PL>: tiles of eight 2 x 2 windows in quad
    order) also writes AvgPool2d(2) of its output -- the pooled block input of layer3.0's K-concatenated GEMM ([U] clip/model.py
    Bottleneck.downsample) -- and the avgpool2_kernel launch is skipped.  Same features, bit for bit, as the plan that pools in
    its own pass (child process, EC_RN50_POOLOUT=0) at 128 and 44 frames (the smallest even count with >= 1024 tiles); 43 frames
    (odd: windows not a multiple of 8) and 5 frames run the pooling pass inside the same plan."""
    import os
    import subprocess
    import sys
    from embodied_clip_amd.encoder import RN50Trunk
    sd = syn.rn50_visual_state_dict(0)
    x = syn.synthetic_rgb(33, 8).repeat(16, 1, 1, 1)
    x = torch.stack([x[i].roll(shifts=5 * (i // 8), dims=0) for i in range(128)]).contiguous().to(dev)
    base = RN50Trunk(sd, device=dev)
    mine = {n: base.forward(x[:n].contiguous()).float().cpu() for n in (128, 44, 43, 5)}
    out = str(tmp_path / "nopoolout.pt")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, torch; sys.path.insert(0, %r)\n"
            "from embodied_clip_amd import synthetic as syn\n"
            "from embodied_clip_amd.encoder import RN50Trunk\n"
            "t = RN50Trunk(syn.rn50_visual_state_dict(0), device='cuda:0')\n"
            "x = syn.synthetic_rgb(33, 8).repeat(16, 1, 1, 1)\n"
            "x = torch.stack([x[i].roll(shifts=5 * (i // 8), dims=0) for i in range(128)]).contiguous().to('cuda:0')\n"
            "torch.save({'f': {n: t.forward(x[:n].contiguous()).float().cpu() for n in (128, 44, 43, 5)},\n"
            "            'hash': t.plan_hash(), 'ops': t.lib.ec_rn50_num_ops(t.h)}, %r)\n") % (root, out)
    r = subprocess.run([sys.executable, "-c", code], env={**os.environ, "EC_RN50_POOLOUT": "0"}, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = torch.load(out)
    assert got["hash"] != base.plan_hash() and got["ops"] == base.lib.ec_rn50_num_ops(base.h)   # (the pooling op stays in the plan: skipped per launch)
    for n in (128, 44, 43, 5):
        assert torch.equal(got["f"][n], mine[n]), n
    assert not torch.equal(mine[128][:44], mine[128][44:88])                                     # distinct frames
