"""GPU parity: CLIP ViT-B/32 embedder and AttentionPool2d (through the C-ABI) vs the CPU oracle / golden.

Tolerance: bf16 residual stream + bf16 MFMA operands, fp32 accumulate/statistics:
vs the bf16-emulating oracle rel-L2 <= 1e-2; vs the fp32 oracle rel-L2 <= 3e-2 and per-token cosine >= 0.999.
"""
import os

import pytest
import torch
import torch.nn.functional as F

from embodied_clip_amd import synthetic as syn
from oracle import clip_resnet as ocr
from oracle import clip_vit as ovit

pytestmark = pytest.mark.gpu
G = torch.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_golden.pt"))


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_mha_core_and_layernorm_small_vit(dev):
    """2-block, width-128 ViT (2 heads of 64): exercises patchify, token assembly + ln_pre, LN, the MFMA attention
    core (asymmetric random data => operand/transposition mistakes are caught), QuickGELU MLP, residuals."""
    from embodied_clip_amd.encoder import ViTEmbedder
    sd = syn.vit_visual_state_dict(7, width=128, layers=3, heads=2, patch_size=32, input_resolution=224, output_dim=64)
    rgb = syn.synthetic_rgb(77, 3)
    vit = ViTEmbedder(sd, device=dev, heads=2)
    tok = vit.to_f32(vit.forward(rgb.to(dev))).cpu()
    x = rgb.permute(0, 3, 1, 2)
    ref = ovit.vit_embedder(x, sd, heads=2, drop_last=1)
    ref_emul = ovit.vit_embedder(x, sd, heads=2, drop_last=1, emulate_bf16=True)
    assert tok.shape == ref.shape == (3, 50, 128)
    assert _rel(tok, ref_emul) < 1e-2, _rel(tok, ref_emul)
    assert _rel(tok, ref) < 3e-2, _rel(tok, ref)


def test_vit_b32_matches_oracle_and_golden(dev):
    from embodied_clip_amd.clip_preprocessors import ClipViTPreprocessor
    sd = syn.vit_visual_state_dict(G["vit"]["seed_weights"])
    rgb = syn.synthetic_rgb(G["vit"]["seed_rgb"], 2)
    pre = ClipViTPreprocessor("rgb_lowres", "ViT-B/32", class_emb_only=False, state_dict=sd, device=dev)
    assert pre.observation_space.shape == (50, 768)
    out = pre.process({"rgb_lowres": rgb})
    assert out.shape == (2, 50, 768) and out.dtype == torch.float32 and out.is_cuda
    out = out.cpu()
    assert _rel(out[:, :9], G["vit"]["tokens"]) < 3e-2, _rel(out[:, :9], G["vit"]["tokens"])
    ref = ovit.clip_vit_preprocessor(rgb, sd)
    assert _rel(out, ref) < 3e-2, _rel(out, ref)
    cos = F.cosine_similarity(out.reshape(100, 768), ref.reshape(100, 768)).min().item()
    assert cos > 0.999, cos
    pre_c = ClipViTPreprocessor("rgb_lowres", "ViT-B/32", class_emb_only=True, state_dict=sd, device=dev)
    assert pre_c.observation_space.shape == (768,)
    oc = pre_c.process({"rgb_lowres": rgb}).cpu()
    assert oc.shape == (2, 768) and torch.equal(oc, out[:, 0])


def test_attnpool_matches_oracle_and_golden(dev):
    from embodied_clip_amd.encoder import AttentionPool, RN50Trunk
    sd = syn.rn50_visual_state_dict(G["rn50"]["seed_weights"])
    rgb = syn.synthetic_rgb(G["rn50"]["seed_rgb"], 2)
    trunk = RN50Trunk(sd, device=dev)
    feat = trunk.forward(rgb.to(dev))
    pool = AttentionPool(sd, device=dev)
    out = pool.forward(feat).cpu()
    assert out.shape == (2, 1024)
    # same bf16 features into the oracle's AttentionPool2d: isolates the pool kernels
    ref_same_in = ocr.attnpool(trunk.to_nchw_f32(feat).cpu(), sd)
    assert _rel(out, ref_same_in) < 1.5e-2, _rel(out, ref_same_in)
    # end to end vs the fp32 golden (thor_image_features.py:112 `clip_attnpool`)
    assert _rel(out, G["rn50"]["attnpool"]) < 3e-2, _rel(out, G["rn50"]["attnpool"])
