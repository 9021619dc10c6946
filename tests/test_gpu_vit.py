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


def test_attnpool_rn50x16_geometry_145_tokens(dev):
    """AttentionPool2d of RN50x16 (384 px: 12x12 + 1 = 145 tokens, 64-wide heads; `imagenet_vs_objectnav.md:11`) and of a
    197-token map: the CLS-query core walks any token count.  Reduced width (192 channels, 3 heads) keeps the oracle quick."""
    from embodied_clip_amd.encoder import AttentionPool
    for (S, C, out_dim) in ((12, 192, 64), (14, 128, 32)):
        n = lambda seed, *shape: torch.from_numpy(syn.hash_normal(seed, int(torch.tensor(shape).prod()))).float().reshape(*shape)
        sd = {"attnpool.positional_embedding": n(1, S * S + 1, C) * C ** -0.5}
        for i, name in enumerate(("q_proj", "k_proj", "v_proj")):
            sd[f"attnpool.{name}.weight"] = n(10 + i, C, C) * C ** -0.5
            sd[f"attnpool.{name}.bias"] = n(20 + i, C) * 0.1
        sd["attnpool.c_proj.weight"] = n(30, out_dim, C) * C ** -0.5
        sd["attnpool.c_proj.bias"] = n(31, out_dim) * 0.1
        feat = (n(40, 3, S, S, C).abs() * 0.7).to(torch.bfloat16)             # NHWC trunk features (post-ReLU)
        pool = AttentionPool(sd, device=dev, num_heads=C // 64)
        out = pool.forward(feat.to(dev)).cpu()
        ref = ocr.attnpool(feat.float().permute(0, 3, 1, 2).contiguous(), sd, num_heads=C // 64)
        assert out.shape == ref.shape == (3, out_dim)
        assert _rel(out, ref) < 1.5e-2, (S, _rel(out, ref))


def test_vit_b16_style_197_tokens_general_attention(dev):
    """ClipViTPreprocessor('ViT-B/16'): 14x14+1 = 197 tokens run the general LDS attention core (L > 64)."""
    from embodied_clip_amd.clip_preprocessors import ClipViTPreprocessor
    from oracle import clip_vit as ovit
    sd = syn.vit_visual_state_dict(4, width=768, layers=3, heads=12, patch_size=16, input_resolution=224, output_dim=64)
    x = syn.synthetic_rgb(6, 2)
    ref = ovit.vit_embedder(x.permute(0, 3, 1, 2).contiguous(), sd, heads=12, drop_last=1)
    pre = ClipViTPreprocessor("rgb", "ViT-B/16", class_emb_only=False, state_dict=sd, device=dev)
    assert pre.observation_space.shape == (197, 768)
    got = pre.process({"rgb": x}).cpu()
    assert got.shape == ref.shape == (2, 197, 768)
    rel = float((got - ref).norm() / ref.norm())
    assert rel < 2e-2, rel


def test_text_tower_matches_oracle_and_is_causal(dev):
    """CLIP.encode_text on HIP (ec_text_forward) vs oracle/clip_text.py: small tower and the RN50-CLIP geometry
    (width 512, 8 heads, ctx 77) with a reduced vocabulary / depth to keep the CPU oracle quick."""
    from embodied_clip_amd.encoder import ClipTextEncoder
    from oracle import clip_text as otxt
    for (width, layers, ctx, vocab, out, n) in ((128, 2, 20, 300, 64, 5), (512, 3, 77, 2000, 1024, 12)):
        sd = syn.text_state_dict(3, width=width, layers=layers, heads=width // 64, context_length=ctx,
                                 vocab_size=vocab, embed_dim=out)
        tokens = syn.synthetic_tokens(5, n, ctx, vocab)
        ref = otxt.encode_text(tokens, sd, heads=width // 64)
        enc = ClipTextEncoder(sd, device=dev)
        got = enc.encode_text(tokens).cpu()
        assert got.shape == ref.shape == (n, out)
        rel = float((got - ref).norm() / ref.norm())
        assert rel < 2e-2, (width, rel)
        cos = torch.nn.functional.cosine_similarity(got, ref).min().item()
        assert cos > 0.999, cos
        # padding after EOT never reaches the EOT feature (causal mask): garbage in the padded tail changes nothing
        t2 = tokens.clone()
        eot = tokens.argmax(dim=-1)
        for i in range(n):
            t2[i, int(eot[i]) + 1:] = 1
        assert torch.equal(enc.encode_text(t2).cpu(), got)
        tab = enc.goal_table(tokens).cpu()
        assert torch.allclose(tab.norm(dim=-1), torch.ones(n), atol=1e-5)


def test_folded_layernorm_with_a_large_row_mean(dev):
    """The LayerNorm folded into the QKV / c_fc GEMMs (round 6) computes rstd * (x Wg^T - mean * s) + c on the RAW residual rows:
    the mean term must cancel against the GEMM's own sum.  Stress: a residual stream whose rows have |mean| = 3-4 x their standard
    deviation (ln_pre's bias shifted by +6, block LayerNorm gains spread over 0.25 .. 4) -- same tolerances as the plain tower."""
    from embodied_clip_amd.encoder import ViTEmbedder
    sd = syn.vit_visual_state_dict(11, width=128, layers=3, heads=2, patch_size=32, input_resolution=224, output_dim=64)
    sd = {k: v.clone() for k, v in sd.items()}
    sd["ln_pre.bias"] += 6.0
    g = torch.Generator().manual_seed(3)
    for k in list(sd):
        if k.endswith("ln_1.weight") or k.endswith("ln_2.weight"):
            sd[k] = sd[k] * torch.exp2(torch.rand(sd[k].shape, generator=g) * 4 - 2)
        if k.endswith("ln_1.bias") or k.endswith("ln_2.bias"):
            sd[k] = sd[k] + torch.randn(sd[k].shape, generator=g)
    rgb = syn.synthetic_rgb(78, 3)
    vit = ViTEmbedder(sd, device=dev, heads=2)
    tok = vit.to_f32(vit.forward(rgb.to(dev))).cpu()
    x = rgb.permute(0, 3, 1, 2)
    ref = ovit.vit_embedder(x, sd, heads=2, drop_last=1)
    m, s_ = ref.mean(-1).abs().mean().item(), ref.std(-1).mean().item()
    assert m > 2.0 * s_, (m, s_)                        # the rows really are mean-dominated
    # (centred comparison as well: the mean itself is several times the signal, it would mask an error in the part LayerNorm keeps)
    cen = lambda t: t - t.mean(-1, keepdim=True)
    assert _rel(tok, ref) < 3e-2, _rel(tok, ref)
    assert _rel(cen(tok), cen(ref)) < 3e-2, _rel(cen(tok), cen(ref))
