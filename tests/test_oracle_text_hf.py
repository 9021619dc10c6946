"""Independent cross-check of the CLIP text-tower oracle against HuggingFace CLIPTextModelWithProjection
(same weights, different implementation) -- the goal-embedding source of the zero-shot ObjectNav variant."""
import pytest
import torch

from embodied_clip_amd import synthetic as syn
from oracle import clip_text as otxt


def test_text_param_checksum():
    # CLIP-RN50 text tower: 49408*512 + 77*512 + 12*(4*512^2+4*512 + 8*512^2+5*512 + 4*512) + 2*512 + 512*1024
    sd = syn.text_state_dict(0, vocab_size=1000)
    full = otxt.param_count(sd) + (49408 - 1000) * 512
    assert full == 49408 * 512 + 77 * 512 + 12 * (12 * 512 * 512 + 13 * 512) + 2 * 512 + 512 * 1024 == 63_690_240


def _to_hf(sd, layers):
    m = {"text_model.embeddings.token_embedding.weight": sd["token_embedding.weight"],
         "text_model.embeddings.position_embedding.weight": sd["positional_embedding"],
         "text_model.final_layer_norm.weight": sd["ln_final.weight"],
         "text_model.final_layer_norm.bias": sd["ln_final.bias"],
         "text_projection.weight": sd["text_projection"].t().contiguous()}
    D = sd["ln_final.weight"].numel()
    for i in range(layers):
        p = f"transformer.resblocks.{i}"; h = f"text_model.encoder.layers.{i}"
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


def test_text_oracle_matches_hf_clip_text_model():
    transformers = pytest.importorskip("transformers")
    width, layers, heads, ctx, vocab, out = 64, 3, 4, 20, 300, 48
    sd = syn.text_state_dict(7, width=width, layers=layers, heads=heads, context_length=ctx, vocab_size=vocab,
                             embed_dim=out)
    cfg = transformers.CLIPTextConfig(vocab_size=vocab, hidden_size=width, intermediate_size=4 * width,
                                      num_hidden_layers=layers, num_attention_heads=heads,
                                      max_position_embeddings=ctx, hidden_act="quick_gelu", layer_norm_eps=1e-5,
                                      projection_dim=out, eos_token_id=2, bos_token_id=0, pad_token_id=1)
    hf = transformers.CLIPTextModelWithProjection(cfg).eval()
    mapped = _to_hf(sd, layers)
    keys = hf.state_dict().keys()
    if not any(k.startswith("text_model.") for k in keys):            # transformers>=5 may drop the prefix
        mapped = {(k[len("text_model."):] if k.startswith("text_model.") else k): v for k, v in mapped.items()}
    missing, unexpected = hf.load_state_dict(mapped, strict=False)
    assert not [k for k in missing if "position_ids" not in k], missing
    assert not unexpected, unexpected
    tokens = syn.synthetic_tokens(3, 5, ctx, vocab)
    with torch.no_grad():
        o = hf(input_ids=tokens)       # eos_token_id == 2 selects the legacy arg-max pooling == OpenAI CLIP
    mine = otxt.encode_text(tokens, sd, heads=heads)
    assert torch.allclose(otxt.encode_text(tokens, sd, heads=heads, return_hidden=True), o.last_hidden_state, atol=2e-5)
    assert torch.allclose(mine, o.text_embeds, atol=2e-5)
    # the explicit-attention (bf16-emulation) code path computes the same function when rounding is off
    from oracle.clip_vit import residual_attention_block
    x = torch.randn(ctx, 2, width)
    a = residual_attention_block(x, sd, "transformer.resblocks.0", heads, attn_mask=otxt.causal_mask(ctx))
    assert a.shape == x.shape and torch.isfinite(a).all()
    # causality: changing a later token never changes an earlier position
    t2 = tokens.clone(); t2[:, 6] = (t2[:, 6] + 1) % (vocab - 2)
    h1 = otxt.encode_text(tokens, sd, heads=heads, return_hidden=True)
    h2 = otxt.encode_text(t2, sd, heads=heads, return_hidden=True)
    assert torch.allclose(h1[:, :6], h2[:, :6], atol=1e-6) and not torch.allclose(h1[:, 6:], h2[:, 6:], atol=1e-4)
