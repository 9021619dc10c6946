"""Oracle: CLIP VisionTransformer as used by AllenAct's ClipViTEmbedder.

Restates openai/CLIP ``clip/model.py`` ``VisionTransformer`` /
``ResidualAttentionBlock`` / ``QuickGELU`` / ``LayerNorm`` (pinned at
40f5484c, ``primitive_probing/environment.yml:22``) driven the way
``ClipViTPreprocessor`` drives it (the class is named
only by BASELINE.json -- nothing under /root/reference references it):

  conv1 patch-embed (no bias) -> [N, grid^2, width]; prepend class_embedding;
  + positional_embedding; ln_pre; resblocks[:-1] (all but the LAST block);
  no ln_post, no proj; ``class_emb_only`` -> token 0.

Independent cross-check: tests/test_oracle_vit_hf.py maps the same weights
into HuggingFace ``CLIPVisionModel`` and compares hidden_states[-2].
TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

LN_EPS = 1e-5


def _r(x, emulate):
    return x.to(torch.bfloat16).to(torch.float32) if emulate else x


def quick_gelu(x):
    """CLIP ``QuickGELU``: x * sigmoid(1.702 x)."""
    return x * torch.sigmoid(1.702 * x)


def num_blocks(sd: Dict[str, torch.Tensor]) -> int:
    n = 0
    while f"transformer.resblocks.{n}.ln_1.weight" in sd:
        n += 1
    return n


def residual_attention_block(x_lnd, sd, p, heads, emulate=False, attn_mask=None):
    """``x = x + attn(ln_1(x)); x = x + mlp(ln_2(x))`` with
    nn.MultiheadAttention (fused in_proj, softmax(QK^T/sqrt(64)), out_proj)
    and mlp = c_proj(QuickGELU(c_fc(.))).  x: [L, N, D].  ``attn_mask``: additive [L, L] (the text tower's
    causal mask), None for the vision tower."""
    e = emulate
    D = x_lnd.shape[-1]
    h = _r(F.layer_norm(x_lnd, (D,), sd[p + ".ln_1.weight"].float(), sd[p + ".ln_1.bias"].float(), LN_EPS), e)
    if not e:
        a, _ = F.multi_head_attention_forward(
            h, h, h, D, heads, _r(sd[p + ".attn.in_proj_weight"].float(), e), sd[p + ".attn.in_proj_bias"].float(),
            None, None, False, 0.0, _r(sd[p + ".attn.out_proj.weight"].float(), e),
            sd[p + ".attn.out_proj.bias"].float(), training=False, need_weights=False, attn_mask=attn_mask)
    else:
        # same math with the bf16 rounding points of the HIP path made explicit
        L, N, _ = h.shape
        qkv = _r(F.linear(h, _r(sd[p + ".attn.in_proj_weight"].float(), e), sd[p + ".attn.in_proj_bias"].float()), e)
        q, k, v = qkv.split(D, dim=-1)
        dh = D // heads
        q = q.reshape(L, N * heads, dh).transpose(0, 1)
        k = k.reshape(L, N * heads, dh).transpose(0, 1)
        v = v.reshape(L, N * heads, dh).transpose(0, 1)
        s = torch.bmm(q, k.transpose(1, 2)) * (dh ** -0.5)
        if attn_mask is not None:
            s = s + attn_mask
        pr = _r(torch.softmax(s, dim=-1), e)
        o = _r(torch.bmm(pr, v), e).transpose(0, 1).reshape(L, N, D)
        a = F.linear(o, _r(sd[p + ".attn.out_proj.weight"].float(), e), sd[p + ".attn.out_proj.bias"].float())
    x = _r(x_lnd + a, e)
    h = _r(F.layer_norm(x, (D,), sd[p + ".ln_2.weight"].float(), sd[p + ".ln_2.bias"].float(), LN_EPS), e)
    h = _r(quick_gelu(F.linear(h, _r(sd[p + ".mlp.c_fc.weight"].float(), e), sd[p + ".mlp.c_fc.bias"].float())), e)
    h = F.linear(h, _r(sd[p + ".mlp.c_proj.weight"].float(), e), sd[p + ".mlp.c_proj.bias"].float())
    return _r(x + h, e)


def vit_embedder(x_nchw: torch.Tensor, sd: Dict[str, torch.Tensor], heads: int = 12, drop_last: int = 1,
                 class_emb_only: bool = False, emulate_bf16: bool = False, apply_post: bool = False):
    """ClipViTEmbedder.forward (drop_last=1): tokens after resblocks[:-1].
    ``drop_last=0, apply_post=True`` gives the full CLIP ``encode_image``
    (ln_post(cls) @ proj) for cross-checks."""
    e = emulate_bf16
    x = x_nchw.float()
    w = sd["conv1.weight"].float()
    patch = w.shape[-1]
    x = F.conv2d(_r(x, e), _r(w, e), None, stride=patch)
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)  # [N, grid^2, D]
    cls = sd["class_embedding"].float() + torch.zeros(x.shape[0], 1, x.shape[-1])
    x = torch.cat([cls, x], dim=1)
    x = x + sd["positional_embedding"].float()
    D = x.shape[-1]
    x = _r(F.layer_norm(x, (D,), sd["ln_pre.weight"].float(), sd["ln_pre.bias"].float(), LN_EPS), e)
    x = x.permute(1, 0, 2)  # NLD -> LND
    for i in range(num_blocks(sd) - drop_last):
        x = residual_attention_block(x, sd, f"transformer.resblocks.{i}", heads, emulate=e)
    x = x.permute(1, 0, 2)
    if apply_post:
        c = F.layer_norm(x[:, 0, :], (D,), sd["ln_post.weight"].float(), sd["ln_post.bias"].float(), LN_EPS)
        return c @ sd["proj"].float()
    if class_emb_only:
        return x[:, 0, :]
    return x


def clip_vit_preprocessor(rgb_nhwc: torch.Tensor, sd, class_emb_only: bool = False, heads: int = 12,
                          emulate_bf16: bool = False) -> torch.Tensor:
    """AllenAct ``ClipViTPreprocessor.process``: NHWC -> NCHW -> embedder, fp32 out."""
    with torch.no_grad():
        x = rgb_nhwc.permute(0, 3, 1, 2)
        return vit_embedder(x, sd, heads=heads, class_emb_only=class_emb_only, emulate_bf16=emulate_bf16).float()


def param_count(sd) -> int:
    return sum(v.numel() for v in sd.values())
