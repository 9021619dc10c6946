"""Oracle: CLIP text tower (``CLIP.encode_text``), the goal-embedding source of the zero-shot ObjectNav variant
(readme_files/zeroshot_objectnav.md:3-8; SURVEY.md §8f rank 3 / BASELINE config 5).

Restates openai/CLIP ``clip/model.py`` (pinned at 40f5484c, ``primitive_probing/environment.yml:22``), which is not
vendored under /root/reference:

    x = token_embedding(text) + positional_embedding            # [B, ctx, width]
    x = transformer(x)  with the causal additive mask ``build_attention_mask`` (-inf above the diagonal)
    x = ln_final(x)
    x = x[arange(B), text.argmax(dim=-1)] @ text_projection     # features at the EOT token (highest id)

RN50 CLIP: width 512, 8 heads, 12 layers, ctx 77, vocab 49,408, embed_dim 1024.  The BPE tokenizer is host string
processing and not part of this path: token ids are the input.

Independent cross-check: tests/test_oracle_text_hf.py maps the same weights into HuggingFace
``CLIPTextModelWithProjection``.  TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

from .clip_vit import LN_EPS, _r, num_blocks, residual_attention_block


def causal_mask(ctx: int) -> torch.Tensor:
    """``CLIP.build_attention_mask``: additive, -inf strictly above the diagonal."""
    return torch.full((ctx, ctx), float("-inf")).triu_(1)


def encode_text(tokens: torch.Tensor, sd: Dict[str, torch.Tensor], heads: int = 8, emulate_bf16: bool = False,
                return_hidden: bool = False) -> torch.Tensor:
    """tokens int64 [B, ctx] -> fp32 [B, embed_dim]."""
    e = emulate_bf16
    B, ctx = tokens.shape
    x = sd["token_embedding.weight"].float()[tokens] + sd["positional_embedding"].float()[:ctx]
    x = _r(x, e).permute(1, 0, 2)                     # NLD -> LND
    mask = causal_mask(ctx)
    for i in range(num_blocks(sd)):
        x = residual_attention_block(x, sd, f"transformer.resblocks.{i}", heads, e, attn_mask=mask)
    x = x.permute(1, 0, 2)
    D = x.shape[-1]
    hidden = F.layer_norm(x, (D,), sd["ln_final.weight"].float(), sd["ln_final.bias"].float(), LN_EPS)
    if return_hidden:
        return hidden
    eot = hidden[torch.arange(B), tokens.argmax(dim=-1)]
    return _r(eot, e) @ _r(sd["text_projection"].float(), e)


def param_count(sd) -> int:
    return sum(v.numel() for v in sd.values())
