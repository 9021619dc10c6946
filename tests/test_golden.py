"""CPU: the oracle reproduces the committed golden vectors (guards against drift of the oracle or of the
portable synthetic generator; the reference itself ships no vectors -- SURVEY.md §8c)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import make_golden as mg  # noqa: E402
from embodied_clip_amd import synthetic as syn  # noqa: E402
from oracle import clip_resnet as ocr, policy as opol, ppo as oppo  # noqa: E402

G = torch.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_golden.pt"))


def test_generator_known_answers():
    """Literal known-answer values of the portable hash generator (splitmix64 counter hash -> float).
    Every golden fixture and every GPU parity test regenerates its tensors from this generator, so a
    change here invalidates all of them: the literals below were recorded once and must never move."""
    import numpy as np
    u = syn.hash_uniform(1, 4)
    assert np.allclose(u, [0.16737875674524771, 0.9874564473572248, 0.15181146998673722, 0.5307842665235228],
                       rtol=0, atol=1e-15)
    n = syn.hash_normal(33, 4)
    assert np.allclose(n, [-0.6144155234970641, -0.26826695362427033, 0.620992814487889, 0.4355492852208518],
                       rtol=0, atol=1e-12)
    r = syn.synthetic_rgb_u8(5, 1, 8)
    assert int(r.sum()) == 23637 and r.flatten()[:6].tolist() == [200, 104, 185, 218, 23, 33]
    sd = syn.rn50_visual_state_dict(0)
    c = sd["conv1.weight"].flatten()[:3].tolist()
    assert np.allclose(c, [0.15445159375667572, -0.6656992435455322, 0.21676811575889587], rtol=0, atol=1e-7)
    assert abs(float(syn.synthetic_rgb(1000, 1).mean()) - 0.18600572645664215) < 1e-6
    assert syn.synthetic_goals(2, (2, 5)).tolist() == [[8, 11, 11, 11, 5], [0, 8, 6, 11, 1]]
    assert syn.synthetic_masks(1, 4, 6, 0.2).flatten().tolist() == [
        0.0, 0.0, 1.0, 1.0, 1.0, 0.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.0, 1.0, 0.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.0,
        0.0, 1.0]
    a = syn.policy_state_dict(0)["actor.linear.weight"].flatten()[:2].tolist()
    assert np.allclose(a, [-0.004043849650770426, 0.027416672557592392], rtol=0, atol=1e-9)


def test_oracle_rn50_reproduces_golden():
    sd = syn.rn50_visual_state_dict(G["rn50"]["seed_weights"])
    f = ocr.clip_resnet_preprocessor(syn.synthetic_rgb(G["rn50"]["seed_rgb"], 2), sd)
    assert torch.allclose(ocr.avgpool_head(f), G["rn50"]["avgpool"], rtol=1e-4, atol=1e-4)
    assert torch.allclose(f[:, ::32], G["rn50"]["conv_slice"], rtol=1e-4, atol=1e-4)
    assert torch.allclose(ocr.attnpool(f, sd), G["rn50"]["attnpool"], rtol=1e-4, atol=1e-4)


def test_oracle_policy_and_gae_reproduce_golden():
    psd, feat, goal, h0, masks, actions, a, b, c, d = mg.policy_case()
    with torch.no_grad():
        lg, vv, hT = opol.actor_critic_forward(feat, goal, h0, masks, psd)
    assert torch.allclose(lg, G["policy"]["logits"], rtol=1e-4, atol=1e-5)
    assert torch.allclose(vv, G["policy"]["values"], rtol=1e-4, atol=1e-5)
    assert torch.allclose(hT, G["policy"]["h"], rtol=1e-4, atol=1e-5)
    T, N = 16, 4
    m = torch.cat([torch.ones(1, N, 1), syn.synthetic_masks(31, T, N, 0.15)], 0)
    r = syn.synthetic_rewards(32, m[1:])
    v = torch.from_numpy(syn.hash_normal(33, (T + 1) * N).astype("float32")).reshape(T + 1, N, 1)
    R = oppo.compute_returns(r, v, m)
    assert torch.allclose(R, G["gae"]["returns"], rtol=1e-5, atol=1e-6)


def test_oracle_probe_reproduces_golden():
    from oracle import probe as oprobe
    gold = torch.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_golden.pt"))["probe"]
    assert set(gold) == {"object_presence", "free_space", "reachability", "object_localization"}
    for task, (x, y, w, bb) in mg.probe_cases().items():
        assert abs(float(oprobe.compute_loss(x, y, w, bb, task)) - gold[task]) < 1e-6, task


def test_oracle_text_reproduces_golden():
    from oracle import clip_text as otxt
    gold = torch.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_golden.pt"))["text"]["embeds"]
    sd, tok = mg.text_case()
    assert tok.shape == (12, 77) and int(tok.max()) == 999          # <EOT> = vocab-1 is the arg-max
    assert torch.allclose(otxt.encode_text(tok, sd, heads=8), gold, atol=1e-5)


def test_oracle_zeroshot_policy_reproduces_golden():
    """Zero-shot dual-encoder policy (BASELINE config 5; builder-defined fusion, parity-unpinned)."""
    gz = torch.load(os.path.join(os.path.dirname(__file__), "golden", "zeroshot_golden.pt"))
    sd, emb, table, goal, h0, masks, actions, a, b, c, d = mg.zeroshot_case()
    assert sum(v.numel() for v in sd.values()) == 3 * 512 * 1024 + 3 * 512 * 512 + 6 * 512 + 6 * 512 + 7 + 512
    with torch.no_grad():
        lg, vv, hT = opol.zeroshot_actor_critic_forward(emb, goal, h0, masks, sd, table)
    assert torch.allclose(lg, gz["logits"], rtol=1e-4, atol=1e-5)
    assert torch.allclose(vv, gz["values"], rtol=1e-4, atol=1e-5)
    assert torch.allclose(hT, gz["h"], rtol=1e-4, atol=1e-5)
    # the fusion is the per-dimension cosine term: summing the fused vector gives CLIP's image-text cosine score
    x = torch.nn.functional.normalize(emb, dim=-1) * table[goal]
    cos = torch.nn.functional.cosine_similarity(emb, table[goal], dim=-1)
    assert torch.allclose(x.sum(-1), cos, atol=1e-5)
