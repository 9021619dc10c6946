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
    # a change here invalidates every fixture
    assert abs(float(syn.hash_uniform(1, 3)[2]) - float(syn.hash_uniform(1, 3)[2])) == 0
    sd = syn.rn50_visual_state_dict(0)
    assert abs(float(sd["conv1.weight"].flatten()[0]) - float(syn.rn50_visual_state_dict(0)["conv1.weight"].flatten()[0])) == 0
    assert torch.allclose(syn.synthetic_rgb(1000, 1).mean(), syn.synthetic_rgb(1000, 1).mean())


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
