"""GPU: the HIP path (through the C-ABI) against the COMMITTED golden vectors (no oracle involved at run time
except for regenerating the portable inputs)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import make_golden as mg  # noqa: E402
from embodied_clip_amd import synthetic as syn  # noqa: E402

pytestmark = pytest.mark.gpu
G = torch.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_golden.pt"))


def _rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()


def test_rn50_against_golden():
    from embodied_clip_amd.encoder import RN50Trunk
    dev = torch.device("cuda:0")
    trunk = RN50Trunk(syn.rn50_visual_state_dict(G["rn50"]["seed_weights"]), device=dev)
    feat = trunk.forward(syn.synthetic_rgb(G["rn50"]["seed_rgb"], 2).to(dev))
    f = trunk.to_nchw_f32(feat).cpu()
    assert _rel(f[:, ::32], G["rn50"]["conv_slice"]) < 2e-2          # bf16 path vs fp32 golden
    assert _rel(trunk.spatial_mean(feat), G["rn50"]["avgpool"]) < 1e-2
    assert _rel(f.flatten(1).norm(dim=1), G["rn50"]["norm"]) < 5e-3


def test_policy_loss_grads_gae_against_golden():
    from embodied_clip_amd import ppo
    from embodied_clip_amd.policy import PolicyHandle
    dev = torch.device("cuda:0")
    psd, feat, goal, h0, masks, actions, a, b, c, d = mg.policy_case()
    T, N = feat.shape[:2]
    h = PolicyHandle()
    flat = h.flatten(psd, dev)
    rows = feat.permute(0, 1, 3, 4, 2).reshape(T * N, 49, 2048).to(torch.bfloat16).contiguous().to(dev)
    m = masks.reshape(-1).to(dev)
    ws = torch.empty(h.workspace_bytes(T, N, True), dtype=torch.uint8, device=dev)
    hv, hf = h.forward(flat, rows, goal.reshape(-1).to(dev), h0[0].contiguous().to(dev), m, T, N, ws)
    hvv = hv.view(T, N, 7)
    assert _rel(hvv[..., :6], G["policy"]["logits"]) < 2e-5
    assert _rel(hvv[..., 6:], G["policy"]["values"]) < 2e-5
    assert _rel(hf, G["policy"]["h"][0]) < 2e-5
    old_lp = G["policy"]["logits"].log_softmax(-1).gather(-1, actions.unsqueeze(-1)) + 0.2 * a
    old_v = G["policy"]["values"] + 0.2 * b
    f = lambda t: t.reshape(-1).contiguous().to(dev)
    dhv, sums = ppo.ppo_loss_raw(hv, f(actions), f(old_lp), f(old_v), f(c), f(d), 6)
    s = (sums / (T * N)).tolist()
    L = G["policy"]["loss"]
    assert abs(s[0] - L["action"]) < 1e-5 and abs(s[1] - L["value"]) < 1e-5 and abs(s[2] - L["entropy"]) < 1e-5
    grads = torch.zeros_like(flat)
    h.backward(flat, rows, m, T, N, ws, dhv, None, grads)
    gv = h.views(grads)
    for k, ref in G["policy"]["grad_norms"].items():
        assert abs(float(gv[k].norm()) - ref) < 2e-4 * ref + 1e-9, k
        assert _rel(gv[k].reshape(-1)[:4096:7], G["policy"]["grad_slices"][k]) < 3e-4, k
    Tg, Ng = 16, 4
    mm = torch.cat([torch.ones(1, Ng, 1), syn.synthetic_masks(31, Tg, Ng, 0.15)], 0)
    r = syn.synthetic_rewards(32, mm[1:])
    v = torch.from_numpy(syn.hash_normal(33, (Tg + 1) * Ng).astype("float32")).reshape(Tg + 1, Ng, 1)
    R, _, nadv = ppo.compute_returns(r.to(dev), v.to(dev), mm.to(dev))
    assert _rel(R, G["gae"]["returns"]) < 1e-6 and _rel(nadv, G["gae"]["norm_adv"]) < 1e-5


def test_text_tower_against_golden():
    """CLIP.encode_text on HIP vs the committed oracle embeddings of 12 goal-token rows (bf16 tolerance)."""
    from embodied_clip_amd.encoder import ClipTextEncoder
    gold = G["text"]["embeds"]
    sd, tok = mg.text_case()
    got = ClipTextEncoder(sd, device="cuda:0").encode_text(tok).cpu()
    rel = float((got - gold).norm() / gold.norm())
    assert rel < 2e-2, rel
    assert torch.nn.functional.cosine_similarity(got, gold).min().item() > 0.999


def test_imagenet_resnet50_against_golden():
    """The ImageNet tower (torchvision ResNet-50 minus avgpool / fc, thor_image_features.py:46-54,102-106) on HIP vs the committed
    vectors of the HuggingFace-pinned oracle (tests/golden/tvresnet_golden.pt): imagenet_conv slice, imagenet_avgpool, norms."""
    from embodied_clip_amd.encoder import ImageNetRN50Trunk
    gold = torch.load(os.path.join(os.path.dirname(__file__), "golden", "tvresnet_golden.pt"))
    dev = torch.device("cuda:0")
    trunk = ImageNetRN50Trunk(syn.tv_resnet_state_dict(gold["seed_weights"]), device=dev)
    raw = syn.synthetic_rgb_u8(gold["seed_rgb"], gold["n"])
    for feat in (trunk.forward(syn.normalize_rgb_imagenet(raw).contiguous().to(dev)), trunk.forward_u8(raw.to(dev))):
        f = trunk.to_nchw_f32(feat).cpu()
        assert _rel(f[:, ::32], gold["conv_slice"]) < 2e-2
        assert _rel(trunk.spatial_mean(feat), gold["avgpool"]) < 1e-2
        assert _rel(f.flatten(1).norm(dim=1), gold["norm"]) < 5e-3
