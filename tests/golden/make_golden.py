"""Generates tests/golden/*.pt from the CPU oracle (run from the repo root: python tests/golden/make_golden.py).

The reference holds no golden vectors for this path (parity unpinned), and its third-party
dependencies cannot be imported here, so these fixtures pin the ORACLE's outputs on portable hash-seeded inputs:
(a) a change in the oracle or in the synthetic generator is caught on CPU (tests/test_golden.py),
(b) the HIP path is checked against committed numbers on the GPU box (tests/test_gpu_golden.py).
Only seeds and expected outputs are stored (weights are regenerated), so the files stay small.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from embodied_clip_amd import synthetic as syn  # noqa: E402
from oracle import clip_resnet as ocr, clip_text as otxt, clip_vit as ovit, policy as opol, ppo as oppo, probe as oprobe  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def policy_case(T=8, N=4, seed=21):
    sd = syn.policy_state_dict(seed)
    n = T * N * 2048 * 49
    feat = torch.from_numpy(syn.hash_normal(seed + 1, n).astype("float32")).abs().reshape(T, N, 2048, 7, 7)
    feat = feat.to(torch.bfloat16).float()          # bf16-representable: HIP and oracle see identical inputs
    goal = syn.synthetic_goals(seed + 2, (T, N))
    h0 = torch.from_numpy(syn.hash_normal(seed + 3, N * 512).astype("float32")).reshape(1, N, 512) * 0.5
    masks = syn.synthetic_masks(seed + 4, T, N, p_reset=0.15)
    actions = syn.synthetic_goals(seed + 5, (T, N), num_goals=6)
    u = lambda s: torch.from_numpy(syn.hash_normal(seed + s, T * N).astype("float32")).reshape(T, N, 1)
    return sd, feat, goal, h0, masks, actions, u(6), u(7), u(8), u(9)


def text_case():
    """Seeded text tower (width 512, 8 heads, ctx 77, 2 blocks, vocab 1000, out 1024) and 12 'goal' token rows."""
    sd = syn.text_state_dict(21, width=512, layers=2, heads=8, context_length=77, vocab_size=1000, embed_dim=1024)
    return sd, syn.synthetic_tokens(22, 12, 77, 1000)


def zeroshot_case(T=6, N=5, E=1024, H=512, seed=61):
    """Zero-shot dual-encoder policy (fusion=1): seeded image embeddings, goal table, GRU + heads, loss inputs."""
    sd = syn.policy_state_dict(seed, in_channels=E, spatial=1, hidden=H, fusion=1)
    f = lambda s, n: torch.from_numpy(syn.hash_normal(seed + s, n).astype("float32"))
    emb = f(1, T * N * E).reshape(T, N, E) * 0.7
    table = torch.nn.functional.normalize(f(2, 12 * E).reshape(12, E), dim=-1)
    goal = syn.synthetic_goals(seed + 3, (T, N))
    h0 = f(4, N * H).reshape(1, N, H) * 0.5
    masks = syn.synthetic_masks(seed + 5, T, N, p_reset=0.2)
    actions = syn.synthetic_goals(seed + 6, (T, N), num_goals=6)
    u = lambda s: f(s, T * N).reshape(T, N, 1)
    return sd, emb, table, goal, h0, masks, actions, u(7), u(8), u(9), u(10)


def zeroshot_golden():
    sd, emb, table, goal, h0, masks, actions, a, b, c, d = zeroshot_case()
    with torch.no_grad():
        lg, vv, hT = opol.zeroshot_actor_critic_forward(emb, goal, h0, masks, sd, table)
        old_lp = opol.categorical_log_prob(lg, actions).unsqueeze(-1) + 0.2 * a
        old_v = vv + 0.2 * b
    names = [k for k, v in sd.items() if v.numel()]
    leaves = {k: (v.clone().requires_grad_(True) if v.numel() else v) for k, v in sd.items()}
    lg2, vv2, _ = opol.zeroshot_actor_critic_forward(emb, goal, h0, masks, leaves, table)
    total, info = oppo.ppo_loss(lg2, vv2, actions, old_lp, old_v, c, d)
    total.backward()
    return {"seed": 61, "logits": lg.clone(), "values": vv.clone(), "h": hT.clone(), "loss": info,
            "grad_norms": {k: float(leaves[k].grad.norm()) for k in names},
            "grad_slices": {k: leaves[k].grad.reshape(-1)[:4096:7].clone() for k in names}}


def probe_cases():
    """Seeded (x, y, weight, bias) for the four probe tasks (clip_avgpool / clip_conv embeddings)."""
    cases = {}
    x = torch.from_numpy(syn.hash_normal(41, 8 * 2048).astype("float32")).reshape(8, 2048)
    xc = torch.from_numpy(syn.hash_normal(44, 8 * 2048 * 49).astype("float32")).reshape(8, 2048, 7, 7)
    for task, odim in (("object_presence", 52), ("free_space", 11), ("reachability", 110), ("object_localization", 52)):
        w = torch.from_numpy(syn.hash_normal(42, odim * 2048).astype("float32")).reshape(odim, 2048) * 0.02
        bb = torch.zeros(odim)
        if task == "object_presence":
            y = syn.synthetic_goals(43, (8, odim), 2)
        elif task == "free_space":
            y = syn.synthetic_goals(43, (8,), 14)
        elif task == "reachability":
            y = (syn.synthetic_goals(45, (8,), 110), syn.synthetic_goals(46, (8,), 2))
        else:
            y = syn.synthetic_goals(47, (8, 9, odim), 2)
        cases[task] = (xc if task == "object_localization" else x, y, w, bb)
    return cases


def main():
    g = {}
    # (1) RN50 trunk on 2 frames: pooled embedding + a slice of the conv features (fp32 oracle)
    sd = syn.rn50_visual_state_dict(0)
    rgb = syn.synthetic_rgb(1000, 2)
    f = ocr.clip_resnet_preprocessor(rgb, sd)
    g["rn50"] = {"seed_weights": 0, "seed_rgb": 1000, "avgpool": ocr.avgpool_head(f).clone(),
                 "conv_slice": f[:, ::32].clone(), "attnpool": ocr.attnpool(f, sd).clone(),
                 "norm": f.flatten(1).norm(dim=1).clone()}
    # (2) ViT-B/32 embedder: CLS token + 8 patch tokens
    vsd = syn.vit_visual_state_dict(0)
    tok = ovit.clip_vit_preprocessor(rgb, vsd)
    g["vit"] = {"seed_weights": 0, "seed_rgb": 1000, "tokens": tok[:, :9].clone(), "norm": tok.flatten(1).norm(dim=1).clone()}
    # (2b) CLIP text tower (goal embeddings of the zero-shot variant): RN50-CLIP geometry, reduced depth / vocabulary
    tsd, ttok = text_case()
    g["text"] = {"embeds": otxt.encode_text(ttok, tsd, heads=8).clone()}
    # (3) policy forward + PPO loss + gradients for a T=8, N=4 minibatch
    psd, feat, goal, h0, masks, actions, a, b, c, d = policy_case()
    with torch.no_grad():
        lg, vv, hT = opol.actor_critic_forward(feat, goal, h0, masks, psd)
        old_lp = opol.categorical_log_prob(lg, actions).unsqueeze(-1) + 0.2 * a
        old_v = vv + 0.2 * b
    leaves = {k: v.clone().requires_grad_(True) for k, v in psd.items()}
    lg2, vv2, _ = opol.actor_critic_forward(feat, goal, h0, masks, leaves)
    total, info = oppo.ppo_loss(lg2, vv2, actions, old_lp, old_v, c, d)
    total.backward()
    g["policy"] = {"seed": 21, "logits": lg.clone(), "values": vv.clone(), "h": hT.clone(), "loss": info,
                   "grad_norms": {k: float(v.grad.norm()) for k, v in leaves.items()},
                   "grad_slices": {k: v.grad.reshape(-1)[:4096:7].clone() for k, v in leaves.items()}}
    # (4) GAE returns for T=16, N=4 with random masks
    T, N = 16, 4
    m = torch.cat([torch.ones(1, N, 1), syn.synthetic_masks(31, T, N, 0.15)], 0)
    r = syn.synthetic_rewards(32, m[1:])
    v = torch.from_numpy(syn.hash_normal(33, (T + 1) * N).astype("float32")).reshape(T + 1, N, 1)
    R = oppo.compute_returns(r, v, m)
    adv, nadv = oppo.normalized_advantages(R, v)
    g["gae"] = {"seeds": (31, 32, 33), "returns": R.clone(), "norm_adv": nadv.clone()}
    # (5) linear-probe losses for the 4 tasks of primitive_probing/train.py (incl. the double softmax)
    g["probe"] = {task: float(oprobe.compute_loss(x, y, w, bb, task)) for task, (x, y, w, bb) in probe_cases().items()}
    torch.save(g, os.path.join(OUT, "oracle_golden.pt"))
    print("wrote", os.path.join(OUT, "oracle_golden.pt"), os.path.getsize(os.path.join(OUT, "oracle_golden.pt")), "bytes")


def main_zeroshot():
    path = os.path.join(OUT, "zeroshot_golden.pt")
    torch.save(zeroshot_golden(), path)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "zeroshot":      # added in round 2; oracle_golden.pt is left untouched
        main_zeroshot()
    else:
        main()
