"""Diagnostic for tests/test_gpu_engine.py::test_worker_iteration_matches_oracle: distribution of the per-element update differences."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from embodied_clip_amd import synthetic as syn
from oracle import ppo as oppo
from embodied_clip_amd.engine import Worker
T, N, R = 3, 2, 2
enc_sd, pol_sd = syn.rn50_visual_state_dict(0), syn.policy_state_dict(0)
w = Worker(N, T=T, device="cuda:0", seed=3, update_repeats=R, encoder_sd=enc_sd, policy_sd=pol_sd)
w.collect_rollout(); w.compute_returns(); torch.cuda.synchronize()
S, C = w.S, w.C
feat_gpu = w.feat.float().cpu().view(T + 1, N, S, S, C).permute(0, 1, 4, 2, 3).contiguous()
masks = w.env.masks.cpu().unsqueeze(-1); goals = w.env.goals.cpu(); actions = w.actions.cpu()
sd_ref = {k: v.clone() for k, v in pol_sd.items()}
batch = dict(feat=feat_gpu[:T], goal=goals[:T], h0=torch.zeros(1, N, w.H), masks=masks[:T], actions=actions,
             old_log_probs=w.logp.cpu().unsqueeze(-1), old_values=w.values[:T].cpu().unsqueeze(-1),
             returns=w.returns[:T].cpu().unsqueeze(-1), norm_adv=w.nadv.cpu().unsqueeze(-1))
st = {}
for _ in range(R):
    info, _ = oppo.ppo_update_step(sd_ref, batch, st)
w.update(); torch.cuda.synchronize()
pv = w.policy.views(w.params)
thr = 0.15 * R * 3e-4 + 1e-7
for name, pref in sd_ref.items():
    upd, upd_ref = pv[name].cpu() - pol_sd[name], pref - pol_sd[name]
    d = (upd - upd_ref).abs()
    k = int((d > thr).sum())
    idx = d.flatten().topk(min(3, d.numel())).indices
    print(f"{name:55s} n={d.numel():8d} max={d.max():.2e} over_thr={k} top: " +
          " ".join(f"({upd.flatten()[i]:+.2e} vs {upd_ref.flatten()[i]:+.2e})" for i in idx))
