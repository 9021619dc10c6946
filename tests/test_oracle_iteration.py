"""CPU: the oracle's full iteration runs and learns something sensible on a tiny instance."""
import torch

from embodied_clip_amd import synthetic as syn
from oracle import iteration as oit


def test_oracle_iteration_tiny():
    enc_sd = syn.rn50_visual_state_dict(0, width=64, layers=(1, 1, 1, 1), input_resolution=64)
    T, N = 3, 2
    pol_sd = syn.policy_state_dict(0, in_channels=2048, spatial=2, hidden=32)
    frames = syn.synthetic_rgb(5, N, 64).unsqueeze(0)
    masks = torch.cat([torch.ones(1, N, 1), syn.synthetic_masks(6, T, N, 0.3)], 0)
    goals = syn.synthetic_goals(7, (T + 1, N))
    rewards = syn.synthetic_rewards(8, masks[1:])
    before = {k: v.clone() for k, v in pol_sd.items()}
    info = oit.run_iteration(enc_sd, pol_sd, frames, goals, masks, rewards, T, N, update_repeats=2)
    assert info["frames"] == T * N and info["frames_per_s"] > 0
    assert all(torch.isfinite(v).all() for v in pol_sd.values())
    assert any(not torch.equal(before[k], pol_sd[k]) for k in pol_sd)
    assert abs(info["ratio_mean"] - 1.0) < 0.2
