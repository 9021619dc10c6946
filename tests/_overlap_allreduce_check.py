"""Run as a subprocess by tests/test_gpu_multi.py: one rank over RCCL (world size 1, so the SUM is an identity) -- two workers
from the same seed, one reducing the GRU + heads section of the gradient bucket on the communication stream under the goal
encoder's backward, one with the single all-reduce after the backward, must stay the same worker."""
import json
import os
import sys
import tempfile

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from embodied_clip_amd.engine import Worker
    n_actors, slices = int(sys.argv[1]), int(sys.argv[2])
    store = tempfile.mktemp(prefix="ec_overlap_")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"file://{store}", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    out = {}
    ws = {}
    for name, ov in (("overlap", True), ("single", False), ("none", None)):
        w = Worker(n_actors, T=8, device="cuda:0", seed=3, encoder_streams=slices, force_allreduce=ov is not None,
                   overlap_allreduce=bool(ov))
        # ONE iteration: the rollout precedes every update, so the three workers collect the same rollout bit for bit and
        # differ only by what the update does (with a second iteration, a rounding difference in the parameters can flip a
        # sampled action and the trajectories part ways: seen, 3e-5 in the losses)
        w.iteration()
        torch.cuda.synchronize()
        ws[name] = w.params.clone()
        out[name + "_loss"] = w.loss_info()
    # (since round 6 the backward holds no floating-point atomics: the three workers' gradients are the same sums in the same order)
    out["max_abs_overlap_vs_single"] = float((ws["overlap"] - ws["single"]).abs().max())
    out["max_abs_overlap_vs_none"] = float((ws["overlap"] - ws["none"]).abs().max())
    out["moved"] = float((ws["overlap"] - Worker(n_actors, T=8, device="cuda:0", seed=3, encoder_streams=slices).params).abs().max())
    dist.destroy_process_group()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
