"""Micro-benchmark of one policy act step (T=1) for a slice of actors, alone on the GPU (what is exposed in the
rollout while the other slice's encoder runs)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from embodied_clip_amd.engine import Worker
ap = argparse.ArgumentParser(); ap.add_argument("actors_pos", nargs="?", type=int); ap.add_argument("--actors", type=int, default=256); ap.add_argument("--iters", type=int, default=50)
a = ap.parse_args(); a.actors = a.actors_pos or a.actors
w = Worker(a.actors, T=4, device="cuda:0", encoder_streams=2)
w.collect_rollout(); torch.cuda.synchronize()
sl = w.slices[0]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3): w._act_slice(sl, 0)
torch.cuda.synchronize()
e0.record()
for _ in range(a.iters): w._act_slice(sl, 0)
e1.record(); torch.cuda.synchronize()
print(f"act step, {sl.n} actors: {e0.elapsed_time(e1) / a.iters * 1e3:.1f} us")
