"""Times only the PPO update phase (HOT LOOP B) of the engine, for profiling."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from embodied_clip_amd.engine import Worker
ap = argparse.ArgumentParser(); ap.add_argument("--actors", type=int, default=256); ap.add_argument("--streams", type=int, default=2)
ap.add_argument("--iters", type=int, default=3)
a = ap.parse_args()
w = Worker(a.actors, T=128, device="cuda:0", encoder_streams=a.streams)
# fill the rollout buffers cheaply: one real act/encode step, then replicate the features over T
w.collect_rollout(); w.compute_returns(); torch.cuda.synchronize()
w.update(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    w.update()
torch.cuda.synchronize()
print(f"update: {(time.perf_counter() - t0) / a.iters * 1e3:.1f} ms per update ({w.update_repeats} epochs), streams={a.streams}")
