"""Micro-benchmark of the RN50 trunk forward (encoder-only), for kernel tuning."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from embodied_clip_amd import synthetic as syn
from embodied_clip_amd.encoder import RN50Trunk

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--chunk", type=int, default=0)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--min-tiles", type=int, default=0, help="ec_rn50_set_conv8_min_tiles (the engine sets 50 for 2 x >= 128 frames)")
a = ap.parse_args()
dev = torch.device("cuda:0")
trunk = RN50Trunk(syn.rn50_visual_state_dict(0), device=dev, chunk=a.chunk)
trunk.set_conv8_min_tiles(a.min_tiles)
rgb = syn.synthetic_rgb(1, 8).to(dev).repeat((a.batch + 7) // 8, 1, 1, 1)[:a.batch].contiguous()
out = trunk.forward(rgb)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    trunk.forward(rgb, out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.iters
fps = a.batch / ms * 1e3
print(f"batch={a.batch} chunk={a.chunk} {ms:.3f} ms/forward  {fps:.0f} frames/s  "
      f"{fps * 2 * 5.367226368e9 / 1e12:.1f} TFLOP/s (trunk 5.367 GMAC/frame)")
print('plan_hash', trunk.plan_hash())
print('num_ops', trunk.lib.ec_rn50_num_ops(trunk.h))
