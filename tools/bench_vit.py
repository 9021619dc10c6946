"""Micro-benchmark of the ViT-B/32 embedder forward (encoder-only), for kernel tuning."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from embodied_clip_amd import synthetic as syn
from embodied_clip_amd.encoder import ViTEmbedder
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=256); ap.add_argument("--iters", type=int, default=5); ap.add_argument("--min-tiles", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda:0")
vit = ViTEmbedder(syn.vit_visual_state_dict(0), device=dev)
vit.set_conv8_min_tiles(a.min_tiles)
rgb = syn.synthetic_rgb(1, 8).to(dev).repeat((a.batch + 7) // 8, 1, 1, 1)[:a.batch].contiguous()
out = vit.forward(rgb)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    vit.forward(rgb, out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.iters
print(f"batch={a.batch} {ms:.3f} ms/forward  {a.batch / ms * 1e3:.0f} frames/s  {a.batch / ms * 1e3 * 2 * 4.050683904e9 / 1e12:.1f} TFLOP/s")
print('plan_hash', vit.plan_hash())
