"""Times AttentionPool2d (RN50 geometry: 7x7x2048 -> 1024) alone: python tools/bench_attnpool.py --batch 128"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from embodied_clip_amd import synthetic as syn
from embodied_clip_amd.encoder import AttentionPool
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=128); ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
sd = syn.rn50_visual_state_dict(0)
pool = AttentionPool(sd, device="cuda:0")
feat = (torch.randn(a.batch, 7, 7, 2048).abs() * 0.5).to(torch.bfloat16).cuda()
out = pool.forward(feat); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters): pool.forward(feat, out=out)
torch.cuda.synchronize()
print(f"attnpool batch={a.batch}: {(time.perf_counter() - t0) / a.iters * 1e3:.3f} ms/forward")
