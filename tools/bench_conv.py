"""Micro-benchmark of one conv layer through ec_conv_bf16 (for kernel tuning / ablation)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from embodied_clip_amd import encoder as enc
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=256); ap.add_argument("--H", type=int, default=14)
ap.add_argument("--Cin", type=int, default=256); ap.add_argument("--Cout", type=int, default=256)
ap.add_argument("--ks", type=int, default=3); ap.add_argument("--pool", type=int, default=0)
ap.add_argument("--res", type=int, default=0); ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
x = torch.randn(a.B, a.H, a.H, a.Cin, generator=g).to(torch.bfloat16).to(dev)
w = (torch.randn(a.Cout, a.ks * a.ks * a.Cin, generator=g) * 0.05).to(torch.bfloat16).to(dev)
b = torch.randn(a.Cout, generator=g).to(dev)
r = torch.randn(a.B, a.H, a.H, a.Cout, generator=g).to(torch.bfloat16).to(dev) if a.res else None
for _ in range(3):
    enc.conv_bf16(x, w, b, r, ksize=a.ks, pool=bool(a.pool), act=1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    enc.conv_bf16(x, w, b, r, ksize=a.ks, pool=bool(a.pool), act=1)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / a.iters * 1e3
fl = 2.0 * a.B * a.H * a.H * a.Cout * a.ks * a.ks * a.Cin
print(f"{a.ks}x{a.ks} {a.Cin}->{a.Cout} @{a.H} B={a.B} pool={a.pool} res={a.res}: {us:8.1f} us  {fl/us/1e6:7.0f} TFLOP/s  ablate={os.environ.get('EC_CONV_ABLATE','0')}")
