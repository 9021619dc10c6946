"""Micro-benchmark + check of the band-fused layer-2 bottleneck (bneck_band_kernel) against the three conv launches it replaces.

  python tools/bench_band.py [--B 128] [--iters 20]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from embodied_clip_amd import encoder as enc

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=128)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
C, H = 128, 28
mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc)
x = mk(a.B, H, H, 4 * C).relu().to(torch.bfloat16).to(dev)
w1 = mk(C, 4 * C, sc=(4 * C) ** -0.5).to(torch.bfloat16).to(dev)
w2 = mk(C, 9 * C, sc=(9 * C) ** -0.5).to(torch.bfloat16).to(dev)
w3 = mk(4 * C, C, sc=C ** -0.5).to(torch.bfloat16).to(dev)
b1, b2, b3 = mk(C, sc=0.1).to(dev), mk(C, sc=0.1).to(dev), mk(4 * C, sc=0.1).to(dev)
c1u = torch.empty(a.B, H, H, C, dtype=torch.bfloat16, device=dev)
c2u = torch.empty_like(c1u)
yu, yf = torch.empty_like(x), torch.empty_like(x)

def unfused():
    enc.conv_bf16(x, w1, b1, None, ksize=1, pool=False, act=1, out=c1u)
    enc.conv_bf16(c1u, w2, b2, None, ksize=3, pool=False, act=1, out=c2u)
    enc.conv_bf16(c2u, w3, b3, x, ksize=1, pool=False, act=1, out=yu)

def fused():
    enc.bneck_band_bf16(x, w1, b1, w2, b2, w3, b3, out=yf)

unfused(); fused(); torch.cuda.synchronize()
d = yf.float() - yu.float()
print(f"band-fused vs three conv launches: equal={torch.equal(yf, yu)} max|d|={d.abs().max().item():.4g} "
      f"rel={d.norm().item() / yu.float().norm().item():.3g} nonzero={(yf != 0).float().mean().item():.3f}")
if not torch.equal(yf, yu):
    bad = (yf != yu).nonzero()
    print("first mismatches (b, y, x, c):", bad[:8].tolist(), " count", bad.shape[0])
    per_row = (yf != yu).any(-1).any(0).float()
    print("mismatching pixels map:\n", per_row.int())
for name, fn in (("three conv launches", unfused), ("band-fused", fused)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.iters * 1e3
    fl = 2.0 * a.B * H * H * (C * 9 * C + 8 * C * C)
    by = a.B * H * H * 4 * C * 2 * 2
    print(f"{name:22s} B={a.B}: {us:8.1f} us  {fl / us / 1e6:7.0f} TFLOP/s  {by / us / 1e6:6.2f} TB/s (x in + y out)")

from embodied_clip_amd import _lib
lib = _lib.load()
buf = torch.zeros(16, dtype=torch.int64, device=dev)
lib.ec_bneck_set_debug(buf.data_ptr())
for _ in range(3): fused()
torch.cuda.synchronize()
lib.ec_bneck_set_debug(None)
t = buf.cpu().tolist()
names = ["entry", "conv1 -> T", "conv2 done", "c2 in T", "pass 0", "pass 1", "pass 2", "pass 3"]
for i in range(1, 8):
    dc, dr = t[2 * i] - t[2 * i - 2], t[2 * i + 1] - t[2 * i - 1]
    print(f"  {names[i]:12s} +{dc:8d} clk  +{dr / 100.0:7.2f} us  ({dc / max(dr, 1) * 100:.0f} MHz)")
print(f"  total {t[14] - t[0]} clk, {(t[15] - t[1]) / 100.0:.2f} us")
