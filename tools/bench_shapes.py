"""Micro-benchmark of the RN50 trunk's implicit-GEMM conv shapes through ec_conv_bf16, one process for all of them.

  python tools/bench_shapes.py [--B 256] [--set trunk|c3|k1|all] [--iters 20]
Prints one line per shape: time, TFLOP/s, algorithmic GB/s (input + weights + output [+ residual])."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from embodied_clip_amd import encoder as enc

# (H, Cin, Cout, ks, pool, res, label)
C3 = [
    (56, 128, 128, 3, 1, 0, "L2.0 conv2 pooled"),
    (28, 128, 128, 3, 0, 0, "L2.x conv2"),
    (28, 256, 256, 3, 1, 0, "L3.0 conv2 pooled"),
    (14, 256, 256, 3, 0, 0, "L3.x conv2"),
    (14, 512, 512, 3, 1, 0, "L4.0 conv2 pooled"),
    (7, 512, 512, 3, 0, 0, "L4.x conv2"),
]
K1 = [
    (28, 512, 256, 1, 0, 0, "L3.0 conv1"),
    (14, 512, 1024, 1, 0, 0, "L3.0 downsample"),
    (14, 256, 1024, 1, 0, 1, "L3.x conv3+res"),
    (14, 1024, 256, 1, 0, 0, "L3.x conv1"),
    (14, 1024, 512, 1, 0, 0, "L4.0 conv1"),
    (7, 1024, 2048, 1, 0, 0, "L4.0 downsample"),
    (7, 512, 2048, 1, 0, 1, "L4.x conv3+res"),
    (7, 2048, 512, 1, 0, 0, "L4.x conv1"),
]
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=256)
ap.add_argument("--set", default="all")
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
shapes = {"c3": C3, "k1": K1, "all": C3 + K1}[a.set]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
tot = 0.0
for (H, Cin, Cout, ks, pool, res, label) in shapes:
    x = torch.randn(a.B, H, H, Cin, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(Cout, ks * ks * Cin, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    Ho = H // 2 if pool else H
    r = torch.randn(a.B, Ho, Ho, Cout, generator=g).to(torch.bfloat16).to(dev) if res else None
    out = None
    for _ in range(3):
        out = enc.conv_bf16(x, w, b, r, ksize=ks, pool=bool(pool), act=1, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        enc.conv_bf16(x, w, b, r, ksize=ks, pool=bool(pool), act=1, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.iters * 1e3
    fl = 2.0 * a.B * H * H * Cout * ks * ks * Cin
    by = 2.0 * (a.B * H * H * Cin + Cout * ks * ks * Cin + a.B * Ho * Ho * Cout * (2 if res else 1))
    tot += us
    label = label + (" [ws]" if ws is not None else "")
    print(f"{label:25s} {ks}x{ks} {Cin:4d}->{Cout:4d} @{H:2d} B={a.B} pool={pool} res={res}: {us:8.1f} us {fl/us/1e6:7.0f} TFLOP/s {by/us/1e3:6.0f} GB/s", flush=True)
print(f"sum {tot:.1f} us")
