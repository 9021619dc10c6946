"""Times the narrow 3x3 layers (stem conv2 / conv3+pool, layer-1 conv2) through ec_conv_bf16."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from embodied_clip_amd import encoder as enc
ap = argparse.ArgumentParser(); ap.add_argument("--B", type=int, default=256); ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0"); g = torch.Generator().manual_seed(0)
for (H, Cin, Cout, pool, label) in [(112, 32, 32, 0, "stem conv2"), (112, 32, 64, 1, "stem conv3+pool"), (56, 64, 64, 0, "L1 conv2")]:
    x = torch.randn(a.B, H, H, Cin, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(Cout, 9 * Cin, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    for _ in range(3): enc.conv_bf16(x, w, b, None, ksize=3, pool=bool(pool), act=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters): enc.conv_bf16(x, w, b, None, ksize=3, pool=bool(pool), act=1)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.iters * 1e3
    Ho = H // 2 if pool else H
    fl = 2.0 * a.B * H * H * Cout * 9 * Cin; by = 2.0 * a.B * (H * H * Cin + Ho * Ho * Cout)
    print(f"{label:16s} {Cin}->{Cout} @{H} B={a.B} pool={pool}: {us:7.1f} us {fl/us/1e6:6.0f} TFLOP/s {by/us/1e3:6.0f} GB/s  dbg={os.environ.get('EC_ROWS_DBG','0')}", flush=True)
