"""Check + micro-benchmark of the image-resident small-launch 3x3 kernel (conv_bneck.hip conv3x3_img_kernel) against ec_conv_bf16."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from embodied_clip_amd import encoder as enc
ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=30)
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (H, C, pool) in ((14, 256, False), (7, 512, False), (14, 512, True)):
    for B in (1, 5, 32, 64, 128):
        x = torch.randn(B, H, H, C, generator=g).relu().to(torch.bfloat16).to(dev)
        w = (torch.randn(C, 9 * C, generator=g) * (9 * C) ** -0.5).to(torch.bfloat16).to(dev)
        b = (torch.randn(C, generator=g) * 0.1).to(dev)
        Ho = H // 2 if pool else H
        ref = torch.empty((B, Ho, Ho, C), dtype=x.dtype, device=dev); out = torch.empty_like(ref)
        res = []
        for name, fn in (("conv_igemm", lambda: enc.conv_bf16(x, w, b, None, ksize=3, pool=pool, act=1, out=ref)),
                         ("img3x3", lambda: enc.conv3x3_img_bf16(x, w, b, out=out, pool=pool))):
            for _ in range(3): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters): fn()
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / a.iters * 1e3)
        d = out.float() - ref.float()
        tag = " pooled" if pool else ""
        print(f"{H}x{H}x{C}{tag} B={B:3d}: conv_igemm {res[0]:6.1f} us  img3x3 {res[1]:6.1f} us   rel {d.norm().item() / ref.float().norm().item():.2e} "
              f"differing {(out != ref).float().mean().item():.4f}", flush=True)
