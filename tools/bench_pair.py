"""Micro-benchmark: fused layer-1 block boundary (ec_conv1x1_pair_bf16) vs the separate conv launches it replaces."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from embodied_clip_amd import encoder as enc
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=256); ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
M = a.B * 56 * 56
g = torch.Generator().manual_seed(0)
bf = lambda t: t.to(torch.bfloat16).to(dev)
a0, a1 = bf(torch.randn(M, 64, generator=g)), bf(torch.randn(M, 64, generator=g))
w0, w1 = bf(torch.randn(256, 64, generator=g) * 0.1), bf(torch.randn(256, 64, generator=g) * 0.1)
b0, b1 = torch.randn(256, generator=g).to(dev), torch.randn(256, generator=g).to(dev)
r = bf(torch.randn(M, 256, generator=g))
def timeit(f):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters * 1e3
for N2 in (64, 128):
    w2 = bf(torch.randn(N2, 256, generator=g) * 0.05); b2 = torch.randn(N2, generator=g).to(dev)
    def unf_res():
        y = enc.gemm_bf16(a0, w0, b0, res=r, act=1); return enc.gemm_bf16(y, w2, b2, act=1)
    def unf_two():
        d = enc.gemm_bf16(a1, w1, b1, act=0); y = enc.gemm_bf16(a0, w0, b0, res=d, act=1); return enc.gemm_bf16(y, w2, b2, act=1)
    t_res_f = timeit(lambda: enc.conv1x1_pair_bf16(a0, w0, b0, w2, b2, res=r))
    t_res_u = timeit(unf_res)
    gb = M * (64 + 256 + 256 + N2) * 2 / 1e9
    print(f"N2={N2} residual : fused {t_res_f:7.1f} us ({gb / t_res_f * 1e3:5.2f} TB/s)   unfused {t_res_u:7.1f} us")
    if N2 == 64:
        t_two_f = timeit(lambda: enc.conv1x1_pair_bf16(a0, w0, b0, w2, b2, a1=a1, w1=w1, b1=b1))
        t_two_u = timeit(unf_two)
        gb = M * (64 + 64 + 256 + N2) * 2 / 1e9
        print(f"N2={N2} downsample: fused {t_two_f:7.1f} us ({gb / t_two_f * 1e3:5.2f} TB/s)   unfused {t_two_u:7.1f} us")

# layer-2 geometry (28x28): conv3 128->512 + residual, next conv1 512->128
M2 = a.B * 28 * 28
a2 = bf(torch.randn(M2, 128, generator=g)); r2 = bf(torch.randn(M2, 512, generator=g))
w02 = bf(torch.randn(512, 128, generator=g) * 0.1); w22 = bf(torch.randn(128, 512, generator=g) * 0.05)
b02 = torch.randn(512, generator=g).to(dev); b22 = torch.randn(128, generator=g).to(dev)
def unf2():
    y = enc.gemm_bf16(a2, w02, b02, res=r2, act=1); return enc.gemm_bf16(y, w22, b22, act=1)
tf = timeit(lambda: enc.conv1x1_pair_bf16(a2, w02, b02, w22, b22, res=r2)); tu = timeit(unf2)
gb = M2 * (128 + 512 + 512 + 128) * 2 / 1e9
print(f"layer-2 boundary : fused {tf:7.1f} us ({gb / tf * 1e3:5.2f} TB/s)   unfused {tu:7.1f} us")
