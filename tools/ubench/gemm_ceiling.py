"""Yardstick, not product: what the vendor GEMM library (rocBLAS / hipBLASLt behind ``torch.matmul``) sustains in bf16 on THIS
box -- on a large square problem (the practical MFMA ceiling under the chip's power limit) and on the GEMM shapes of the
trunk's convolutions at a 128-frame launch (M = frames x Ho x Wo, N = Cout, K = k*k*Cin; a 3x3 convolution is handed over as the
plain [M, 9*Cin] GEMM, i.e. WITHOUT the im2col traffic a library convolution would add).  The build's own kernels are timed on
the same shapes beside it (implicit GEMM on the NHWC map, bias + ReLU fused).

    python tools/ubench/gemm_ceiling.py [frames per launch, default 128]           # on the GPU box
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def timed(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    from embodied_clip_amd import encoder as enc
    dev = torch.device("cuda:0")
    print(f"# {torch.cuda.get_device_name(0)}; torch {torch.__version__}; bf16, fp32 accumulate; TFLOP/s = 2*M*N*K / time")
    for n in (4096, 8192, 16384) if len(sys.argv) < 2 else ():
        a = torch.randn(n, n, device=dev, dtype=torch.bfloat16)
        b = torch.randn(n, n, device=dev, dtype=torch.bfloat16)
        t = timed(lambda: torch.matmul(a, b), iters=10 if n > 8192 else 30)
        print(f"library GEMM {n}^3: {t * 1e6:9.1f} us  {2 * n ** 3 / t / 1e12:7.1f} TFLOP/s  ({2 * n ** 3 / t / 2.5e15:.3f} of 2.5 PFLOP/s)")
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    shapes = [  # (name, H, W, Cin, Cout, ksize)
        ("layer2 3x3 128->128 @28", 28, 28, 128, 128, 3),
        ("layer3 1x1 1024->256 @14", 14, 14, 1024, 256, 1),
        ("layer3 3x3 256->256 @14", 14, 14, 256, 256, 3),
        ("layer3 1x1 256->1024 @14", 14, 14, 256, 1024, 1),
        ("layer4 1x1 2048->512 @7", 7, 7, 2048, 512, 1),
        ("layer4 3x3 512->512 @7", 7, 7, 512, 512, 3),
        ("layer4 1x1 512->2048 @7", 7, 7, 512, 2048, 1),
    ]
    print(f"# trunk shapes at a {B}-frame launch: library GEMM on [M, K] x [K, N] against this build's conv_bf16 (implicit GEMM, bias + ReLU fused)")
    for name, H, W, Ci, Co, k in shapes:
        M, K = B * H * W, k * k * Ci
        a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        b = torch.randn(K, Co, device=dev, dtype=torch.bfloat16)
        t_lib = timed(lambda: torch.matmul(a, b), iters=30)
        x = torch.randn(B, H, W, Ci, device=dev, dtype=torch.bfloat16)
        w = (torch.randn(Co, K, device=dev, dtype=torch.float32) / K ** 0.5).to(torch.bfloat16)
        bias = torch.zeros(Co, device=dev, dtype=torch.float32)
        t_own = timed(lambda: enc.conv_bf16(x, w, bias, ksize=k, act=1), iters=30)
        fl = 2.0 * M * Co * K
        print(f"{name:28s} M={M:6d} N={Co:4d} K={K:4d}: library {t_lib * 1e6:7.1f} us {fl / t_lib / 1e12:6.1f} TFLOP/s | "
              f"conv_bf16 {t_own * 1e6:7.1f} us {fl / t_own / 1e12:6.1f} TFLOP/s  ({t_lib / t_own:4.2f}x)")


if __name__ == "__main__":
    main()
