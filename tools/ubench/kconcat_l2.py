"""Would layer2.0 gain from the K-concatenated conv3 | downsample GEMM?  There the chained plan's conv3 is already fused with
the NEXT block's conv1 (conv1x1_pair512_kernel), so folding the downsample conv in costs that fusion:
    now:  downsample conv (register-weight kernel) -> [conv3 + identity + ReLU -> next conv1] in one launch
    then: [c2 | xp] GEMM (K = 384) -> next conv1 as its own launch
(tools only; python tools/ubench/kconcat_l2.py on the GPU box)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from embodied_clip_amd import encoder as enc
dev = torch.device("cuda:0")


def timed(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for B in (32, 64, 128):
    R, K1, K2, Co = 28, 128, 256, 512
    M = B * R * R
    r = lambda *s: torch.randn(*s, device=dev)
    c2 = r(B, R, R, K1).relu().to(torch.bfloat16); xp = r(B, R, R, K2).relu().to(torch.bfloat16)
    cat = torch.cat([c2, xp], -1).contiguous()
    w3 = (r(Co, K1) / K1 ** .5).to(torch.bfloat16); wd = (r(Co, K2) / K2 ** .5).to(torch.bfloat16)
    wc = torch.cat([w3, wd], 1).contiguous()
    w1n = (r(K1, Co) / Co ** .5).to(torch.bfloat16)
    b = torch.zeros(Co, device=dev); b1 = torch.zeros(K1, device=dev)
    ds = torch.empty(B, R, R, Co, device=dev, dtype=torch.bfloat16)

    def now():
        enc.conv_bf16(xp, wd, b, act=0, out=ds)
        return enc.conv1x1_pair_bf16(c2.view(M, K1), w3, b, w1n, b1, res=ds.view(M, Co))

    y = torch.empty(B, R, R, Co, device=dev, dtype=torch.bfloat16)

    def then():
        enc.conv_bf16(cat, wc, b, act=1, out=y)
        return enc.conv_bf16(y, w1n, b1, act=1)

    t_now, t_then = timed(now), timed(then)
    t_ds = timed(lambda: enc.conv_bf16(xp, wd, b, act=0, out=ds))
    t_cat = timed(lambda: enc.conv_bf16(cat, wc, b, act=1, out=y))
    print(f"B={B:3d} layer2.0: downsample {t_ds:5.1f} + pair(conv3 + identity -> next conv1) {t_now - t_ds:5.1f} = {t_now:6.1f} us | "
          f"K-concatenated GEMM {t_cat:5.1f} + next conv1 {t_then - t_cat:5.1f} = {t_then:6.1f} us  (saves {t_now - t_then:5.1f})")
