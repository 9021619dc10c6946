import sys, torch
sys.path.insert(0, "/root/repo")
from embodied_clip_amd import encoder as enc
dev = torch.device("cuda:0")
def timed(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for B in (32, 64, 128):
    for name, R, K1, K2, Co in (("layer2.0", 28, 128, 256, 512), ("layer3.0", 14, 256, 512, 1024), ("layer4.0", 7, 512, 1024, 2048)):
        c2 = torch.randn(B, R, R, K1, device=dev).to(torch.bfloat16)
        xp = torch.randn(B, R, R, K2, device=dev).to(torch.bfloat16)
        cat = torch.cat([c2, xp], -1).contiguous()
        w3 = (torch.randn(Co, K1, device=dev) / K1 ** .5).to(torch.bfloat16)
        wd = (torch.randn(Co, K2, device=dev) / K2 ** .5).to(torch.bfloat16)
        wc = torch.cat([w3, wd], 1).contiguous()
        b = torch.zeros(Co, device=dev)
        ds_out = torch.empty(B, R, R, Co, device=dev, dtype=torch.bfloat16)
        y = torch.empty_like(ds_out)
        t_ds = timed(lambda: enc.conv_bf16(xp, wd, b, act=0, out=ds_out))
        t_c3 = timed(lambda: enc.conv_bf16(c2, w3, b, res=ds_out, act=1, out=y))
        def two():
            enc.conv_bf16(xp, wd, b, act=0, out=ds_out); enc.conv_bf16(c2, w3, b, res=ds_out, act=1, out=y)
        t_two = timed(two)
        y2 = torch.empty_like(y)
        t_f = timed(lambda: enc.conv_bf16(cat, wc, b, act=1, out=y2))
        two(); enc.conv_bf16(cat, wc, b, act=1, out=y2); torch.cuda.synchronize()
        err = float((y.float() - y2.float()).abs().max() / y.float().abs().max())
        print(f"B={B:3d} {name}: ds {t_ds:6.1f} + conv3 {t_c3:6.1f} = chained {t_two:6.1f} us | K-concatenated single GEMM {t_f:6.1f} us  (saves {t_two - t_f:5.1f})  rel diff {err:.1e}")
