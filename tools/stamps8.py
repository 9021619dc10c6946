"""Per-segment s_memtime stamps of the 8-wave conv kernel (block 0, waves 0 and 4); EC_CONV_ABLATE=32 EC_CONV_BIG=1."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["EC_CONV_BIG"] = "1"; os.environ["EC_CONV_ABLATE"] = str(32 | int(os.environ.get("ABL", "0")))
import _toolslib  # noqa: F401  (selects the tools build)
import torch
from embodied_clip_amd import encoder as enc, _lib
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B = int(os.environ.get("B", "334"))
x = torch.randn(B, 14, 14, 256, generator=g).to(torch.bfloat16).to(dev)
w = (torch.randn(256, 9 * 256, generator=g) * 0.05).to(torch.bfloat16).to(dev)
b = torch.randn(256, generator=g).to(dev)
for _ in range(3):
    enc.conv_bf16(x, w, b, None, ksize=3, pool=False, act=1)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 2048)()
_lib.check(_lib.load().ec_debug_stamps(buf, 2048))
for grp in (0, 1):
    st = [buf[grp * 1024 + i] for i in range(256)]
    st = [v for v in st if v]
    d = [st[i + 1] - st[i] for i in range(len(st) - 1)]
    # stamps alternate: before barrier, after barrier -> d[even] = barrier wait, d[odd] = segment work
    print(f"group {grp}: {len(st)} stamps; K-tiles 8..11 (work, wait) per segment MEM0 CMP0 MEM1 CMP1:")
    for kt in range(8, 12):
        row = []
        for sgm in range(4):
            i = (kt * 4 + sgm) * 2
            work = d[i - 1] if i > 0 else 0
            wait = d[i]
            row.append(f"{work:5d}+{wait:4d}")
        print("   kt", kt, "  ".join(row))
    print("   mean cycles per K-tile:", (st[-1] - st[0]) / (len(st) / 8))
    t_entry, t_loop_end, t_end = (buf[grp * 1024 + k] for k in (300, 301, 302))
    print(f"   phases (clk): entry -> first loop barrier {st[0] - t_entry}, K loop {t_loop_end - st[0]}, epilogue {t_end - t_loop_end}, total {t_end - t_entry}")
