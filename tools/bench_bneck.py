"""Micro-benchmark + check of the fused bottleneck launch (conv_bneck.hip) against the two conv_igemm launches it replaces.

  python tools/bench_bneck.py [--B 128] [--iters 20] [--streams 1|2]
--streams 2: two launches of B frames in flight on two HIP streams (the engine's two actor slices)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
if "--stamps" in sys.argv or "--wino-emu" in sys.argv:
    import _toolslib  # noqa: F401  (the phase stamps / the Winograd feed emulation need the tools build: `make tools`)
import torch
from embodied_clip_amd import _lib as _l0
if os.environ.get("EC_AMD_LIB"):      # same-box A/B of two builds
    _l0.LIB_PATH = os.environ["EC_AMD_LIB"]
from embodied_clip_amd import encoder as enc

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=128)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--streams", type=int, default=1)
ap.add_argument("--stamps", action="store_true", help="print workgroup 0's phase stamps (shader clocks, us, implied MHz)")
ap.add_argument("--wino-emu", action="store_true", help="time (and stamp) the Winograd F(2x2,3x3) FEED EMULATION of the whole-block "
                "launch (tools build; garbage results, timing only) next to the real launches")
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
C, H = 256, 14
mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc)
sets = []
for _ in range(a.streams):
    c1 = mk(a.B, H, H, C).relu().to(torch.bfloat16).to(dev)
    x = mk(a.B, H, H, 4 * C).relu().to(torch.bfloat16).to(dev)
    w2 = mk(C, 9 * C, sc=(9 * C) ** -0.5).to(torch.bfloat16).to(dev)
    w3 = mk(4 * C, C, sc=C ** -0.5).to(torch.bfloat16).to(dev)
    b2, b3 = mk(C, sc=0.1).to(dev), mk(4 * C, sc=0.1).to(dev)
    w1 = mk(C, 4 * C, sc=(4 * C) ** -0.5).to(torch.bfloat16).to(dev)
    b1 = mk(C, sc=0.1).to(dev)
    sets.append((c1, x, w2, w3, b2, b3, torch.empty_like(x), torch.empty_like(c1), torch.empty_like(x), w1, b1, torch.empty_like(x), torch.empty_like(c1)))
c1, x, w2, w3, b2, b3, yf, c2u, yu, w1, b1, y3, c1u = sets[0]

def unfused3(s):
    c1, x, w2, w3, b2, b3, yf, c2u, yu, w1, b1, y3, c1u = s
    enc.conv_bf16(x, w1, b1, None, ksize=1, pool=False, act=1, out=c1u)
    enc.conv_bf16(c1u, w2, b2, None, ksize=3, pool=False, act=1, out=c2u)
    enc.conv_bf16(c2u, w3, b3, x, ksize=1, pool=False, act=1, out=yu)

def fused3(s):
    c1, x, w2, w3, b2, b3, yf, c2u, yu, w1, b1, y3, c1u = s
    enc.bneck_conv123_bf16(x, w1, b1, w2, b2, w3, b3, out=y3)

def wino_emu(s):
    """bneck23_kernel<.., F1, WEMU>: conv2's K loop with Winograd's operand traffic and MFMA count (timing only)."""
    import ctypes
    from embodied_clip_amd import _lib
    c1, x, w2, w3, b2, b3, yf, c2u, yu, w1, b1, y3, c1u = s
    lib = _lib.load()
    if not hasattr(wino_emu, "fn"):
        wino_emu.fn = lib.ec_bneck_wino_emu
        wino_emu.fn.restype = ctypes.c_int
        wino_emu.fn.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
        enc.bneck_conv123_bf16(x, w1, b1, w2, b2, w3, b3, out=y3)       # (packs the weights into the cache)
    packed = enc._packed_lookup("bneck3", (w1, w2, w3))
    _lib.check(wino_emu.fn(x.data_ptr(), packed.data_ptr(), b1.data_ptr(), b2.data_ptr(), b3.data_ptr(), y3.data_ptr(), x.shape[0], 14, 14, 256,
                           _lib.stream_ptr()), "ec_bneck_wino_emu")

def unfused(s):
    c1, x, w2, w3, b2, b3, yf, c2u, yu = s[:9]
    enc.conv_bf16(c1, w2, b2, None, ksize=3, pool=False, act=1, out=c2u)
    enc.conv_bf16(c2u, w3, b3, x, ksize=1, pool=False, act=1, out=yu)

def fused(s):
    c1, x, w2, w3, b2, b3, yf, c2u, yu = s[:9]
    enc.bneck_conv23_bf16(c1, w2, b2, w3, b3, x, out=yf)

unfused(sets[0]); fused(sets[0]); torch.cuda.synchronize()
d = (yf.float() - yu.float())
print(f"fused vs unfused: equal={torch.equal(yf, yu)} max|d|={d.abs().max().item():.4g} rel={d.norm().item() / yu.float().norm().item():.3g} "
      f"nonzero={(yf != 0).float().mean().item():.3f}")
unfused3(sets[0]); fused3(sets[0]); torch.cuda.synchronize()
d3 = (y3.float() - yu.float())
print(f"whole block fused vs three conv launches: equal={torch.equal(y3, yu)} max|d|={d3.abs().max().item():.4g} rel={d3.norm().item() / yu.float().norm().item():.3g}")
streams = [torch.cuda.Stream() for _ in range(a.streams)]
for st in streams:            # (first use of a stream is slow: not inside a timed region)
    with torch.cuda.stream(st):
        fused(sets[0])
torch.cuda.synchronize()
cases = [("unfused (3x3 + 1x1+res)", unfused), ("fused bneck23", fused), ("unfused (1x1, 3x3, 1x1+res)", unfused3), ("fused bneck123", fused3)]
if a.wino_emu:
    cases = [("fused bneck123", fused3), ("bneck123 Winograd FEED EMULATION", wino_emu)]
for name, fn in cases:
    for _ in range(3):
        for s in sets: fn(s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for st in streams: st.wait_stream(torch.cuda.current_stream())
    for _ in range(a.iters):
        for st, s in zip(streams, sets):
            with torch.cuda.stream(st):
                fn(s)
    for st in streams: torch.cuda.current_stream().wait_stream(st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.iters * 1e3
    fl = 2.0 * a.B * a.streams * H * H * (C * 9 * C + 4 * C * C + (4 * C * C if '1x1, 3x3' in name or '123' in name else 0))   # (direct-conv flop, also for the emulation)
    print(f"{name:26s} B={a.B} x {a.streams} stream(s): {us:8.1f} us  {fl / us / 1e6:7.0f} TFLOP/s")

if a.stamps and os.environ.get("EMPTY"):      # the launch's fixed cost: every workgroup returns at entry (tools build)
    import ctypes
    from embodied_clip_amd import _lib
    lib = _lib.load()
    lib.ec_bneck_set_mode.argtypes = [ctypes.c_int]; lib.ec_bneck_set_mode.restype = None
    lib.ec_bneck_conv123_repeat.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 5 + [ctypes.c_void_p]
    lib.ec_bneck_conv123_repeat.restype = ctypes.c_int
    c1, x, w2, w3, b2, b3, yf, c2u, yu, w1, b1, y3, c1u = sets[0]
    fused3(sets[0])
    packed = enc._packed_lookup("bneck3", (w1, w2, w3))
    def rep(n):
        _lib.check(lib.ec_bneck_conv123_repeat(x.data_ptr(), packed.data_ptr(), b1.data_ptr(), b2.data_ptr(), b3.data_ptr(), y3.data_ptr(),
                                               a.B, 14, 14, 256, n, _lib.stream_ptr()))
    for mode, name in ((1, "EMPTY (every workgroup returns at entry)"), (0, "REAL")):
        lib.ec_bneck_set_mode(mode)
        rep(5); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); rep(200); e1.record(); torch.cuda.synchronize()
        print(f"EMPTY-test {name} whole-block launch, B={a.B} ({a.B} workgroups x 512 threads, 152 KB LDS), 200 launches issued from C: "
              f"{e0.elapsed_time(e1) / 200 * 1e3:.2f} us per launch")
    lib.ec_bneck_set_mode(0)
    # device-clock view of consecutive launches: every launch stamps all its workgroups into its own block
    NL = 12
    big = torch.zeros(NL * (64 + 4 * a.B), dtype=torch.int64, device=dev)
    lib.ec_bneck_set_debug.argtypes = [ctypes.c_void_p]; lib.ec_bneck_set_debug.restype = None
    rep(20); torch.cuda.synchronize()
    lib.ec_bneck_set_debug(big.data_ptr()); rep(NL); torch.cuda.synchronize(); lib.ec_bneck_set_debug(None)
    tt = big.cpu().view(NL, 64 + 4 * a.B)[:, 64:].view(NL, a.B, 4)
    st, en = tt[:, :, 0].min(dim=1).values, tt[:, :, 1].max(dim=1).values
    for i in range(4, NL - 1):
        print(f"  launch {i}: first start -> last end {(en[i] - st[i]).item() / 100:.2f} us; gap to the next launch's first start {(st[i + 1] - en[i]).item() / 100:.2f} us; period {(st[i + 1] - st[i]).item() / 100:.2f} us")
if a.stamps:
    from embodied_clip_amd import _lib
    lib = _lib.load()
    buf = torch.zeros(64 + 4 * a.B, dtype=torch.int64, device=dev)
    lib.ec_bneck_set_debug(buf.data_ptr())
    stamped = wino_emu if a.wino_emu else (fused3 if os.environ.get("STAMP_F1") else fused)
    for _ in range(3):
        for s in sets: stamped(s)
    torch.cuda.synchronize()
    lib.ec_bneck_set_debug(None)
    t = buf.cpu().tolist()
    names = ["entry", "T loaded", "conv2 done", "c2 in T", "pass 0", "pass 1", "pass 2", "pass 3"]
    for i in range(1, 8):
        dc, dr = t[2 * i] - t[2 * i - 2], t[2 * i + 1] - t[2 * i - 1]
        print(f"  {names[i]:12s} +{dc:8d} clk  +{dr / 100.0:7.2f} us  ({dc / max(dr, 1) * 100:.0f} MHz)")
    print(f"  total {t[14] - t[0]} clk, {(t[15] - t[1]) / 100.0:.2f} us")
    if a.streams == 1 and a.B > 1:      # every workgroup's [start, end] on the 100-MHz counter (last stamped launch), by XCD
        import statistics
        wg = [(t[64 + 4 * i], t[64 + 4 * i + 1], t[64 + 4 * i + 2] & 15) for i in range(a.B)]
        t00 = min(w[0] for w in wg)
        dur = [(w[1] - w[0]) / 100.0 for w in wg]
        print(f"  all {a.B} workgroups: start spread {(max(w[0] for w in wg) - t00) / 100.0:.2f} us, duration min {min(dur):.1f} / median "
              f"{statistics.median(dur):.1f} / max {max(dur):.1f} us, last end {(max(w[1] for w in wg) - t00) / 100.0:.1f} us after the first start")
        for x in range(8):
            d = [(w[1] - w[0]) / 100.0 for w in wg if w[2] == x]
            if d:
                print(f"    XCD {x}: {len(d):3d} workgroups, duration median {statistics.median(d):6.1f} max {max(d):6.1f} us, "
                      f"first start +{(min(w[0] for w in wg if w[2] == x) - t00) / 100.0:.2f} us")
    t0 = t[6]   # (c2 in T: the shader clock at conv3's start)
    for w in (0, 1):
        row = t[16 + 16 * w: 32 + 16 * w]
        print(f"  wave {4 * w}: " + "  ".join(f"pass {q}: start {row[4 * q] - t0:6d} K-loop end {row[4 * q + 1] - t0:6d} end {row[4 * q + 2] - t0:6d}" for q in range(4)))
