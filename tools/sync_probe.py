"""What made bench.py's `sync_actions` leg read 0.77 x for three rounds: every SECOND two-slice worker of a process ran at
48-51 k instead of 61-63 k env-frames/s, action-synchronous or not, with byte-identical buffer addresses -- its two slice
streams had been bound to ONE hardware queue (the HIP runtime binds a stream at its first submission), so the two encoder
launches of an env step ran one after the other.  Modes (python tools/sync_probe.py <mode>, one MI355X):
  fresh / second / twice / long / second_keep / second_free / tiny_first / *_1stream   throughput of the n-th worker of a process
  streamsN        N extra live streams in front of a fresh worker (no effect)
  layout_*        buffer addresses of fast and slow workers (identical: not a layout effect)
  concurrency     both-busy / alone time of a worker's slice streams, before and after its run
  workers         six workers in a row through _lib.concurrent_streams (the fix): all at the first worker's rate
  artifact        the same six workers with plain torch.cuda.Stream() slice streams (the behaviour before the fix)
  sync_h2d        the engine with uint8 frames crossing PCIe every step, free-running and action-synchronous (the plugin route's peer)
(the modes from `second` to `layout_second` reproduce the slow worker only with Worker's streams created by plain
torch.cuda.Stream(), i.e. before the fix.)"""
import sys, time, gc, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embodied_clip_amd.engine import Worker
mode = sys.argv[1]
def run(w, n=3, warm=1):
    for _ in range(warm): w.iteration()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): w.iteration()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return 256 * 128 * n / dt
kw = dict(T=128, device="cuda:0", seed=0)
if mode == "fresh":
    print("fresh sync worker", round(run(Worker(256, sync_actions=True, **kw))))
elif mode == "second":
    a = Worker(256, **kw); print("first (free-running)", round(run(a)))
    del a; gc.collect(); torch.cuda.empty_cache()
    print("second: sync worker", round(run(Worker(256, sync_actions=True, **kw))))
elif mode == "second_keep":
    a = Worker(64, **kw); print("first (free-running, 64 actors, kept alive)", round(run(a) / 4))
    print("second: sync worker", round(run(Worker(256, sync_actions=True, **kw))))
elif mode == "twice":
    b = Worker(256, sync_actions=True, **kw); print("sync worker", round(run(b)))
    del b; gc.collect(); torch.cuda.empty_cache()
    print("sync worker again", round(run(Worker(256, sync_actions=True, **kw))))
elif mode == "long":
    b = Worker(256, sync_actions=True, **kw)
    for i in range(4): print("sync worker, 2 more iterations", round(run(b, 2, 0)))
elif mode.startswith("streams"):
    n = int(mode[7:])
    keep = [torch.cuda.Stream() for _ in range(n)]
    for st in keep:
        with torch.cuda.stream(st):
            torch.zeros(8, device="cuda:0").add_(1)
    torch.cuda.synchronize()
    print(f"{n} extra streams alive, then sync worker", round(run(Worker(256, sync_actions=True, **kw))))
elif mode == "tiny_first":
    a = Worker(8, **dict(kw, T=4)); a.iteration(); torch.cuda.synchronize()
    print("tiny free-running worker first (kept), then sync worker", round(run(Worker(256, sync_actions=True, **kw))))
elif mode == "second_1stream":
    a = Worker(256, **kw); run(a, 1, 1); del a; gc.collect(); torch.cuda.empty_cache()
    print("second: sync worker, ONE encoder stream", round(run(Worker(256, sync_actions=True, encoder_streams=1, **kw))))
    print("fresh-equivalent reference: see `fresh` with encoder_streams=1")
elif mode == "fresh_1stream":
    print("fresh sync worker, ONE encoder stream", round(run(Worker(256, sync_actions=True, encoder_streams=1, **kw))))
elif mode == "second_free":
    a = Worker(256, sync_actions=True, **kw); print("first (sync)", round(run(a)))
    del a; gc.collect(); torch.cuda.empty_cache()
    print("second: free-running worker", round(run(Worker(256, **kw))))
elif mode.startswith("layout"):
    # where do the two slices' buffers sit?  (fresh worker vs second worker; optional dummy allocation in front)
    def ptrs(w):
        out = []
        for sl in w.slices:
            ws = sl.enc._ws
            out.append((ws.data_ptr() if ws is not None else 0, sl.feat.data_ptr(), w.env.frames.data_ptr()))
        return out
    def show(tag, w):
        r = run(w)
        p = ptrs(w)
        d_ws, d_feat = p[1][0] - p[0][0], p[1][1] - p[0][1]
        print(f"{tag}: {round(r)} env-frames/s; workspace ptrs {p[0][0]:#x} {p[1][0]:#x} (delta {d_ws / 2**20:.1f} MiB, mod 1 GiB {p[0][0] % 2**30 / 2**20:.0f} / {p[1][0] % 2**30 / 2**20:.0f} MiB); "
              f"feat ptrs {p[0][1]:#x} {p[1][1]:#x} (delta {d_feat / 2**20:.1f} MiB)")
    if mode == "layout_fresh":
        show("fresh", Worker(256, **kw))
    elif mode == "layout_second":
        a = Worker(256, **kw); show("first", a); del a; gc.collect(); torch.cuda.empty_cache()
        b = Worker(256, **kw); show("second", b); del b; gc.collect(); torch.cuda.empty_cache()
        show("third", Worker(256, **kw))
    else:
        mb = int(mode[6:])
        dummy = torch.empty(mb << 20, dtype=torch.uint8, device="cuda:0")
        show(f"fresh behind a {mb}-MiB allocation", Worker(256, **kw))
elif mode == "concurrency":
    def overlap(s0, s1, cycles=2000000):
        """wall time of one spin kernel on each stream, in units of one spin kernel alone: ~1 = concurrent, ~2 = serialized"""
        rs = []
        for rep in range(6):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            torch.cuda.synchronize()
            with torch.cuda.stream(s0):
                e[0].record(); torch.cuda._sleep(cycles); e[1].record()
            torch.cuda.synchronize()
            alone = e[0].elapsed_time(e[1])
            cur = torch.cuda.current_stream()
            e[2].record(cur)
            s0.wait_event(e[2]); s1.wait_event(e[2])
            with torch.cuda.stream(s0):
                torch.cuda._sleep(cycles)
            with torch.cuda.stream(s1):
                torch.cuda._sleep(cycles)
            cur.wait_stream(s0); cur.wait_stream(s1)
            e[3].record(cur)
            torch.cuda.synchronize()
            if rep >= 2:
                rs.append(e[2].elapsed_time(e[3]) / alone)
        return sorted(rs)[len(rs) // 2], alone
    for i in range(6):
        w = Worker(256, **kw)
        s0, s1 = w.slices[0].stream, w.slices[1].stream
        before = overlap(s0, s1)
        r = run(w)
        after = overlap(s0, s1)
        print(f"worker {i + 1}: {round(r)} env-frames/s; two slice streams both busy / one alone: {before[0]:.2f} before, {after[0]:.2f} after the run (one spin kernel {after[1]:.2f} ms); "
              f"slice 0 vs comm stream {overlap(s0, w.comm_stream)[0]:.2f}")
        del w, s0, s1; gc.collect(); torch.cuda.empty_cache()
elif mode == "workers":
    lib = __import__("embodied_clip_amd._lib", fromlist=["x"])
    import ctypes as C
    for i in range(6):
        w = Worker(256, sync_actions=(i % 2 == 1), **kw)
        r = run(w)
        ratio = C.c_float()
        lib.load().ec_stream_pair_overlap(w.slices[0].stream.cuda_stream, w.slices[1].stream.cuda_stream, 200, C.byref(ratio))
        print(f"worker {i + 1} ({'action-synchronous' if i % 2 else 'free-running'}): {round(r)} env-frames/s; slice streams both busy / alone = {ratio.value:.2f}")
        del w; gc.collect(); torch.cuda.empty_cache()
elif mode == "sync_h2d":
    # the plugin route's true peer: action-synchronous AND the frames crossing PCIe every step (raw uint8, pinned host memory)
    for sync in (False, True):
        w = Worker(256, sync_actions=sync, frames_host=True, frames_u8=True, **kw)
        print(f"frames on the host (uint8), {'action-synchronous' if sync else 'free-running'}: {round(run(w))} env-frames/s")
        del w; gc.collect(); torch.cuda.empty_cache()
elif mode == "artifact":
    # the behaviour before the fix, reproduced: Worker's streams as plain torch.cuda.Stream() objects (first bound by the worker's
    # own first submissions), six 256-actor workers in one process
    import ctypes as C
    from embodied_clip_amd import _lib as L
    real = L.concurrent_streams
    L.concurrent_streams = lambda n, device, **k: [torch.cuda.Stream(device=device) for _ in range(n)]
    for i in range(6):
        w = Worker(256, sync_actions=(i % 2 == 1), **kw)
        r = run(w)
        ratio = C.c_float()
        L.load().ec_stream_pair_overlap(w.slices[0].stream.cuda_stream, w.slices[1].stream.cuda_stream, 200, C.byref(ratio))
        print(f"plain streams, worker {i + 1} ({'action-synchronous' if i % 2 else 'free-running'}): {round(r)} env-frames/s; slice streams both busy / alone = {ratio.value:.2f}")
        del w; gc.collect(); torch.cuda.empty_cache()
    L.concurrent_streams = real
