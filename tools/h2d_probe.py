"""How many HIP streams of one process really run concurrently (4: the runtime's hardware queues), and what a copy stream that
shares a queue with the OTHER slice's compute stream costs the H2D-inclusive order (before engine.Worker took its copy streams
from the verified pool: worker 3 of a process 42.3 k instead of 59 k).  python tools/h2d_probe.py on one MI355X."""
import sys, time, gc, torch, ctypes as C
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embodied_clip_amd import _lib
from embodied_clip_amd.engine import Worker
def run(w, n=3, warm=1):
    for _ in range(warm): w.iteration()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): w.iteration()
    torch.cuda.synchronize(); return 256 * 128 * n / (time.perf_counter() - t0)
lib = _lib.load()
def ov(a, b):
    r = C.c_float(); lib.ec_stream_pair_overlap(a.cuda_stream, b.cuda_stream, 200, C.byref(r)); return round(r.value, 2)
for n in (3, 4, 5, 6):
    try:
        st = _lib.concurrent_streams(n, "cuda:0"); print(n, "concurrent streams: ok")
        del st
    except RuntimeError as e:
        print(n, "->", e)
for i in range(3):
    w = Worker(256, T=128, device="cuda:0", seed=0, frames_host=True, frames_u8=True)
    r = run(w)
    s0, s1 = w.slices[0], w.slices[1]
    print(f"h2d worker {i+1}: {round(r)}; overlap slice0-slice1 {ov(s0.stream, s1.stream)}, slice0-copy0 {ov(s0.stream, s0.copy_stream)}, slice0-copy1 {ov(s0.stream, s1.copy_stream)}, "
          f"slice1-copy0 {ov(s1.stream, s0.copy_stream)}, slice1-copy1 {ov(s1.stream, s1.copy_stream)}, copy0-copy1 {ov(s0.copy_stream, s1.copy_stream)}")
    del w, s0, s1; gc.collect(); torch.cuda.empty_cache()
