"""Experiment (tools only): do the engine's two slice streams gain from DISJOINT compute-unit sets?  hipExtStreamCreateWithCUMask
through ctypes, the streams handed to torch as ExternalStream and put in place of a worker's slice streams before its first launch.
Masks (256 CUs = 8 words): none (baseline) | xcd (bit i -> slice (i % 8) // 4: four XCDs each, if the mask bits go round the XCDs) |
halves (bits 0-127 / 128-255) | evenodd (i % 2).   python tools/cumask_probe.py on one MI355X."""
import ctypes as C
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embodied_clip_amd import _lib           # noqa: E402
from embodied_clip_amd.engine import Worker  # noqa: E402

hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = C.c_int


def masked_stream(pred):
    words = (C.c_uint32 * 8)()
    for i in range(256):
        if pred(i):
            words[i // 32] |= 1 << (i % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device="cuda:0")


def run(w, n=4, warm=2):
    for _ in range(warm):
        w.iteration()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        w.iteration()
    torch.cuda.synchronize()
    return 256 * 128 * n / (time.perf_counter() - t0)


MASKS = {"none": None,
         "xcd": [lambda i: (i % 8) < 4, lambda i: (i % 8) >= 4],
         "halves": [lambda i: i < 128, lambda i: i >= 128],
         "evenodd": [lambda i: i % 2 == 0, lambda i: i % 2 == 1],
         "xcd_pairs": [lambda i: (i % 8) % 2 == 0, lambda i: (i % 8) % 2 == 1]}
lib = _lib.load()
for name in (sys.argv[1:] or list(MASKS)):
    for rep in range(2):
        w = Worker(256, T=128, device="cuda:0", seed=0)
        if MASKS[name] is not None:
            st = [masked_stream(p) for p in MASKS[name]]
            arr = (C.c_void_p * 2)(*[s.cuda_stream for s in st])
            _lib.check(lib.ec_bind_streams(arr, 2, 200))
            for sl, s in zip(w.slices, st):
                sl.stream = s
        r = run(w)
        ov = C.c_float()
        lib.ec_stream_pair_overlap(w.slices[0].stream.cuda_stream, w.slices[1].stream.cuda_stream, 200, C.byref(ov))
        print(f"{name:10s} run {rep}: {r:9.0f} env-frames/s   (slice streams both busy / alone = {ov.value:.2f})", flush=True)
        del w
        gc.collect(); torch.cuda.empty_cache()
