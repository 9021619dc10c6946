"""Experiment: two half-batches of the RN50 trunk on two HIP streams (offset by a few layers in practice)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from embodied_clip_amd import synthetic as syn
from embodied_clip_amd.encoder import RN50Trunk
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=256); ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--streams", type=int, default=2); ap.add_argument("--offset-ms", type=float, default=0.0)
a = ap.parse_args()
dev = torch.device("cuda:0")
sd = syn.rn50_visual_state_dict(0)
S = a.streams
trunks = [RN50Trunk(sd, device=dev) for _ in range(S)]
streams = [torch.cuda.Stream() for _ in range(S)]
nb = a.batch // S
rgbs = [syn.synthetic_rgb(1 + i, 8).to(dev).repeat((nb + 7) // 8, 1, 1, 1)[:nb].contiguous() for i in range(S)]
outs = [None] * S
for i in range(S):
    with torch.cuda.stream(streams[i]):
        outs[i] = trunks[i].forward(rgbs[i])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for s in streams: s.wait_stream(torch.cuda.current_stream())
for i in range(1, S):
    if a.offset_ms > 0:
        with torch.cuda.stream(streams[i]):
            torch.cuda._sleep(int(a.offset_ms * i * 1e-3 * 2.0e9))
for _ in range(a.iters):
    for i in range(S):
        with torch.cuda.stream(streams[i]):
            trunks[i].forward(rgbs[i], outs[i])
for s in streams: torch.cuda.current_stream().wait_stream(s)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.iters
print(f"offset={a.offset_ms} streams={S} batch={a.batch} wgs={os.environ.get('EC_CONV_WGS','768')}: {ms:.3f} ms per {a.batch} frames  {a.batch/ms*1e3:.0f} frames/s")
