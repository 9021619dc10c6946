"""Host cost per kernel launch through the C-ABI (ctypes + hipLaunchKernelGGL), and of one ec_rn50_forward call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from embodied_clip_amd import _lib, synthetic as syn
from embodied_clip_amd.encoder import RN50Trunk
lib = _lib.load()
dev = torch.device("cuda:0")
x = torch.zeros(64, dtype=torch.bfloat16, device=dev); y = torch.zeros(64, device=dev)
sp = _lib.stream_ptr()
for _ in range(100): lib.ec_bf16_to_f32(x.data_ptr(), y.data_ptr(), 1, 64, 64, sp)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000): lib.ec_bf16_to_f32(x.data_ptr(), y.data_ptr(), 1, 64, 64, sp)
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"tiny kernel via ctypes: {(t1 - t0) / 2000 * 1e6:.1f} us per launch (host)")
trunk = RN50Trunk(syn.rn50_visual_state_dict(0), device=dev)
rgb = syn.synthetic_rgb(1, 8).to(dev).repeat(16, 1, 1, 1).contiguous()
out = trunk.forward(rgb); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): trunk.forward(rgb, out)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"ec_rn50_forward(128 frames): host {(t1 - t0) / 20 * 1e3:.3f} ms per call ({(t1 - t0) / 20 / 54 * 1e6:.1f} us per launch), gpu {(t2 - t0) / 20 * 1e3:.3f} ms")
s2 = torch.cuda.Stream()
with torch.cuda.stream(s2):
    sp2 = _lib.stream_ptr()
    t0 = time.perf_counter()
    for _ in range(2000): lib.ec_bf16_to_f32(x.data_ptr(), y.data_ptr(), 1, 64, 64, sp2)
    t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"tiny kernel on a side stream: {(t1 - t0) / 2000 * 1e6:.1f} us per launch (host)")
