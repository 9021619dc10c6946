"""Known-byte-count launches for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on THIS library's access patterns
(MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports 1/2 of the bytes of a 16-B/lane streaming read; other widths are
uncalibrated).  Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, tools/calibrate_fetch.sh);
this script prints the EXACT algorithmic bytes each launch reads / writes, tools/calibrate_fetch_summary.py divides.

The inputs are sized past the 256-MiB Infinity Cache and freshly written by a different kernel each time, so a launch
cannot be served from on-die copies of its own input."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json
import torch
from embodied_clip_amd import _lib, encoder as enc, synthetic as syn

dev = torch.device("cuda:0")
lib = _lib.load()
B = int(os.environ.get("B", "640"))          # 640 frames: fp32 RGB 385 MB, stem activations 514 MB
g = torch.Generator().manual_seed(0)
rec = []
n = 512 * 1024 * 1024 // 4
src = torch.randn(n, device=dev); dst = torch.empty_like(src)
flush = torch.empty(96 * 1024 * 1024, device=dev)      # 384 MB written between launches: evicts L2 + Infinity Cache
def evict(): flush.add_(1.0)
# (2) stem conv1: fp32 NHWC frames in (12 B per pixel, float4 staging loads), bf16 32-channel rows out
sd = syn.rn50_visual_state_dict(0)
trunk = enc.RN50Trunk(sd, device=dev)
rgb = torch.randn(B, 224, 224, 3, generator=g).to(dev)
out1 = torch.empty(B, 112, 112, 32, dtype=torch.bfloat16, device=dev)
evict()
_lib.check(lib.ec_stem_conv1(rgb.data_ptr(), trunk.stem_w.data_ptr(), trunk.bias.data_ptr(), out1.data_ptr(), B, 224, 224, 32,
                             _lib.stream_ptr()))
torch.cuda.synchronize()
rec.append({"kernel": "stem_conv1_kernel", "match": "stem_conv1_kernel", "read": rgb.numel() * 4, "write": out1.numel() * 2})
# (3) stem conv2 (conv3x3 row tiles, 32 -> 32 @112x112): footprint rows re-read ~1.5x from L2 -- the HBM-side bytes are the tensor once
w2 = (torch.randn(32, 9 * 32, generator=g) * 0.05).to(torch.bfloat16).to(dev); b2 = torch.randn(32, generator=g).to(dev)
evict()
out2 = enc.conv_bf16(out1, w2, b2, None, ksize=3, pool=False, act=1)
torch.cuda.synchronize()
rec.append({"kernel": "conv3x3_rows*<32, 32>", "match": "conv3x3_rows", "read": out1.numel() * 2, "write": out2.numel() * 2})
# (4) an 8-wave implicit-GEMM conv (LDS-DMA, 16 B per lane): 1x1 1024 -> 256 @14x14
x4 = torch.randn(B, 14, 14, 1024, generator=g).to(torch.bfloat16).to(dev)
w4 = (torch.randn(256, 1024, generator=g) * 0.05).to(torch.bfloat16).to(dev); b4 = torch.randn(256, generator=g).to(dev)
evict()
out4 = enc.conv_bf16(x4, w4, b4, None, ksize=1, pool=False, act=1)
torch.cuda.synchronize()
rec.append({"kernel": "conv_igemm8_kernel<256, 1>", "match": "conv_igemm8_kernel", "read": x4.numel() * 2 + w4.numel() * 2, "write": out4.numel() * 2})
# (1) the reference pattern, LAST (tools/calibrate_fetch.sh takes each kernel's last dispatch): a 16-B-per-lane streaming
#     elementwise kernel reading 512 MiB and writing 512 MiB
evict(); torch.add(src, 1.0, out=dst); torch.cuda.synchronize()
rec.insert(0, {"kernel": "vectorized_elementwise_kernel (out = in + 1, 512 MiB each way)", "match": "vectorized_elementwise_kernel",
               "read": n * 4, "write": n * 4})
print("CALIB " + json.dumps(rec))
