"""The ImageNet tower (torchvision ResNet-50) on one MI355X: ms per launch for fp32 and raw uint8 frames, and the
stem kernel alone.  torchvision resnet50 trunk = 4,087 MMAC per 224 x 224 frame (conv1 118 + layers 3,969)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from embodied_clip_amd import _lib
if os.environ.get("EC_AMD_LIB"):      # same-box A/B of two builds
    _lib.LIB_PATH = os.environ["EC_AMD_LIB"]
from embodied_clip_amd import encoder as enc, synthetic as syn

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--trunk-only", action="store_true", help="only the fp32-frame trunk timing (the LAST launches of the process are one forward: what tools/pmc_summary.py joins)")
a = ap.parse_args()
dev = torch.device("cuda:0")
sd = syn.tv_resnet_state_dict(0)
trunk = enc.ImageNetRN50Trunk(sd, device=dev)
u8 = syn.synthetic_rgb_u8(1, 8).to(dev).repeat((a.batch + 7) // 8, 1, 1, 1)[:a.batch].contiguous()
x = syn.normalize_rgb_imagenet(u8).contiguous()
res = {1: 56, 2: 28, 3: 14, 4: 7}
mac = 112 * 112 * 64 * 147
for k, v in sd.items():
    if k.startswith("layer") and k.endswith("weight") and v.dim() == 4:
        l, blk = int(k[5]), int(k.split(".")[1])
        r = res[l - 1] if (blk == 0 and l > 1 and ".conv1." in k) else res[l]      # (layerN.0.conv1 of layers 2-4 runs at the input resolution)
        mac += v.numel() * r * r
def tm(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters
out = trunk.forward(x)
if a.trunk_only:
    ms = tm(lambda: trunk.forward(x, out))
    print(f"batch={a.batch} {ms:.3f} ms/forward")
    print('plan_hash', trunk.plan_hash())
    print('num_ops', trunk.lib.ec_rn50_num_ops(trunk.h))
    sys.exit(0)
ms = tm(lambda: trunk.forward(x, out)); ms8 = tm(lambda: trunk.forward_u8(u8, out))
(_w, _l), stem_w, _wf, bias = enc.pack_tv_resnet(sd)
sw, sb = stem_w.to(dev), bias[:64].contiguous().to(dev)
ms_stem = tm(lambda: enc.stem7_pool(x, sw, sb)); ms_stem8 = tm(lambda: enc.stem7_pool(u8, sw, sb, mean=syn.IMAGENET_RGB_MEANS, std=syn.IMAGENET_RGB_STDS))
print(f"torchvision ResNet-50 trunk, batch {a.batch}: {ms:.3f} ms per launch (fp32 frames) = {a.batch / ms * 1e3:.0f} frames/s, "
      f"{a.batch * 2 * mac / ms / 1e9:.0f} TFLOP/s ({mac / 1e6:.0f} MMAC per frame); raw uint8 frames {ms8:.3f} ms; {trunk.lib.ec_rn50_num_ops(trunk.h)} launches")
b_in, b_out = 224 * 224 * 3, 56 * 56 * 64 * 2
print(f"stem7_pool_kernel alone: fp32 frames {ms_stem * 1e3:.1f} us ({a.batch * (4 * b_in + b_out) / ms_stem / 1e9:.2f} TB/s of algorithmic bytes), "
      f"uint8 frames {ms_stem8 * 1e3:.1f} us ({a.batch * (b_in + b_out) / ms_stem8 / 1e9:.2f} TB/s)")
print("plan_hash", trunk.plan_hash())
