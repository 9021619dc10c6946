import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from embodied_clip_amd.engine import Worker
w = Worker(256, T=128, device="cuda:0")
w.collect_rollout(); torch.cuda.synchronize()
t0 = time.perf_counter(); w.collect_rollout(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"rollout: host issue {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms")
w.compute_returns(); torch.cuda.synchronize()
t0 = time.perf_counter(); w.update(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"update: host issue {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms")
