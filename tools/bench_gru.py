"""Times ec_policy_forward for one slice of the update phase (T=128, N=128) -- dominated by the GRU recurrence when the
compressor input is tiny (in_channels=64 here, so the front GEMMs are negligible)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from embodied_clip_amd import synthetic as syn
from embodied_clip_amd.policy import PolicyHandle
dev = torch.device("cuda:0")
T, N = 128, 128
h = PolicyHandle(in_channels=64, spatial=7)
flat = h.flatten(syn.policy_state_dict(0, in_channels=64, spatial=7), dev)
feat = torch.randn(T * N, 49, 64, device=dev).abs().to(torch.bfloat16)
goal = torch.zeros(T * N, dtype=torch.int64, device=dev); h0 = torch.zeros(N, 512, device=dev); m = torch.ones(T * N, device=dev)
ws = torch.empty(h.workspace_bytes(T, N, True), dtype=torch.uint8, device=dev)
for _ in range(2): h.forward(flat, feat, goal, h0, m, T, N, ws)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): h.forward(flat, feat, goal, h0, m, T, N, ws)
torch.cuda.synchronize()
print(f"policy forward T={T} N={N}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms  (EC_GRU_FUSED={os.environ.get('EC_GRU_FUSED', '1')})")
