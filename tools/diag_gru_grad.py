"""Diagnostic: HIP policy gradients vs the oracle's, per parameter; saves them so that two EC_GRU_FUSED settings can be compared."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from embodied_clip_amd import synthetic as syn
from embodied_clip_amd.policy import PolicyHandle
from embodied_clip_amd.ppo import ppo_loss_raw
import _a18_case
T, N = int(sys.argv[1]), int(sys.argv[2]); tag = sys.argv[3]
dev = torch.device("cuda:0")
_, _, c = _a18_case.make(T, N)
sd = syn.policy_state_dict(0)
h = PolicyHandle(); flat = h.flatten(sd, dev)
f = lambda t: t.reshape(-1).contiguous().to(dev)
rows = c["feat"].reshape(T * N, 49, 2048).contiguous().to(dev)
ws = torch.empty(h.workspace_bytes(T, N, True), dtype=torch.uint8, device=dev)
hv, _ = h.forward(flat, rows, f(c["goal"]), c["h0"].contiguous().to(dev), f(c["masks"]), T, N, ws)
dhv, _ = ppo_loss_raw(hv, f(c["actions"]), f(c["old_lp"]), f(c["old_v"]), f(c["ret"]), f(c["nadv"]), 6)
gr = torch.zeros_like(flat)
h.backward(flat, rows, f(c["masks"]), T, N, ws, dhv, None, gr)
torch.cuda.synchronize()
gr = gr.cpu(); hv = hv.cpu()
ref = _a18_case.oracle_gradient(sd, T, N) if T * N <= 512 else None
prev = torch.load(f"/tmp/diag_{T}_{N}.pt") if os.path.exists(f"/tmp/diag_{T}_{N}.pt") else None
for name, (o, k) in h.offsets.items():
    a = gr[o:o + k]
    line = f"{name:52s}"
    if ref is not None:
        b = ref[name].reshape(-1)
        line += f" vs oracle rel {((a - b).norm() / b.norm()).item():.2e} maxabs/max {((a - b).abs().max() / b.abs().max()).item():.2e}"
    if prev is not None:
        p = prev["gr"][o:o + k]
        line += f" | vs prev run rel {((a - p).norm() / p.norm()).item():.2e} maxabs/max {((a - p).abs().max() / p.abs().max()).item():.2e}"
    print(line)
if prev is not None:
    print("hv vs prev: max abs", (hv - prev["hv"]).abs().max().item())
torch.save({"gr": gr, "hv": hv}, f"/tmp/diag_{T}_{N}.pt")
