"""Where the plugin route's iteration goes: per-call timings (synchronised) of the drop-in classes at 256 actors."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from embodied_clip_amd.plugin_path import PluginPathRunner
N, T = int(os.environ.get("N", "256")), int(os.environ.get("T", "16"))
r = PluginPathRunner(N, T, "cuda:0")
r.iteration(); torch.cuda.synchronize()
def tm(f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
fr = r.host_frames[0]
print("process (H2D + trunk + to_nchw) ms:", round(tm(lambda: r.pre.process({"rgb": fr})), 3))
d = fr.to("cuda:0")
print("process on device-resident frames ms:", round(tm(lambda: r.pre.process({"rgb": d})), 3))
print("H2D alone ms:", round(tm(lambda: fr.to("cuda:0", non_blocking=True)), 3))
feat = r.pre.process({"rgb": fr})
print("storage insert (copy 103 MB) ms:", round(tm(lambda: r.feat[1].copy_(feat)), 3))
def act():
    with torch.no_grad():
        out, mem = r.model({"rgb_clip_resnet": r.feat[0:1], "goal": r.goals[0:1]}, r._mem(0), None, r.masks[0:1])
        a = out.distributions.sample(); lp = out.distributions.log_prob(a)
print("act step (model fwd T=1 + sample + log_prob) ms:", round(tm(act), 3))
def trans():
    return r.feat[0:1].reshape(N, 2048, -1).transpose(1, 2).contiguous()
print("  of which NCHW->NHWC transpose ms:", round(tm(trans), 3))
T2 = T
def learn():
    out, _ = r.model({"rgb_clip_resnet": r.feat[:T2], "goal": r.goals[:T2]}, r._mem(0), r.actions, r.masks[:T2])
    batch = dict(actions=r.actions, old_action_log_probs=r.logp, values=r.values[:T2], returns=r.values[:T2] + 0.1,
                 norm_adv_targ=r.values[:T2], adv_targ=r.values[:T2])
    total, info = r.loss.loss(0, batch, out)
    r.opt.zero_grad(); total.backward()
    torch.nn.utils.clip_grad_norm_(r.model.parameters(), 0.5); r.opt.step()
print(f"learn epoch at T={T2} ms:", round(tm(learn, 3), 3), "-> per 128 steps:", round(tm(learn, 3) * 128 / T2, 1))
