"""Host-side (Python) cost of the plugin route's per-env-step section: cProfile over act steps with the GPU drained each step."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from embodied_clip_amd.plugin_path import PluginPathRunner
r = PluginPathRunner(256, 8, "cuda:0", frames_u8=True)
r.iteration(); torch.cuda.synchronize()
def act(t):
    obs = {"rgb_clip_resnet": r.feat[t:t + 1], "goal": r.goals[t:t + 1]}
    out, mem = r.model(obs, r._mem(t), None, r.masks[t:t + 1])
    a = out.distributions.sample()
    r.actions[t] = a[0]
    r.logp[t] = out.distributions.log_prob(a)[0].unsqueeze(-1)
    r.values[t] = out.values[0]
    r.memory[t + 1] = mem.tensor("rnn")
with torch.no_grad():
    for t in range(4): act(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(40):
        act(k % 8); torch.cuda.synchronize()
    print("act section, synchronised each step: %.3f ms/step" % ((time.perf_counter() - t0) / 40 * 1e3))
    pr = cProfile.Profile(); pr.enable()
    for k in range(40):
        act(k % 8); torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(35)
    fr = r.host_frames[0]
    pr = cProfile.Profile(); pr.enable()
    for k in range(20):
        r.pre.process({"rgb": fr}); torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
