"""Phase times of ONE full plugin-route iteration (256 actors x rollout 128): where `plugin_path` in the bench line goes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from embodied_clip_amd.plugin_path import PluginPathRunner
from embodied_clip_amd.ppo import compute_returns, linear_decay_lr
N, T = int(os.environ.get("N", "256")), int(os.environ.get("T", "128"))
u8 = os.environ.get("U8", "0") == "1"
r = PluginPathRunner(N, T, "cuda:0", frames_u8=u8)
r.iteration(); torch.cuda.synchronize()
def now():
    torch.cuda.synchronize(); return time.perf_counter()
def run(report):
    t0 = now()
    host_proc = host_act = 0.0
    with torch.no_grad():
        for t in range(T):
            h0 = time.perf_counter()
            obs = {"rgb_clip_resnet": r.feat[t:t + 1], "goal": r.goals[t:t + 1]}
            out, mem = r.model(obs, r._mem(t), None, r.masks[t:t + 1])
            a = out.distributions.sample()
            lp = out.distributions.log_prob(a)
            h1 = time.perf_counter()
            r.feat[t + 1] = r.pre.process({"rgb": r._observe()})      # (same order as PluginPathRunner.iteration: insert after the preprocessor)
            h2 = time.perf_counter()
            r.actions[t] = a[0]
            r.logp[t] = lp[0].unsqueeze(-1)
            r.values[t] = out.values[0]
            r.memory[t + 1] = mem.tensor("rnn")
            host_act += h1 - h0; host_proc += h2 - h1
        t1 = now()
        out, _ = r.model({"rgb_clip_resnet": r.feat[T:T + 1], "goal": r.goals[T:T + 1]}, r._mem(T), None, r.masks[T:T + 1])
        r.values[T] = out.values[0]
        returns, adv, nadv = compute_returns(r.rewards, r.values, r.masks, r.gamma, r.tau)
    t2 = now()
    batch = dict(actions=r.actions, old_action_log_probs=r.logp, values=r.values[:T], returns=returns[:T], norm_adv_targ=nadv, adv_targ=adv)
    ep = []
    for e in range(r.update_repeats):
        s0 = now()
        obs = {"rgb_clip_resnet": r.feat[:T], "goal": r.goals[:T]}
        out, _ = r.model(obs, r._mem(0), r.actions, r.masks[:T])
        s1 = now()
        total, info = r.loss.loss(0, batch, out)
        s2 = now()
        r.opt.zero_grad(); total.backward()
        s3 = now()
        r.step_optimizer() if hasattr(r, "step_optimizer") else (torch.nn.utils.clip_grad_norm_(r.model.parameters(), r.max_grad_norm), r.opt.step())
        s4 = now()
        ep.append([round((b - a) * 1e3, 2) for a, b in ((s0, s1), (s1, s2), (s2, s3), (s3, s4))])
    t3 = now()
    if report: print(f"rollout {1e3*(t1-t0):.1f} ms ({1e3*(t1-t0)/T:.3f}/step; host time in act {1e3*host_act/T:.3f}, in process {1e3*host_proc/T:.3f} per step)")
    if report: print(f"final value + GAE {1e3*(t2-t1):.2f} ms")
    if report: print("update epochs [forward, loss, backward, clip+adam] ms:", ep, f"total {1e3*(t3-t2):.1f}")


run(False)      # (first pass of THIS loop: the caching allocator settles on the loop's allocation pattern)
run(True)
