"""The DROP-IN route end to end, driven the way AllenAct's ``OnPolicyRLEngine`` drives it -- through the plugin classes
only, with the reference's tensor contracts at every hand-over:

    sensor frames  (HOST fp32 NHWC [N,224,224,3], already CLIP-normalised)
      -> ``ClipResNetPreprocessor.process``              -> fp32 NCHW [N,2048,7,7] on the device   (the plugin contract)
      -> rollout storage (fp32 NCHW, [T+1,N,2048,7,7])   [U] ``RolloutStorage.insert``
      -> ``ResnetTensorObjectNavActorCritic.forward``    T=1 act steps (no_grad), ``Memory`` round trip, ``sample()``
    -> ``compute_returns`` (GAE) -> ``update_repeats`` x { forward over [T,N] -> ``PPO.loss`` -> ``backward()`` ->
       per-parameter ``.grad`` -> ``optimizer.step()`` }, the optimiser being what the config's ``optimizer_builder`` names:
       ``FusedClipAdam`` (clip + Adam in one launch over the policy's flat bucket: INTEGRATION.md shows the one-line swap) or,
       with ``optimizer="torch"``, the reference's ``clip_grad_norm_`` + ``torch.optim.Adam``

(readme_files/baselines_robothor_objectnav.md:25,48-51: this is what ``allenact_main`` runs when the experiment configs
are used unchanged.)  ``engine.Worker`` is the same arithmetic with the MI355X-native data flow (bf16 NHWC features
written straight into the rollout buffer, frames resident in HBM, flat-bucket clip+Adam); ``bench.py`` times both and
reports the ratio, so a user of the plugin route knows what they get.
"""
from __future__ import annotations

import time
from typing import Dict

import torch

from . import spaces
from . import synthetic as syn
from .clip_preprocessors import ClipResNetPreprocessor
from .policy import Memory, ResnetTensorObjectNavActorCritic
from .ppo import PPO, FusedClipAdam, compute_returns, linear_decay_lr


class PluginPathRunner:
    """One DD-PPO worker built from the plugin classes (single GPU; synthetic frames, goals, masks and rewards as
    ``engine.SyntheticEnv`` makes them)."""

    def __init__(self, n_actors: int, T: int = 128, device="cuda:0", seed: int = 0, update_repeats: int = 4,
                 lr: float = 3e-4, max_grad_norm: float = 0.5, gamma: float = 0.99, tau: float = 0.95,
                 pool_steps: int = 4, frames_u8: bool = False, optimizer: str = "fused"):
        """``frames_u8``: the RGB sensor hands over raw uint8 HWC frames (``ClipResNetPreprocessor.process`` then fuses /255
        and the CLIP mean / std into the stem kernel): a quarter of the PCIe bytes of the normalised fp32 frames."""
        self.N, self.T, self.dev = n_actors, T, torch.device(device)
        self.update_repeats, self.max_grad_norm, self.gamma, self.tau, self.lr = update_repeats, max_grad_norm, gamma, tau, lr
        dev = self.dev
        self.pre = ClipResNetPreprocessor("rgb", "RN50", pool=False, device=dev, state_dict=syn.rn50_visual_state_dict(0))
        obs_space = spaces.Dict({"rgb_clip_resnet": self.pre.observation_space, "goal": spaces.Discrete(12)})
        self.model = ResnetTensorObjectNavActorCritic(spaces.Discrete(6), obs_space, goal_sensor_uuid="goal",
                                                      rgb_resnet_preprocessor_uuid="rgb_clip_resnet",
                                                      state_dict=syn.policy_state_dict(0), device=dev)
        self.loss = PPO()
        self.optimizer_kind = optimizer
        if optimizer == "fused":     # Builder(FusedClipAdam, dict(lr=lr, max_grad_norm=0.5)) in the experiment config
            self.opt = FusedClipAdam(self.model.parameters(), lr=lr, max_grad_norm=max_grad_norm)
        else:                        # the reference's Builder(optim.Adam, dict(lr=lr)) + the engine's clip_grad_norm_
            self.opt = torch.optim.Adam(self.model.parameters(), lr=lr)
        (dims, _), = self.model.recurrent_memory_specification.values()
        self.sampler_dim = [d[0] for d in dims].index("sampler")
        H = self.model.recurrent_hidden_state_size
        # what the simulators hand over: host fp32 NHWC frames (pinned, so the copy inside process() can be asynchronous)
        base = syn.synthetic_rgb_u8(1000 + seed, n_actors) if frames_u8 else syn.synthetic_rgb(1000 + seed, n_actors)
        self.host_frames = [base.roll(shifts=s + 1, dims=0).roll(shifts=7 * (s + 1), dims=2).contiguous().pin_memory()
                            for s in range(pool_steps)]
        masks = torch.cat([torch.ones(1, n_actors, 1), syn.synthetic_masks(1001 + seed, T, n_actors)], 0)
        self.masks = masks.to(dev)                                                     # [T+1, N, 1]
        self.goals = syn.synthetic_goals(1002 + seed, (T + 1, n_actors)).to(dev)      # [T+1, N]
        self.rewards = syn.synthetic_rewards(1003 + seed, masks[1:]).to(dev)          # [T, N, 1]
        # [U] RolloutStorage: observations / memory / actions / log-probs / values, step-major
        self.feat = torch.zeros((T + 1, n_actors, 2048, 7, 7), dtype=torch.float32, device=dev)
        self.memory = torch.zeros((T + 1, 1, n_actors, H), dtype=torch.float32, device=dev)
        self.actions = torch.zeros((T, n_actors), dtype=torch.int64, device=dev)
        self.logp = torch.zeros((T, n_actors, 1), dtype=torch.float32, device=dev)
        self.values = torch.zeros((T + 1, n_actors, 1), dtype=torch.float32, device=dev)
        self.total_steps, self._k = 0, 0
        self.feat[0] = self.pre.process({"rgb": self._observe()})
        self.info: Dict[str, float] = {}

    def step_optimizer(self):
        if self.optimizer_kind != "fused":
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.max_grad_norm)
        for g in self.opt.param_groups:
            g["lr"] = linear_decay_lr(self.lr, self.total_steps, 300_000_000)
        self.opt.step()

    def _observe(self) -> torch.Tensor:
        f = self.host_frames[self._k % len(self.host_frames)]
        self._k += 1
        return f

    def _mem(self, t: int) -> Memory:
        return Memory().check_append("rnn", self.memory[t:t + 1], self.sampler_dim + 1).step_squeeze(0)

    def iteration(self):
        T = self.T
        with torch.no_grad():
            for t in range(T):
                obs = {"rgb_clip_resnet": self.feat[t:t + 1], "goal": self.goals[t:t + 1]}
                out, mem = self.model(obs, self._mem(t), None, self.masks[t:t + 1])
                a = out.distributions.sample()
                lp = out.distributions.log_prob(a)
                # [U] OnPolicyRLEngine.collect_step_across_all_task_samplers: act -> vector_tasks.step(actions) (env.step happens
                # here in the real system; its frames come back on the host) -> the preprocessor graph on the new observations ->
                # ONE rollouts.insert(observations, memory, actions, action_log_probs, value_preds, rewards, masks)
                nxt = self.pre.process({"rgb": self._observe()})
                self.feat[t + 1] = nxt
                self.actions[t] = a[0]
                self.logp[t] = lp[0].unsqueeze(-1)
                self.values[t] = out.values[0]
                self.memory[t + 1] = mem.tensor("rnn")
            out, _ = self.model({"rgb_clip_resnet": self.feat[T:T + 1], "goal": self.goals[T:T + 1]}, self._mem(T), None,
                                self.masks[T:T + 1])
            self.values[T] = out.values[0]
            returns, adv, nadv = compute_returns(self.rewards, self.values, self.masks, self.gamma, self.tau)
        batch = dict(actions=self.actions, old_action_log_probs=self.logp, values=self.values[:T], returns=returns[:T],
                     norm_adv_targ=nadv, adv_targ=adv)
        for _ in range(self.update_repeats):
            obs = {"rgb_clip_resnet": self.feat[:T], "goal": self.goals[:T]}
            out, _ = self.model(obs, self._mem(0), self.actions, self.masks[:T])
            total, self.info = self.loss.loss(self.total_steps, batch, out)
            self.opt.zero_grad()
            total.backward()
            self.step_optimizer()
        with torch.no_grad():
            self.feat[0].copy_(self.feat[T])
            self.memory[0].copy_(self.memory[T])
        self.total_steps += T * self.N


def time_plugin_path(n_actors: int, T: int, device, steps: int = 1, warmup: int = 1, update_repeats: int = 4,
                     frames_u8: bool = False, optimizer: str = "fused") -> Dict:
    r = PluginPathRunner(n_actors, T, device, update_repeats=update_repeats, frames_u8=frames_u8, optimizer=optimizer)
    for _ in range(warmup):
        r.iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r.iteration()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"value": round(T * n_actors * steps / dt, 1), "unit": "env-frames/s", "steps": steps, "warmup": warmup,
            "ms_per_step": round(dt / steps * 1e3, 2), "loss": {k: round(float(v), 6) for k, v in r.info.items()},
            "route": ("HOST uint8 HWC frames (raw sensor output; /255 + CLIP mean/std fused into the stem)" if frames_u8 else
                      "HOST fp32 NHWC frames") + " -> ClipResNetPreprocessor.process (fp32 NCHW out) -> fp32 NCHW rollout storage -> "
                     "ResnetTensorObjectNavActorCritic.forward (T=1 act, T=rollout learn) -> PPO.loss -> backward() -> "
                     "per-parameter grads -> " + ("FusedClipAdam.step (clip + Adam, one launch on the flat bucket)" if optimizer == "fused"
                                                  else "clip_grad_norm_ -> torch.optim.Adam")}
