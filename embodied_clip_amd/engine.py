"""One DD-PPO worker on one MI355X: rollout storage + the two hot loops.

This is the per-GPU body of [U] AllenAct ``OnPolicyTrainer.run_pipeline``
(launched by the reference at
readme_files/baselines_robothor_objectnav.md:48-51) reduced to its
data-parallel hot path:

  HOT LOOP A (act), x T:  frames -> frozen CLIP encoder -> policy act step -> sample
  GAE returns + advantage normalisation
  HOT LOOP B (learn), x update_repeats: policy forward over [T,N] -> PPO loss ->
      backward -> (grad *= local/global) -> ONE flat-bucket RCCL all-reduce -> clip + Adam

Everything between the simulator's frames and the optimiser update runs in
the HIP library through the C-ABI; torch supplies device memory, the stream
and ``torch.distributed`` (backend "nccl" == RCCL over xGMI).  The simulator
itself is out of scope: ``SyntheticEnv`` plays pre-rendered frame batches that
are already resident in HBM.

Rollout storage layout (the encoder writes straight into it):
  feat    bf16 [T+1, N, S*S, C]   channels-last rows == the compressor GEMM's A operand
  memory  f32  [N, H] at rollout start; masks f32 [T+1, N]; goal i64 [T+1, N]
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from . import _lib
from . import synthetic as syn
from .dist import allreduce_flat
from .encoder import RN50Trunk, ViTEmbedder
from .policy import PolicyHandle
from .ppo import FlatAdam, linear_decay_lr, ppo_loss_raw


class SyntheticEnv:
    """N synthetic actors: a pool of pre-rendered, CLIP-normalised 224x224 frames in HBM,
    random goal ids, episode resets w.p. 1/100, RoboTHOR-style rewards."""

    def __init__(self, n_actors: int, T: int, device, seed: int, pool_steps: int = 4, res: int = 224):
        self.N, self.T = n_actors, T
        base = syn.synthetic_rgb(seed, min(n_actors, 32), res)
        reps = (n_actors + base.shape[0] - 1) // base.shape[0]
        frames = []
        for s in range(pool_steps):   # distinct frame batches so consecutive steps differ
            f = base.roll(shifts=s + 1, dims=0).roll(shifts=7 * (s + 1), dims=2)
            frames.append(f.repeat(reps, 1, 1, 1)[:n_actors])
        self.frames = torch.stack(frames).to(device).contiguous()      # [P, N, R, R, 3] fp32
        self.pool_steps = pool_steps
        masks = torch.cat([torch.ones(1, n_actors, 1), syn.synthetic_masks(seed + 1, T, n_actors)], 0)
        self.masks = masks.reshape(T + 1, n_actors).to(device).contiguous()
        self.goals = syn.synthetic_goals(seed + 2, (T + 1, n_actors)).to(device).contiguous()
        self.rewards = syn.synthetic_rewards(seed + 3, masks[1:]).reshape(T, n_actors).to(device).contiguous()
        self._k = 0

    def observe(self) -> torch.Tensor:
        f = self.frames[self._k % self.pool_steps]
        self._k += 1
        return f


class Worker:
    def __init__(self, n_actors: int, T: int = 128, device="cuda:0", seed: int = 0, rank: int = 0, world: int = 1,
                 update_repeats: int = 4, lr: float = 3e-4, max_grad_norm: float = 0.5, gamma: float = 0.99,
                 tau: float = 0.95, encoder_sd=None, policy_sd=None, lr_total_steps: int = 300_000_000,
                 encoder_chunk: int = 0, encoder: str = "rn50", encoder_streams: int = 2):
        self.lib = _lib.load()
        self.dev = torch.device(device)
        self.N, self.T, self.rank, self.world = n_actors, T, rank, world
        self.update_repeats, self.gamma, self.tau = update_repeats, gamma, tau
        self.base_lr, self.lr_total_steps = lr, lr_total_steps
        self.encoder = encoder
        if encoder == "rn50":
            self.trunk = RN50Trunk(encoder_sd if encoder_sd is not None else syn.rn50_visual_state_dict(0),
                                   device=self.dev, chunk=encoder_chunk)
            self.S, self.C = self.trunk.out_spatial, self.trunk.out_channels
            # Two halves of the actor batch are encoded concurrently on two HIP streams: bandwidth-bound and
            # MFMA-bound conv launches of the two halves overlap on the chip (+11 % measured at N=256).
            self.enc_streams = []
            if encoder_streams > 1 and n_actors >= 64 and n_actors % encoder_streams == 0:
                self.enc_streams = [torch.cuda.Stream(device=self.dev) for _ in range(encoder_streams)]
                self.trunks = [self.trunk] + [RN50Trunk(encoder_sd if encoder_sd is not None
                                                        else syn.rn50_visual_state_dict(0), device=self.dev,
                                                        chunk=encoder_chunk) for _ in range(encoder_streams - 1)]
        elif encoder == "vit":
            # BASELINE config 3 (builder-defined fusion, SURVEY.md §8d note): ClipViTEmbedder tokens, CLS dropped,
            # the 49 patch tokens are the 7x7 channels-last "feature map" [N,49,768] of the goal encoder
            self.vit = ViTEmbedder(encoder_sd if encoder_sd is not None else syn.vit_visual_state_dict(0),
                                   device=self.dev)
            self.S, self.C = 7, self.vit.D
            self._tok = torch.empty((n_actors, self.vit.L, self.vit.D), dtype=torch.bfloat16, device=self.dev)
        else:
            raise ValueError(encoder)
        if not hasattr(self, 'enc_streams'):
            self.enc_streams = []
        pkw = dict(in_channels=self.C, spatial=self.S)
        self.policy = PolicyHandle(**pkw)
        self.H, self.A = self.policy.H, self.policy.A
        self.params = self.policy.flatten(policy_sd if policy_sd is not None else syn.policy_state_dict(0, **pkw),
                                          self.dev)
        self.grads = torch.zeros_like(self.params)
        self.opt = FlatAdam(self.params, lr=lr, max_grad_norm=max_grad_norm)
        N, S2 = n_actors, self.S * self.S
        d = self.dev
        self.feat = torch.empty((T + 1, N, S2, self.C), dtype=torch.bfloat16, device=d)
        self.actions = torch.zeros((T, N), dtype=torch.int64, device=d)
        self.logp = torch.zeros((T, N), dtype=torch.float32, device=d)
        self.values = torch.zeros((T + 1, N), dtype=torch.float32, device=d)
        self.returns = torch.zeros((T + 1, N), dtype=torch.float32, device=d)
        self.adv = torch.zeros((T, N), dtype=torch.float32, device=d)
        self.nadv = torch.zeros((T, N), dtype=torch.float32, device=d)
        self.h_start = torch.zeros((N, self.H), dtype=torch.float32, device=d)
        self.h = torch.zeros((N, self.H), dtype=torch.float32, device=d)
        self.h_next = torch.zeros((N, self.H), dtype=torch.float32, device=d)
        self.hv_act = torch.empty((N, self.A + 1), dtype=torch.float32, device=d)
        self.hv = torch.empty((T * N, self.A + 1), dtype=torch.float32, device=d)
        self.dhv = torch.empty_like(self.hv)
        self.sums = torch.zeros(4, dtype=torch.float64, device=d)
        self.stats = torch.zeros(2, dtype=torch.float64, device=d)
        self.ws_act = torch.empty(self.policy.workspace_bytes(1, N, False), dtype=torch.uint8, device=d)
        self.ws_act_half = [torch.empty(self.policy.workspace_bytes(1, N, False), dtype=torch.uint8, device=d)
                            for _ in range(len(getattr(self, 'enc_streams', [])))]
        self.encode_frames = N // max(1, len(getattr(self, 'enc_streams', [])))   # frames per timed encoder launch
        self.ws_learn = torch.empty(self.policy.workspace_bytes(T, N, True), dtype=torch.uint8, device=d)
        self.env = SyntheticEnv(N, T, d, seed=1000 + rank)
        self.seed = seed + 7919 * rank
        self.total_steps = 0
        self.iter = 0
        self.trunk_events: List = []     # (start, end) HIP event pairs around ec_rn50_forward
        self.time_trunk = False
        # first observation of the first rollout
        self._encode_raw(self.env.observe(), self.feat[0])
        self.last_info: Dict[str, float] = {}

    # ---- HOT LOOP A ---------------------------------------------------------------------------
    def _encode_raw(self, rgb: torch.Tensor, out: torch.Tensor):
        if self.encoder == "rn50":
            if self.enc_streams:
                cur = torch.cuda.current_stream()
                h = self.N // len(self.enc_streams)
                for i, (tr, st) in enumerate(zip(self.trunks, self.enc_streams)):
                    st.wait_stream(cur)
                    with torch.cuda.stream(st):
                        tr.forward(rgb[i * h:(i + 1) * h], out[i * h:(i + 1) * h])
                for st in self.enc_streams:
                    cur.wait_stream(st)
            else:
                self.trunk.forward(rgb, out)          # last conv writes straight into the rollout slice
        else:
            self.vit.forward(rgb, self._tok)
            out.copy_(self._tok[:, 1:, :])            # drop CLS: [N,49,768] channels-last rows

    def _encode(self, rgb: torch.Tensor, out: torch.Tensor):
        if self.time_trunk:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self._encode_raw(rgb, out)
            e1.record()
            self.trunk_events.append((e0, e1))
        else:
            self._encode_raw(rgb, out)

    def _act(self, t: int, sample: bool = True, o: int = 0, n: Optional[int] = None, ws=None):
        """Policy act step for actors [o, o+n) on the current stream (T=1, no grad)."""
        n = self.N if n is None else n
        ws = self.ws_act if ws is None else ws
        sp = _lib.stream_ptr()
        sl = slice(o, o + n)
        h_in, h_out = (self.h, self.h_next) if (t & 1) == 0 else (self.h_next, self.h)   # ping-pong by step parity
        self.policy.forward(self.params, self.feat[t][sl].view(n, self.S * self.S, self.C), self.env.goals[t][sl],
                            h_in[sl], self.env.masks[t][sl], 1, n, ws, hv=self.hv_act[sl], h_final=h_out[sl])
        if sample:
            _lib.check(self.lib.ec_sample_actions(self.hv_act[sl].data_ptr(), self.actions[t][sl].data_ptr(),
                                                  self.logp[t][sl].data_ptr(), self.values[t][sl].data_ptr(), n, self.A,
                                                  self.seed, self.iter * (self.T + 1) + t, o, sp), "ec_sample_actions")
        else:   # bootstrap value of the last observation; memory is NOT advanced
            self.values[t][sl].copy_(self.hv_act[sl, self.A])

    def collect_rollout(self):
        T = self.T
        self.h_start.copy_(self.h)
        if self.enc_streams and self.encoder == "rn50":
            # each slice of the actor batch runs its own act -> (env.step) -> encode chain on its own stream: the small
            # act-step kernels of one slice overlap the encoder of the other; per-actor results are unchanged
            cur = torch.cuda.current_stream()
            ns = len(self.enc_streams)
            hN = self.N // ns
            for st in self.enc_streams:
                st.wait_stream(cur)
            for t in range(T):
                rgb = self.env.observe()
                for i, st in enumerate(self.enc_streams):
                    with torch.cuda.stream(st):
                        self._act(t, True, i * hN, hN, self.ws_act_half[i])
                        sl = slice(i * hN, (i + 1) * hN)
                        if self.time_trunk:
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            e0.record(st)
                            self.trunks[i].forward(rgb[sl], self.feat[t + 1][sl])
                            e1.record(st)
                            self.trunk_events.append((e0, e1))
                        else:
                            self.trunks[i].forward(rgb[sl], self.feat[t + 1][sl])
            for i, st in enumerate(self.enc_streams):
                with torch.cuda.stream(st):
                    self._act(T, False, i * hN, hN, self.ws_act_half[i])
                cur.wait_stream(st)
        else:
            for t in range(T):
                self._act(t)
                # env.step(actions[t]) happens here in the real system; its frames arrive as fp32 NHWC
                self._encode(self.env.observe(), self.feat[t + 1])
            self._act(T, sample=False)
        if (T & 1) == 1:   # after an odd number of advancing steps the live memory sits in h_next
            self.h, self.h_next = self.h_next, self.h

    def compute_returns(self):
        _lib.check(self.lib.ec_gae(self.env.rewards.data_ptr(), self.values.data_ptr(), self.env.masks.data_ptr(),
                                   self.returns.data_ptr(), self.adv.data_ptr(), self.nadv.data_ptr(),
                                   self.stats.data_ptr(), self.T, self.N, self.gamma, self.tau, 1e-5,
                                   _lib.stream_ptr()), "ec_gae")

    # ---- HOT LOOP B ---------------------------------------------------------------------------
    def update(self):
        T, N = self.T, self.N
        feat = self.feat[:T].view(T * N, self.S * self.S, self.C)
        goal = self.env.goals[:T].reshape(-1)
        masks = self.env.masks[:T].reshape(-1)
        grad_scale = 1.0 / self.world        # local_bsize / global_bsize: SUM all-reduce -> global mean
        for _ in range(self.update_repeats):
            self.policy.forward(self.params, feat, goal, self.h_start, masks, T, N, self.ws_learn, hv=self.hv)
            ppo_loss_raw(self.hv, self.actions.view(-1), self.logp.view(-1), self.values[:T].reshape(-1),
                         self.returns[:T].reshape(-1), self.nadv.view(-1), self.A, grad_scale=grad_scale,
                         dhv=self.dhv, sums=self.sums)
            self.grads.zero_()
            self.policy.backward(self.params, feat, masks, T, N, self.ws_learn, self.dhv, None, self.grads)
            if self.world > 1:
                allreduce_flat(self.grads)                   # one flat 13.9 MB bucket over RCCL/xGMI
            self.opt.step(self.grads, lr=linear_decay_lr(self.base_lr, self.total_steps, self.lr_total_steps))

    def after_update(self):
        self.feat[0].copy_(self.feat[self.T])
        self.total_steps += self.T * self.N * self.world
        self.iter += 1

    def iteration(self):
        self.collect_rollout()
        self.compute_returns()
        self.update()
        self.after_update()

    def loss_info(self) -> Dict[str, float]:
        s = (self.sums / (self.T * self.N)).tolist()
        return {"action": s[0], "value": s[1], "entropy": s[2], "ratio": s[3],
                "ppo_total": s[0] + 0.5 * s[1] + 0.01 * s[2], "grad_norm": self.opt.grad_norm()}
