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

import os
import random

import torch

from . import _lib
from . import synthetic as syn
from .dist import (allreduce_flat, check_job_seed, collective_active, gather_actor_counts, global_minibatch_sizes,
                   minibatch_bounds)
from .encoder import AttentionPool, ClipTextEncoder, RN50Trunk, ViTEmbedder
from .policy import PolicyHandle
from .ppo import FlatAdam, linear_decay_lr, ppo_loss_raw


class SyntheticEnv:
    """N synthetic actors: a pool of pre-rendered, CLIP-normalised 224x224 frames in HBM,
    random goal ids, episode resets w.p. 1/100, RoboTHOR-style rewards."""

    def __init__(self, n_actors: int, T: int, device, seed: int, pool_steps: int = 4, res: int = 224,
                 frames_u8: bool = False, host: bool = False):
        self.N, self.T = n_actors, T
        self.host = host
        # frames_u8: raw uint8 frames (what the simulator renders); normalisation is then fused into the stem kernel.
        # Every actor has its OWN frame in every pool entry (n_actors distinct images per env step: no replicated
        # images flattering the cache hit rates of the stem); the pool entries are shifted copies of one another, so
        # consecutive steps differ.  The fp32 wire form is the same arithmetic as synthetic.normalize_rgb.
        base = syn.synthetic_rgb_u8(seed, n_actors, res)
        if not frames_u8:   # (on the device when the pool lives there: 256 frames take 6 s on the host)
            base = syn.normalize_rgb(base if host else base.to(device))
        frames = []
        for s in range(pool_steps):   # distinct frame batches so consecutive steps differ
            frames.append(base.roll(shifts=s + 1, dims=0).roll(shifts=7 * (s + 1), dims=2))
        # host=True: the pool stays in PINNED HOST memory (what the simulators hand over through the plugin contract);
        # the worker then streams each step's frames over PCIe on a copy stream (Worker._encode_slice)
        self.frames = (torch.stack(frames).contiguous().pin_memory() if host
                       else torch.stack(frames).to(device).contiguous())   # [P, N, R, R, 3] fp32 | uint8
        self.pool_steps = pool_steps
        masks = torch.cat([torch.ones(1, n_actors, 1), syn.synthetic_masks(seed + 1, T, n_actors)], 0)
        self.masks = masks.reshape(T + 1, n_actors).to(device).contiguous()
        self.goals = syn.synthetic_goals(seed + 2, (T + 1, n_actors)).to(device).contiguous()
        self.rewards = syn.synthetic_rewards(seed + 3, masks[1:]).reshape(T, n_actors).to(device).contiguous()
        self._k = 0

    def observe(self, actions_host: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Next batch of frames.  ``actions_host`` (the sampled actions, on the host) is what a simulator would consume;
        the synthetic env only requires that it has arrived."""
        f = self.frames[self._k % self.pool_steps]
        self._k += 1
        return f

    def observe_at(self, k: int, actions_host: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The batch ``observe()`` serves as its k-th call (per-slice env stepping: every slice asks for step k itself)."""
        return self.frames[k % self.pool_steps]


class _Slice:
    """Per-slice state: a contiguous group of actors with its own stream, encoder handle and buffers."""
    pass


class Worker:
    """One DD-PPO worker.  The actor batch is processed as ``n_slices`` independent slices (default 2), each with its
    own HIP stream, encoder handle, rollout feature buffer ``[T+1, n, S*S, C]``, policy workspaces and gradient
    bucket.  Slices only meet at the GAE/advantage normalisation and at the gradient sum, so the HBM-bound and
    MFMA-bound launches of one slice overlap the small / latency-bound launches (policy act step, GRU recurrence)
    of the other.  Per-actor arithmetic does not depend on the slicing (tests assert identical features/actions)."""

    def __init__(self, n_actors: int, T: int = 128, device="cuda:0", seed: int = 0, rank: int = 0, world: int = 1,
                 update_repeats: int = 4, lr: float = 3e-4, max_grad_norm: float = 0.5, gamma: float = 0.99,
                 tau: float = 0.95, encoder_sd=None, policy_sd=None, lr_total_steps: int = 300_000_000,
                 encoder_chunk: int = 0, encoder: str = "rn50", encoder_streams: int = 2, frames_u8: bool = False,
                 frames_host: bool = False, zeroshot: bool = False, text_sd=None, goal_tokens=None,
                 num_mini_batch: int = 1, sync_actions: bool = False, force_allreduce: bool = False,
                 overlap_allreduce: bool = True):
        """``zeroshot=True`` (BASELINE config 5, readme_files/zeroshot_objectnav.md): the observation is the CLIP image
        EMBEDDING (RN50 trunk + AttentionPool2d, 1024-d), the goal is the frozen CLIP text embedding of its prompt
        (text tower run once -> [12, 1024] table) and the policy is the fusion=1 variant (GRU + heads trainable)."""
        self.lib = _lib.load()
        self.zeroshot, self._text_sd, self._goal_tokens = zeroshot, text_sd, goal_tokens
        # sync_actions: the action-synchronous order of a real vectorised env ([U] VectorSampledTasks.step(actions)): every
        # env step the sampled actions of ALL actors are copied to the host and waited for before observe() serves the next
        # frames.  Default off: the synthetic env does not read the actions and the host issues ahead.
        self.sync_actions = sync_actions
        # force_allreduce: run the flat-bucket collective also at world size 1 (RCCL first-contact check on a 1-GPU box)
        self.force_allreduce = force_allreduce
        # overlap_allreduce (when there is a collective at all): the GRU + heads section of the flat bucket (92 % of its bytes,
        # final first: ec_policy_backward3) is summed over ranks on a communication stream UNDER the rest of the backward;
        # only the goal encoder's 1.1 MB is reduced after it.  False: one 13.9-MB all-reduce after the backward
        self.overlap_allreduce = overlap_allreduce
        # [U] allenact RolloutStorage.recurrent_generator(num_mini_batch): contiguous sampler ranges, shuffled order
        assert 1 <= num_mini_batch <= n_actors, "num_mini_batch must not exceed the number of samplers"
        self.num_mini_batch = num_mini_batch
        # ONE shuffle stream for all ranks: every rank visits the same minibatch range at the same optimiser step, so the
        # SUM all-reduce with the fixed 1/world scale is the global minibatch mean also when N % num_mini_batch != 0
        # (ranges of different sizes) -- with per-rank streams it would be a mean of differently-sized means
        self._mb_rng = random.Random(seed)                  # (`seed` is the job's seed: identical on every rank)
        self.dev = self.device = torch.device(device)
        if self.dev.index is None:
            self.dev = self.device = torch.device("cuda", torch.cuda.current_device())
        check_job_seed(seed, world, device=self.dev)        # ... which is ENFORCED when a process group is up (on THIS worker's GPU)
        # every rank's actor count: the gradient scale is local / GLOBAL minibatch size ([U] backprop_step), and the global size
        # is the sum over the ranks -- shards need not be equal (dist.shard_actors(10, r, 3) = 4, 3, 3)
        self.shard_counts = gather_actor_counts(n_actors, world, device=self.dev)
        assert all(num_mini_batch <= c for c in self.shard_counts), "num_mini_batch must not exceed the smallest shard"
        self._mb_global = global_minibatch_sizes(self.shard_counts, num_mini_batch)
        self._init(n_actors, T, seed, rank, world, update_repeats, lr, max_grad_norm, gamma, tau, encoder_sd, policy_sd,
                   lr_total_steps, encoder_chunk, encoder, encoder_streams, frames_u8, frames_host)

    @_lib.on_device
    def _init(self, n_actors, T, seed, rank, world, update_repeats, lr, max_grad_norm, gamma, tau, encoder_sd, policy_sd,
              lr_total_steps, encoder_chunk, encoder, encoder_streams, frames_u8, frames_host=False):
        self.N, self.T, self.rank, self.world = n_actors, T, rank, world
        self.update_repeats, self.gamma, self.tau = update_repeats, gamma, tau
        self.base_lr, self.lr_total_steps = lr, lr_total_steps
        self.encoder = encoder
        # two slices (each with its own act -> encode chain on its own stream) from 48 actors on: measured round 3 at
        # 32 / 48 / 64 actors, one vs two slices: 26.8 / 29.0 / 35.0 k vs 26.1 / 31.8 / 36.8 k env-frames/s
        two_min = int(os.environ.get("EC_TWO_SLICE_MIN", "48"))          # (experiment switch for the rule above)
        ns = encoder_streams if (encoder_streams > 1 and n_actors >= two_min and n_actors % encoder_streams == 0) else 1
        self.ns = ns
        n = n_actors // ns
        d = self.dev
        pools = [None] * ns
        if encoder in ("rn50", "rn50x16"):
            # "rn50x16": CLIP RN50x16's tower (width 96, layers (6, 8, 18, 8)) on the plugin's 224 x 224 frames -> a
            # 3072 x 7 x 7 map ([U] ClipResNetPreprocessor's second model type; functional path, not tuned)
            sd = encoder_sd if encoder_sd is not None else (
                syn.rn50_visual_state_dict(0) if encoder == "rn50" else
                syn.rn50_visual_state_dict(0, width=96, layers=(6, 8, 18, 8), output_dim=768, heads=48))
            share = os.environ.get("EC_SHARE_WEIGHTS", "1") != "0"
            encs = [RN50Trunk(sd, device=d, chunk=encoder_chunk)]
            encs += [RN50Trunk(sd, device=d, chunk=encoder_chunk, weights_from=encs[0] if share else None) for _ in range(ns - 1)]
            self.S, self.C = encs[0].out_spatial, encs[0].out_channels
            if self.zeroshot:
                pools = [AttentionPool(sd, device=d)]
                pools += [AttentionPool(sd, device=d, weights_from=pools[0] if share else None) for _ in range(ns - 1)]
                self.trunk_S, self.trunk_C = self.S, self.C
                self.S, self.C = 1, pools[0].out_dim
        elif encoder == "vit":
            # BASELINE config 3 (builder-defined fusion, SURVEY.md §8d note): ClipViTEmbedder tokens, CLS dropped,
            # the 49 patch tokens are the 7x7 channels-last "feature map" [n,49,768] of the goal encoder
            sd = encoder_sd if encoder_sd is not None else syn.vit_visual_state_dict(0)
            share = os.environ.get("EC_SHARE_WEIGHTS", "1") != "0"
            encs = [ViTEmbedder(sd, device=d)]
            encs += [ViTEmbedder(sd, device=d, weights_from=encs[0] if share else None) for _ in range(ns - 1)]
            self.S, self.C = 7, encs[0].D
        else:
            raise ValueError(encoder)
        pkw = dict(in_channels=self.C, spatial=self.S)
        if self.zeroshot:
            assert encoder == "rn50", "the zero-shot variant uses the CLIP-RN50 image embedding"
            pkw["fusion"] = 1
        self.policy = PolicyHandle(**pkw)
        if self.zeroshot:
            # goal table: the CLIP text tower over the 12 goal prompts, once, L2-normalised; frozen
            tsd = self._text_sd if self._text_sd is not None else syn.text_state_dict(0)
            tok = self._goal_tokens if self._goal_tokens is not None else syn.synthetic_tokens(
                7, 12, context_length=tsd["positional_embedding"].shape[0], vocab_size=tsd["token_embedding.weight"].shape[0])
            self.goal_table = ClipTextEncoder(tsd, device=d).goal_table(tok).contiguous()
            self.policy.set_goal_table(self.goal_table)
        self.H, self.A = self.policy.H, self.policy.A
        self.params = self.policy.flatten(policy_sd if policy_sd is not None else syn.policy_state_dict(0, **pkw), d)
        self.grads = torch.zeros_like(self.params)
        self.rec = self.policy.recurrent_section()          # GRU + heads: the part of the bucket whose gradients are final first
        self.opt = FlatAdam(self.params, lr=lr, max_grad_norm=max_grad_norm)
        N, S2 = n_actors, self.S * self.S
        # [T, N] rollout scalars (global; tiny)
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
        self.stats = torch.zeros(2, dtype=torch.float64, device=d)
        self.sums = torch.zeros(4, dtype=torch.float64, device=d)
        self.env = SyntheticEnv(N, T, d, seed=1000 + rank, frames_u8=frames_u8, host=frames_host)
        self.slices: List[_Slice] = []
        # Streams.  The HIP runtime has FOUR hardware queues and binds a stream to one of them at its first submission; two
        # streams on one queue run one after the other (a slice pair that shared a queue serialised the two encoder launches:
        # 48 instead of 63 k; a copy stream on the other slice's compute queue: 59 -> 42 k; tools/sync_probe.py, h2d_probe.py).
        # So the slices' streams, the communication stream of the overlapped all-reduce (only where there is a collective) and
        # the slices' copy streams (frames in host memory) come from ONE pool that _lib.concurrent_streams has verified pairwise
        # concurrent; when that is more than the queues allow, the copy streams, then the communication stream, go unverified.
        n_own = ns if ns > 1 else 0
        n_comm = 1 if (world > 1 or self.force_allreduce) else 0
        n_copy = ns if frames_host else 0
        plain = lambda k: [torch.cuda.Stream(device=d) for _ in range(k)]   # noqa: E731
        pool = None
        for take_comm, take_copy in ((n_comm, n_copy), (n_comm, 0), (0, 0)):
            want = n_own + take_comm + take_copy
            try:
                pool = (_lib.concurrent_streams(want, d) if want > 1 else plain(want))
            except RuntimeError:
                if take_comm == 0 and take_copy == 0:
                    raise
                continue
            if (take_comm, take_copy) != (n_comm, n_copy):
                import warnings
                warnings.warn(f"{n_own + n_comm + n_copy} streams exceed the runtime's hardware queues: only {want} are verified concurrent")
            pool = pool + plain(n_comm - take_comm + n_copy - take_copy)
            order = ["own"] * n_own + ["comm"] * take_comm + ["copy"] * take_copy + ["comm"] * (n_comm - take_comm) + ["copy"] * (n_copy - take_copy)
            break
        by = lambda kind: [s_ for s_, k_ in zip(pool, order) if k_ == kind]   # noqa: E731
        slice_streams = by("own") if n_own else [None]
        copy_streams = by("copy")
        self.comm_stream = by("comm")[0] if n_comm else torch.cuda.Stream(device=d)
        for i in range(ns):
            sl = _Slice()
            sl.o, sl.n, sl.enc, sl.pool = i * n, n, encs[i], pools[i]
            sl.stream = slice_streams[i]
            # zero-shot: the rollout buffer holds fp32 image embeddings [T+1, n, 1, 1024]; else bf16 feature maps
            sl.feat = torch.empty((T + 1, n, S2, self.C), dtype=torch.float32 if self.zeroshot else torch.bfloat16, device=d)
            sl.trunk_out = (torch.empty((n, self.trunk_S, self.trunk_S, self.trunk_C), dtype=torch.bfloat16, device=d)
                            if self.zeroshot else None)
            sl.tok = (torch.empty((n, encs[i].L, encs[i].D), dtype=torch.bfloat16, device=d) if encoder == "vit" else None)
            sl.ws_act = torch.empty(self.policy.workspace_bytes(1, n, False), dtype=torch.uint8, device=d)
            sl.act_tables_valid = False     # weight-derived tables in ws_act (rebuilt by the first act step after an update)
            sl.ws_learn = torch.empty(self.policy.workspace_bytes(T, n, True), dtype=torch.uint8, device=d)
            sl.hv = torch.empty((T * n, self.A + 1), dtype=torch.float32, device=d)
            sl.dhv = torch.empty_like(sl.hv)
            sl.grads = self.grads if ns == 1 else torch.zeros_like(self.params)
            sl.rec_ready = torch.cuda.Event()
            sl.rec_ready.record()          # (torch creates the hipEvent_t at the first record: the library needs the handle)
            sl.sums = torch.zeros(4, dtype=torch.float64, device=d)
            sl.goal = sl.masks = sl.actions = sl.logp = sl.old_v = sl.ret = sl.nadv = None
            if frames_host:   # double-buffered device staging of the slice's frames + its own copy stream (SDMA)
                fshape = (n,) + tuple(self.env.frames.shape[2:])
                sl.stage = [torch.empty(fshape, dtype=self.env.frames.dtype, device=d) for _ in range(2)]
                sl.copy_stream = copy_streams[i]
                sl.copied = [torch.cuda.Event() for _ in range(2)]
                sl.consumed = [None, None]
                sl.k = 0
            self.slices.append(sl)
        self.encode_frames = n                    # frames per timed encoder launch
        # two launches in flight with >= 128 frames each: the 8-wave conv kernel from 50 tiles on -- a property of THIS
        # worker's encoder handles (ec_rn50/vit_set_conv8_min_tiles), not of the process.  With 64..127 frames per slice:
        # from 75 tiles on (round 3, after the long-segment 128-wide tiles: 128 actors 46.6-47.2 k with the default 150,
        # 48.2 k with 50, 49.2-49.3 k with 60 / 75 / 90; two slices of 32 frames keep the default: 39.1 vs 37.6 k)
        self._conv8_min_tiles = (50 if n >= 128 else (75 if n >= 64 else 0)) if ns == 2 else 0
        for e in encs:
            e.set_conv8_min_tiles(self._conv8_min_tiles)
        self._act_fused = os.environ.get("EC_ACT_FUSED_SAMPLE", "1") != "0" and self.A + 1 <= 8   # (A/B switch; > 7 actions: two calls)
        self.seed = seed + 7919 * rank
        self.total_steps = 0
        self.iter = 0
        self.trunk_events: List = []             # (start, end) HIP event pairs around the encoder launches
        self.update_events: List = []            # ... and around the update phase (GAE excluded), one pair per timed iteration
        self.time_trunk = False
        # first observation of the first rollout
        rgb = self.env.observe()
        for sl in self.slices:
            self._encode_slice(sl, rgb, 0)
        torch.cuda.synchronize(d)

    # ---- helpers ------------------------------------------------------------------------------
    @property
    def feat(self) -> torch.Tensor:
        """[T+1, N, S*S, C] view for tests / inspection (concatenates the slices)."""
        return self.slices[0].feat if self.ns == 1 else torch.cat([sl.feat for sl in self.slices], dim=1)

    @property
    def enc_streams(self):
        return [sl.stream for sl in self.slices if sl.stream is not None]

    def _on(self, sl):
        return torch.cuda.stream(sl.stream) if sl.stream is not None else _NullCtx()

    def _fork(self):
        if self.ns > 1:
            cur = torch.cuda.current_stream()
            for sl in self.slices:
                sl.stream.wait_stream(cur)

    def _join(self):
        if self.ns > 1:
            cur = torch.cuda.current_stream()
            for sl in self.slices:
                cur.wait_stream(sl.stream)

    # ---- HOT LOOP A ---------------------------------------------------------------------------
    def _encode_slice(self, sl, rgb: torch.Tensor, t: int):
        src = rgb[sl.o:sl.o + sl.n]
        if self.env.host:
            # H2D of this slice's frames on its copy stream, overlapping whatever the other slice is computing; the
            # staging buffer is reused only after the encoder launch that read it two steps ago has finished
            i = sl.k & 1
            sl.k += 1
            cur = torch.cuda.current_stream()
            with torch.cuda.stream(sl.copy_stream):
                if sl.consumed[i] is not None:
                    sl.copy_stream.wait_event(sl.consumed[i])
                sl.stage[i].copy_(src, non_blocking=True)
                sl.copied[i].record(sl.copy_stream)
            cur.wait_event(sl.copied[i])
            src = sl.stage[i]
        timed = self.time_trunk
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream())
        if self.zeroshot:
            (sl.enc.forward_u8 if src.dtype == torch.uint8 else sl.enc.forward)(src, sl.trunk_out)
            sl.pool.forward(sl.trunk_out, sl.feat[t].view(sl.n, self.C))     # AttentionPool2d -> rollout slice
        elif self.encoder in ("rn50", "rn50x16"):
            # the last conv writes straight into the rollout slice
            (sl.enc.forward_u8 if src.dtype == torch.uint8 else sl.enc.forward)(src, sl.feat[t])
        else:
            sl.enc.forward(src, sl.tok)
            sl.feat[t].copy_(sl.tok[:, 1:, :])        # drop CLS: [n,49,768] channels-last rows
        if timed:
            e1.record(torch.cuda.current_stream())
            self.trunk_events.append((e0, e1))
        if self.env.host:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            sl.consumed[i] = ev

    def _act_slice(self, sl, t: int, sample: bool = True):
        """Policy act step (T=1, no grad) for the slice's actors on the current stream."""
        o, n, sp = sl.o, sl.n, _lib.stream_ptr()
        rs = slice(o, o + n)
        h_in, h_out = (self.h, self.h_next) if (t & 1) == 0 else (self.h_next, self.h)   # ping-pong by step parity
        if sample and self._act_fused:
            # forward + CategoricalDistr.sample / log_prob in one call: the heads launch samples (ec_policy_act, one launch less)
            self.policy.act(self.params, sl.feat[t], self.env.goals[t][rs], h_in[rs], self.env.masks[t][rs], n, sl.ws_act,
                            self.hv_act[rs], h_out[rs], self.actions[t][rs], self.logp[t][rs], self.values[t][rs], self.seed,
                            self.iter * (self.T + 1) + t, o, reuse_tables=sl.act_tables_valid)
            sl.act_tables_valid = True
            return
        self.policy.forward(self.params, sl.feat[t], self.env.goals[t][rs], h_in[rs], self.env.masks[t][rs], 1, n,
                            sl.ws_act, hv=self.hv_act[rs], h_final=h_out[rs], for_backward=False,
                            reuse_tables=sl.act_tables_valid)      # (E1 depends on the parameters only)
        sl.act_tables_valid = True
        if sample:
            _lib.check(self.lib.ec_sample_actions(self.hv_act[rs].data_ptr(), self.actions[t][rs].data_ptr(),
                                                  self.logp[t][rs].data_ptr(), self.values[t][rs].data_ptr(), n, self.A,
                                                  self.seed, self.iter * (self.T + 1) + t, o, sp), "ec_sample_actions")
        else:   # bootstrap value of the last observation; memory is NOT advanced
            self.values[t][rs].copy_(self.hv_act[rs, self.A])

    @_lib.on_device
    def collect_rollout(self):
        T = self.T
        self.h_start.copy_(self.h)
        self._fork()
        if self.sync_actions:
            return self._collect_rollout_sync()
        for t in range(T):
            rgb = self.env.observe()      # env.step(actions[t]) happens here in the real system (fp32 NHWC frames)
            for sl in self.slices:
                with self._on(sl):
                    self._act_slice(sl, t)
                    self._encode_slice(sl, rgb, t + 1)
        for sl in self.slices:
            with self._on(sl):
                self._act_slice(sl, T, sample=False)
        self._join()
        if (T & 1) == 1:   # after an odd number of advancing steps the live memory sits in h_next
            self.h, self.h_next = self.h_next, self.h

    def _collect_rollout_sync(self):
        """The action-synchronous orders: one host round trip per env step sits between act(t) and encode(t+1), as in the
        real engine.

        ``sync_actions=True`` ("all"): ONE vectorised env for all actors ([U] ``VectorSampledTasks.step(actions)``): act(t)
        on every slice -> actions[t] D2H, WAITED FOR -> env.step -> encode(t+1) on every slice.  The slices then run in
        lock step and the chip idles over the round trip only: 0.99 x the free-running rate at 256 actors (measured round 5;
        the 0.77 x of rounds 3-4 was two slice streams sharing one hardware queue: _lib.concurrent_streams).

        ``sync_actions="slice"``: one vectorised env PER SLICE (two ``VectorSampledTasks`` groups): the host waits for the
        actions of slice s only, steps that slice's env and issues its encode(t+1) + act(t+1) -- while the other slice's
        encoder is still running.  Same per-actor arithmetic and the same number of round trips per actor, but the two
        slices stay out of phase as they do in the free-running order."""
        T = self.T
        if getattr(self, "_actions_host", None) is None:
            self._actions_host = torch.empty((self.N,), dtype=torch.int64).pin_memory()
        k0 = self.env._k

        def d2h(sl, t):
            self._actions_host[sl.o:sl.o + sl.n].copy_(self.actions[t][sl.o:sl.o + sl.n], non_blocking=True)

        def wait(sl):
            (sl.stream if sl.stream is not None else torch.cuda.current_stream()).synchronize()

        if self.sync_actions == "slice":
            for sl in self.slices:
                with self._on(sl):
                    self._act_slice(sl, 0)
                    d2h(sl, 0)
            for t in range(T):
                for sl in self.slices:
                    wait(sl)                                                 # this slice's actions[t] are on the host
                    rgb = self.env.observe_at(k0 + t, self._actions_host)   # env.step of this slice's samplers
                    with self._on(sl):
                        self._encode_slice(sl, rgb, t + 1)
                        if t + 1 < T:
                            self._act_slice(sl, t + 1)
                            d2h(sl, t + 1)
                        else:
                            self._act_slice(sl, T, sample=False)
            self.env._k = k0 + T
        else:
            for t in range(T):
                for sl in self.slices:
                    with self._on(sl):
                        self._act_slice(sl, t)
                        d2h(sl, t)
                for sl in self.slices:
                    wait(sl)
                rgb = self.env.observe(self._actions_host)          # env.step(actions[t])
                for sl in self.slices:
                    with self._on(sl):
                        self._encode_slice(sl, rgb, t + 1)
            for sl in self.slices:
                with self._on(sl):
                    self._act_slice(sl, T, sample=False)
        self._join()
        if (T & 1) == 1:
            self.h, self.h_next = self.h_next, self.h

    # ---- parameters ---------------------------------------------------------------------------
    def invalidate_act_tables(self):
        """The act workspaces cache weight-derived tables (the goal-embedding table E1, the re-ordered weight_ih).  Call
        this after ANY write to ``self.params`` that does not go through ``update()`` (checkpoint restore, parameter
        broadcast, an external optimiser)."""
        for sl in self.slices:
            sl.act_tables_valid = False

    def set_params(self, flat: torch.Tensor):
        """Replace the flat parameter vector (e.g. a restored checkpoint) and drop everything derived from it."""
        self.params.copy_(flat.to(self.params.device, self.params.dtype).view_as(self.params))
        self.invalidate_act_tables()

    @_lib.on_device
    def compute_returns(self):
        _lib.check(self.lib.ec_gae(self.env.rewards.data_ptr(), self.values.data_ptr(), self.env.masks.data_ptr(),
                                   self.returns.data_ptr(), self.adv.data_ptr(), self.nadv.data_ptr(),
                                   self.stats.data_ptr(), self.T, self.N, self.gamma, self.tau, 1e-5,
                                   _lib.stream_ptr()), "ec_gae")

    # ---- HOT LOOP B ---------------------------------------------------------------------------
    def _gather_slice_batches(self):
        """Contiguous [T*n] copies of the slice's columns of the [T, N] rollout scalars (once per iteration)."""
        T = self.T
        for sl in self.slices:
            rs = slice(sl.o, sl.o + sl.n)
            c = lambda x: x[:T, rs].reshape(-1).contiguous()
            sl.goal, sl.masks = c(self.env.goals), c(self.env.masks)
            sl.actions, sl.logp, sl.old_v = c(self.actions), c(self.logp), c(self.values)
            sl.ret, sl.nadv = c(self.returns), c(self.nadv)

    def minibatch_ranges(self):
        """[U] allenact ``RolloutStorage.recurrent_generator``: the samplers (actors) are cut at
        ``round(linspace(0, N, num_mini_batch + 1))`` into contiguous ranges which are visited in a shuffled order;
        every range keeps its full T-step sequences (the GRU needs them)."""
        inds = minibatch_bounds(self.N, self.num_mini_batch)      # == np.round(np.linspace(0, N, M + 1))
        # (s0, s1, samplers of this range over ALL ranks); the shuffle permutation depends on the list's length only
        pairs = [(s0, s1, g) for s0, s1, g in zip(inds[:-1], inds[1:], self._mb_global)]
        self._mb_rng.shuffle(pairs)
        return pairs

    def _max_partial_range(self, sl) -> int:
        inds = minibatch_bounds(self.N, self.num_mini_batch)
        best = 1
        for s0, s1 in zip(inds[:-1], inds[1:]):
            a, b = max(s0, sl.o) - sl.o, min(s1, sl.o + sl.n) - sl.o
            if b > a and not (a == 0 and b == sl.n):
                best = max(best, b - a)
        return best

    def _gather_part(self, sl, a: int, b: int):
        """Contiguous [T * (b - a)] batch of actors [a, b) of slice ``sl`` (slice-local indices) in the slice's staging
        buffers; the whole slice is used in place."""
        T = self.T
        if a == 0 and b == sl.n:
            return (sl.feat[:T].view(T * sl.n, self.S * self.S, self.C), sl.goal, sl.masks, sl.actions, sl.logp, sl.old_v,
                    sl.ret, sl.nadv)
        m = b - a
        if getattr(sl, "feat_mb", None) is None:                   # staging for partial-slice minibatches (allocated on first use),
            mmax = self._max_partial_range(sl)                      # sized to the largest PARTIAL range, not the whole slice
            sl.feat_mb = torch.empty((T, mmax, self.S * self.S, self.C), dtype=sl.feat.dtype, device=self.dev)
        fm = sl.feat_mb.view(-1)[:T * m * self.S * self.S * self.C].view(T, m, self.S * self.S, self.C)
        fm.copy_(sl.feat[:T, a:b])
        c = lambda x: x.view(T, sl.n)[:, a:b].reshape(-1).contiguous()
        return (fm.view(T * m, self.S * self.S, self.C), c(sl.goal), c(sl.masks), c(sl.actions), c(sl.logp), c(sl.old_v),
                c(sl.ret), c(sl.nadv))

    def _sum_parts(self, parts, sec: slice):
        """self.grads[sec] = sum of the slices' gradient buckets over ``sec`` (one slice: its bucket IS self.grads)."""
        if self.ns == 1:
            return
        if len(parts) == 1:
            self.grads[sec].copy_(parts[0][0].grads[sec])
            return
        torch.add(parts[0][0].grads[sec], parts[1][0].grads[sec], out=self.grads[sec])
        for (sl, _, _) in parts[2:]:
            self.grads[sec].add_(sl.grads[sec])

    def _other_sections(self):
        """What ``self.rec`` leaves of the flat bucket (the goal encoder's tensors; two pieces with the dual encoder)."""
        n = self.grads.numel()
        return [sec for sec in (slice(0, self.rec.start), slice(self.rec.stop, n)) if sec.stop > sec.start]

    @_lib.on_device
    def update(self):
        T = self.T
        collective = collective_active(self.world, self.force_allreduce)
        self._gather_slice_batches()
        for _ in range(self.update_repeats):
            for (s0, s1, nmb_global) in self.minibatch_ranges():
                # the minibatch's actors, slice by slice (each part on its slice's stream)
                parts = []
                for sl in self.slices:
                    a, b = max(s0, sl.o) - sl.o, min(s1, sl.o + sl.n) - sl.o
                    if b > a:
                        parts.append((sl, a, b))
                early = collective and self.overlap_allreduce
                if early:                      # (the previous optimiser step read self.grads on the main stream)
                    self.comm_stream.wait_stream(torch.cuda.current_stream())
                self._fork()
                for (sl, a, b) in parts:
                    with self._on(sl):
                        m = b - a
                        feat, goal, masks, actions, logp, old_v, ret, nadv = self._gather_part(sl, a, b)
                        hv, dhv = sl.hv[:T * m], sl.dhv[:T * m]
                        # d(total)/d(hv) carries 1/B_part inside the loss kernel: rescale to the mean over the WHOLE
                        # local minibatch (m / nmb), then local_bsize / global_bsize (nmb / nmb_global: the sum of the
                        # ranks' sizes of this range, equal shards or not) for the SUM all-reduce = the global mean gradient
                        grad_scale = m / nmb_global
                        # (staggering the slices -- slice 1's forward starting when slice 0's is done, so that wide GEMMs
                        #  run beside the other slice's GRU recurrence -- was measured in round 3: 53 -> 58 ms per update;
                        #  the step kernels wait for CUs the GEMM workgroups hold)
                        self.policy.forward(self.params, feat, goal, self.h_start[sl.o + a:sl.o + b], masks, T, m,
                                            sl.ws_learn, hv=hv)
                        ppo_loss_raw(hv, actions, logp, old_v, ret, nadv, self.A, grad_scale=grad_scale, dhv=dhv,
                                     sums=sl.sums)
                        sl.grads.zero_()
                        self.policy.backward(self.params, feat, masks, T, m, sl.ws_learn, dhv, None, sl.grads,
                                             recurrent_ready=sl.rec_ready if early else None)
                if early:
                    # GRU + heads section: summed over the slices and over the ranks on the communication stream, behind the
                    # events the backwards record -- i.e. under the goal encoder's backward (dx GEMM, tail, dW1) of every slice
                    with torch.cuda.stream(self.comm_stream):
                        for (sl, _, _) in parts:
                            self.comm_stream.wait_event(sl.rec_ready)
                        self._sum_parts(parts, self.rec)
                        allreduce_flat(self.grads[self.rec], force=self.force_allreduce)
                self._join()
                self._loss_parts = [(sl, b - a) for (sl, a, b) in parts]
                if early:
                    # the goal encoder's section(s): after the backwards, on the main stream (1.1 MB)
                    for sec in self._other_sections():
                        self._sum_parts(parts, sec)
                        allreduce_flat(self.grads[sec], force=self.force_allreduce)
                    torch.cuda.current_stream().wait_stream(self.comm_stream)
                else:
                    self._sum_parts(parts, slice(0, self.grads.numel()))
                    if collective:
                        allreduce_flat(self.grads, force=self.force_allreduce)   # one flat 13.9 MB bucket over RCCL/xGMI
                self.opt.step(self.grads, lr=linear_decay_lr(self.base_lr, self.total_steps, self.lr_total_steps))
                self.invalidate_act_tables()

    @_lib.on_device
    def after_update(self):
        for sl in self.slices:
            sl.feat[0].copy_(sl.feat[self.T])
        self.total_steps += self.T * sum(self.shard_counts)
        self.iter += 1

    def iteration(self):
        self.collect_rollout()
        self.compute_returns()
        timed = self.time_trunk          # (bench.py: HIP events around the update phase on the main stream; no host sync)
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.device(self.dev):
                e0.record(torch.cuda.current_stream())
        self.update()
        if timed:
            with torch.cuda.device(self.dev):
                e1.record(torch.cuda.current_stream())
            self.update_events.append((e0, e1))
        self.after_update()

    def loss_info(self) -> Dict[str, float]:
        parts = getattr(self, "_loss_parts", None) or [(sl, sl.n) for sl in self.slices]   # the last minibatch
        tot = sum(sl.sums for sl, _ in parts)
        s = (tot / (self.T * sum(m for _, m in parts))).tolist()
        return {"action": s[0], "value": s[1], "entropy": s[2], "ratio": s[3],
                "ppo_total": s[0] + 0.5 * s[1] + 0.01 * s[2], "grad_norm": self.opt.grad_norm()}


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
