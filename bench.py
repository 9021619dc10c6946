#!/usr/bin/env python
"""Headline benchmark: env-frames/sec of the DD-PPO hot path on MI355X.

One "step" = one full training iteration of one worker per GPU over synthetic
frames already resident in HBM:
  T=128 x [CLIP-RN50 encode N frames + policy act + sample] + GAE
  + 4 x [policy forward over [T,N] + PPO loss + backward + flat-bucket
         all-reduce + clip + Adam]
value = (T * N_total * steps) / wall time, max over ranks, barrier+sync bracketed.

    python bench.py                            # 1 GPU, 256 actors, K=2 W=1
    python bench.py --gpus 8                   # spawns its own 8 ranks (file-store rendezvous, RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8  # ... or is launched one rank per GPU by torchrun

Scaling: BASELINE.json's metric is "256 actors, 1/2/4/8 MI355X", i.e. the SAME 256 actors sharded over the GPUs
(strong scaling: 256/N actors per GPU; `--actors-total 512` is config 4 = 64 per GPU at N=8).  For N > 1 the line
also carries a `weak` object (256 actors PER GPU, the DD-PPO convention) measured in the same process.  Actors shard
with no data-path collective; the only exchange is one SUM all-reduce of the 13.9 MB flat policy-gradient bucket
per optimiser step (RCCL over xGMI), timed separately per rank (`allreduce_ms_per_rank`).
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

TRUNK_MAC_PER_FRAME = 5_367_226_368          # SURVEY.md §8d (RN50 trunk, 224x224)
VIT_MAC_PER_FRAME = 4_050_683_904            # SURVEY.md §8d (ViT-B/32, 11 of 12 blocks)
RN50X16_MAC_PER_FRAME = 23_859_892_224       # RN50x16 trunk (width 96, layers (6, 8, 18, 8)) on the plugin's 224x224 frames
ATTNPOOL_MAC_PER_FRAME = 425_984_000 + 49 * 2048 * 2048   # CLS-only query + k/v projections
POLICY_ACT_MAC = 16_846_336
POLICY_UPDATE_MAC = 150_775_808              # 4 x (fwd + bwd)
ZS_POLICY_ACT_MAC = 1536 * 1024 + 1536 * 512 + 7 * 512            # zero-shot policy: GRU (1024 -> 512) + heads
ZS_POLICY_UPDATE_MAC = 4 * (ZS_POLICY_ACT_MAC + 1536 * 1024 + 2 * 1536 * 512 + 2 * 7 * 512)
MFMA_BF16_PEAK_TFLOPS = 2500.0               # MI355X_MICROARCH.md: dense bf16 MFMA
HBM_ACHIEVABLE_TBS = 6.29                    # MI355X_MICROARCH.md chip table: achievable HBM3E bandwidth (8.0 TB/s spec)
POLICY_FLAT_PARAMS = 3_480_775
# PMC-measured HBM bytes per encoder launch: written by tools/pmc_summary.py from separate rocprofv3 --pmc passes
# (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), keyed by the library's launch-plan hash
TRAFFIC_FILES = {"rn50": os.path.join(ROOT, "profiles", "trunk_hbm_traffic.json"),
                 "vit": os.path.join(ROOT, "profiles", "vit_hbm_traffic.json")}


def _usable_cpus() -> int:
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:   # cgroup v2 CPU quota, if any
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(p))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(n_actors: int, T: int, update_repeats: int, budget_s: float = 30.0):
    """The CPU oracle ("port" of the reference's torch path) timed on this host's cores on a BOUNDED
    sample of the same workload (fewer actors / shorter rollout; per-frame cost is batch-insensitive).
    The thread count is calibrated first (oversubscribing torch's CPU conv is catastrophically slow)."""
    from embodied_clip_amd import synthetic as syn
    from oracle import clip_resnet as ocr
    from oracle import iteration as oit
    enc_sd, pol_sd = syn.rn50_visual_state_dict(0), syn.policy_state_dict(0)
    frames = syn.synthetic_rgb(1000, n_actors).unsqueeze(0)
    usable = _usable_cpus()
    best_threads, best_fps = 1, 0.0
    for th in (8, 16, 32, 64, 128):
        if th > usable and th != 8:
            break
        torch.set_num_threads(min(th, usable))
        ocr.clip_resnet_preprocessor(frames[0][:2], enc_sd)             # warm the thread pool
        t0 = time.perf_counter()
        ocr.clip_resnet_preprocessor(frames[0], enc_sd)
        fps = n_actors / (time.perf_counter() - t0)
        if fps > best_fps:
            best_threads, best_fps = min(th, usable), fps
        elif fps < 0.7 * best_fps:
            break
    torch.set_num_threads(best_threads)
    # bound the sample: encoder dominates (T+1 encodes of n_actors frames)
    est = (T + 1) * n_actors / max(best_fps, 1e-3) * 1.3
    while est > budget_s and T > 2:
        T //= 2
        est = (T + 1) * n_actors / max(best_fps, 1e-3) * 1.3
    masks = torch.cat([torch.ones(1, n_actors, 1), syn.synthetic_masks(1001, T, n_actors)], 0)
    goals = syn.synthetic_goals(1002, (T + 1, n_actors))
    rewards = syn.synthetic_rewards(1003, masks[1:])
    info = oit.run_iteration(enc_sd, pol_sd, frames, goals, masks, rewards, T, n_actors, update_repeats)
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": round(info["frames_per_s"], 2), "unit": "env-frames/s", "cores": best_threads,
            "kind": "port",
            "sample": f"oracle (torch-CPU fp32) full iteration, {n_actors} actors x rollout {T}, "
                      f"{update_repeats} update epochs = {info['frames']} frames in {info['seconds']:.1f} s "
                      f"({best_threads} threads, best of a thread-count calibration; {usable} usable CPUs)",
            "cpu_model": cpu_model}


def measured_traffic(encoder: str, plan_hash, frames_per_launch: int):
    """HBM bytes per encoder launch from the committed PMC summary (profiles/*.json), scaled to the launch size.
    Fails LOUDLY when the summary was measured on a different launch plan / library version (stale evidence);
    returns (None, note) when no summary exists for this encoder."""
    path = TRAFFIC_FILES[encoder]
    if not os.path.exists(path):
        return None, f"no PMC summary at {os.path.relpath(path, ROOT)} (run tools/pmc_collect.sh + tools/pmc_summary.py)", None
    rec = json.load(open(path))
    alt = rec.get("single_launch_256") or {}
    if plan_hash is not None and rec.get("plan_hash") not in (None, "", plan_hash) and alt.get("plan_hash") == plan_hash:
        # a one-slice worker (fewer than EC_TWO_SLICE_MIN actors per GPU: the strong-scaling operating points) runs the plan with the
        # library's default dispatch threshold, which the summary holds as its 256-frame single-launch record
        per_frame = alt["hbm_bytes_per_launch"] / 256.0
        return per_frame * frames_per_launch, (
            f"bytes per launch = PMC-measured {alt['hbm_bytes_per_launch'] / 1e9:.2f} GB per 256-frame launch of THIS plan "
            f"({os.path.relpath(path, ROOT)}: single_launch_256; same counters and corrections as the engine plan's record) scaled per frame to "
            f"{frames_per_launch} frames -- an approximation below 128 frames per launch, where layer 3 runs as separate conv launches "
            "and the per-launch weight reads weigh more"), rec
    if plan_hash is not None and rec.get("plan_hash") not in (None, "", plan_hash):
        # stale evidence is never quoted: the line carries traffic = null and says why (and how to refresh it)
        msg = (f"{os.path.relpath(path, ROOT)} is STALE: measured on launch plan {rec.get('plan_hash')}, the library now runs "
               f"plan {plan_hash} -- not quoted. Refresh: tools/pmc_collect.sh + tools/pmc_summary.py, then commit the summary.")
        print("bench.py: " + msg, file=sys.stderr, flush=True)
        return None, msg, rec
    per_frame = rec["hbm_bytes_per_launch"] / rec["frames_per_launch"]
    return per_frame * frames_per_launch, (
        f"bytes per launch = PMC-measured {rec['hbm_bytes_per_launch'] / 1e9:.2f} GB per {rec['frames_per_launch']}-frame "
        f"launch ({os.path.relpath(path, ROOT)}: FETCH_SIZE x{rec.get('fetch_factor', 2)} + WRITE_SIZE, separate --pmc passes; "
        "these are the L2s' memory-side (fabric) request bytes -- L2 MISS traffic, Infinity-Cache hits included -- i.e. an "
        "upper bound of the HBM bytes; FETCH_SIZE factor: MI355X_MICROARCH.md, re-checked on this library's own access "
        f"patterns: {rec.get('fetch_calibration', 'see profiles/README.md')}) "
        f"scaled to {frames_per_launch} frames; algorithmic bytes {rec.get('algorithmic_mb_per_frame', 45.7)} MB/frame "
        "layer by layer"), rec


def _soft(name, fn, agree=None):
    """Run a SECONDARY measurement; a Python-level failure becomes `{"error": ...}` under its key instead of costing
    the headline line.  `agree` (world > 1): all ranks exchange an ok-flag afterwards, so that a rank whose leg failed
    does not leave the others' result standing alone (the leg is reported failed on every rank)."""
    try:
        out, ok = fn(), 1.0
    except Exception as e:   # noqa: BLE001 -- by design: nothing a secondary leg raises may kill the headline
        import traceback
        traceback.print_exc()
        out, ok = {"error": f"{name}: {type(e).__name__}: {e}"[:300]}, 0.0
    if agree is not None:
        try:
            if agree(ok) < 1.0 and ok:
                out = {"error": f"{name}: failed on another rank"}
        except Exception as e:   # noqa: BLE001
            out = {"error": f"{name}: status exchange failed: {type(e).__name__}: {e}"[:300]}
    return out


class _Watchdog:
    """The secondary legs run under a wall-clock budget: when it expires, rank 0 prints the line with what has been
    measured so far (the unfinished leg carries `{"error": "timeout"}`) and every rank leaves with os._exit -- a hung
    collective in a secondary leg cannot cost the headline number."""

    def __init__(self, seconds: float, emit):
        import threading
        self._t = threading.Timer(seconds, self._fire)
        self._t.daemon = True
        self._emit = emit
        self.leg = "?"
        self._t.start()

    def _fire(self):
        rc = 0
        try:
            self._emit(self.leg)
        except Exception:   # noqa: BLE001 -- the line could not be printed: that must not look like success
            import traceback
            traceback.print_exc()
            rc = 3
        finally:
            sys.stdout.flush()
            os._exit(rc)

    def cancel(self):
        self._t.cancel()


def busy_union_ms(intervals):
    """Total length of the union of [start, end] intervals (ms): the time during which at least one of them is open."""
    iv = sorted(intervals)
    if not iv:
        return 0.0
    busy, cur_s, cur_e = 0.0, iv[0][0], iv[0][1]
    for s_, e_ in iv[1:]:
        if s_ > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s_, e_
        else:
            cur_e = max(cur_e, e_)
    return busy + (cur_e - cur_s)


def _time_iterations(w, steps, warmup, barrier):
    for _ in range(warmup):
        w.iteration()
    w.time_trunk = True
    w.trunk_events = []
    w.update_events = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        w.iteration()
    barrier()
    w.time_trunk = False
    return time.perf_counter() - t0


_LINE_OUT = None     # the process's REAL stdout, kept aside by _claim_stdout(): only the JSON line is written to it


def _claim_stdout():
    """The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a version banner through C
    stdio, which is flushed at exit, i.e. AFTER the line): from here on file descriptor 1 is stderr for everybody, and the
    line goes out through a private handle on the original stdout."""
    global _LINE_OUT
    if _LINE_OUT is None:
        sys.stdout.flush()
        _LINE_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    return _LINE_OUT


def _print_line(obj):
    out = _claim_stdout()
    try:                      # whatever native libraries still hold in C stdio buffers (RCCL's banner) goes out BEFORE the line,
        import ctypes         # so that the line is also the last thing written when a launcher merges the two streams
        sys.stdout.flush(); sys.stderr.flush()
        ctypes.CDLL(None).fflush(None)
    except Exception:         # noqa: BLE001
        pass
    out.write(json.dumps(obj) + "\n")
    out.flush()


def run_rank(a, rank: int, local_rank: int, world: int, init_method=None):
    dist = torch.distributed
    _claim_stdout()
    if a.dry_run:   # launcher / rendezvous / bucket all-reduce only (CPU, gloo): what tests/test_bench_launch.py runs
        if world > 1:
            dist.init_process_group("gloo", init_method=init_method, rank=rank, world_size=world)
        bucket = torch.full((POLICY_FLAT_PARAMS,), float(rank + 1))
        if world > 1:
            from embodied_clip_amd.dist import allreduce_flat
            allreduce_flat(bucket)
        ok = bool((bucket == world * (world + 1) / 2).all())
        if rank == 0:
            _print_line({"dry_run": True, "n_gpus": world, "rccl_ranks": dist.get_world_size() if world > 1 else 1,
                         "bucket_elems": POLICY_FLAT_PARAMS, "allreduce_ok": ok})
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return 0 if ok else 1

    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback for the product path)"
    gpu = 0 if a.share_gpu else local_rank      # --share-gpu (tests only): every rank on cuda:0, exchange over gloo
    torch.cuda.set_device(gpu)
    dev = torch.device(f"cuda:{gpu}")
    use_dist = world > 1 or a.force_dist        # --force-dist: RCCL initialised and used also at world size 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if a.share_gpu:
            dist.init_process_group("gloo", init_method=init_method, rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", init_method=init_method, rank=rank, world_size=world,
                                    device_id=dev)   # RCCL over xGMI

    from embodied_clip_amd.engine import Worker
    total = a.actors_total if a.actors_total else a.actors
    if a.scaling == "strong":          # the same global actor list sharded over the ranks (N/G each; shards
        from embodied_clip_amd.dist import shard_actors   # may differ by one actor: engine.Worker scales by local / GLOBAL size)
        if total < world:
            raise SystemExit(f"--scaling strong needs at least one actor per rank ({total} actors, {world} ranks)")
        per_gpu = shard_actors(total, rank, world)[1]
        global_actors = total
    else:
        per_gpu = total
        global_actors = total * world
    wkw = dict(T=a.rollout, device=dev, seed=0, rank=rank, world=world, update_repeats=a.update_repeats,
               encoder_chunk=a.encoder_chunk, encoder=a.encoder, encoder_streams=a.encoder_streams,
               frames_u8=a.frames_u8, num_mini_batch=a.num_mini_batch, force_allreduce=a.force_dist,
               overlap_allreduce=not a.no_overlap_allreduce)
    if a.encoder == "zeroshot":
        wkw.update(encoder="rn50", zeroshot=True)
    w = Worker(per_gpu, frames_host=a.frames_host, sync_actions=a.sync_actions, **wkw)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def maxreduce(x: float) -> float:
        if not use_dist:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def agree(ok: float) -> float:      # MIN over ranks of a leg's ok-flag (see _soft)
        if not use_dist:
            return ok
        tt = torch.tensor([ok], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MIN)
        return float(tt.item())

    # ---- the headline measurement -----------------------------------------------------------------------------------
    dt = maxreduce(_time_iterations(w, a.steps, a.warmup, barrier))
    # dominant kernel family: the RN50 trunk's MFMA implicit-GEMM convs, one ec_rn50_forward per env step
    trunk_ms = [e0.elapsed_time(e1) for e0, e1 in w.trunk_events]
    avg_trunk_ms = sum(trunk_ms) / max(1, len(trunk_ms))
    # The encoder launches of one env step run concurrently, one per HIP stream.  The chip-level rate of the encoder
    # alone is therefore taken over the UNION of their [start, end] intervals (first start -> last end of the step's
    # launches): the line's `frac`.  `frac_iteration` is the whole iteration's algorithmic flop over the timed wall clock.
    n_conc = max(1, per_gpu // max(1, w.encode_frames))
    # UNION of ALL encoder launch intervals of the timed region (sweep over the HIP-event [start, end] pairs of every launch on
    # every slice stream): the time during which at least one encoder launch runs.  (Until round 4 the union was taken per env
    # step over that step's launches only; the slices' streams drift against each other, so that per-step figure moved by
    # +-10 % between runs at equal throughput and counted the overlap with the neighbouring steps' launches twice.)
    avg_union_ms = avg_trunk_ms
    if w.trunk_events:
        ref = w.trunk_events[0][0]
        busy = busy_union_ms([(ref.elapsed_time(e0), ref.elapsed_time(e1)) for e0, e1 in w.trunk_events])
        avg_union_ms = busy / max(1, len(w.trunk_events) // n_conc)        # per env step (n_conc launches each)
    upd_ms = [e0.elapsed_time(e1) for e0, e1 in getattr(w, "update_events", [])]
    update_ms = round(sum(upd_ms) / len(upd_ms), 2) if upd_ms else None
    info = w.loss_info()
    plan_hash = w.slices[0].enc.plan_hash() if hasattr(w.slices[0].enc, "plan_hash") else None
    enc_frames = w.encode_frames
    rccl_ranks = dist.get_world_size() if use_dist else 1

    frames = a.rollout * global_actors * a.steps
    value = frames / dt
    enc_mac = {"rn50": TRUNK_MAC_PER_FRAME, "vit": VIT_MAC_PER_FRAME, "rn50x16": RN50X16_MAC_PER_FRAME,
               "zeroshot": TRUNK_MAC_PER_FRAME + ATTNPOOL_MAC_PER_FRAME}[a.encoder]
    flop_per_frame = 2 * (enc_mac + (ZS_POLICY_ACT_MAC + ZS_POLICY_UPDATE_MAC if a.encoder == "zeroshot"
                                     else POLICY_ACT_MAC + POLICY_UPDATE_MAC +
                                     # RN50x16: 3072 input channels of the compressor's first conv
                                     (9 * 49 * 1024 * 128 if a.encoder == "rn50x16" else 0)))
    out = None
    if rank == 0:
        # flop of what one timed launch covers (zero-shot: the events bracket trunk + AttentionPool2d)
        flops_call = 2.0 * enc_mac * enc_frames            # one timed launch = one (slice of the) encoder forward
        achieved_launch = flops_call / (avg_trunk_ms * 1e-3) / 1e12
        achieved_union = flops_call * n_conc / (avg_union_ms * 1e-3) / 1e12
        # THE fraction of the line: algorithmic flop of the whole iteration (encoder + act + update) over the
        # driver-visible wall clock, per GPU -- recomputable from `value` / `ms_per_step` alone
        achieved_iter = value * flop_per_frame / world / 1e12
        if a.encoder == "rn50x16":
            traffic, tnote, trec = None, "no PMC summary for the RN50x16 trunk (functional, not tuned)", None
        else:
            traffic, tnote, trec = (None, "--no-traffic", None) if a.no_traffic else measured_traffic(
                "vit" if a.encoder == "vit" else "rn50", plan_hash, enc_frames)
        # the encoder's fraction from the COMMITTED profile alone (reproducible from profiles/): algorithmic flop of the
        # profiled single launch / its summed kernel time (rocprofv3 kernel trace)
        frac_profiles = frac_profiles_256 = None
        if trec and trec.get("kernel_time_us") and trec.get("plan_hash") in (None, "", plan_hash):
            frac_profiles = round(2.0 * enc_mac * trec["frames_per_launch"] / (trec["kernel_time_us"] * 1e-6) / 1e12
                                  / MFMA_BF16_PEAK_TFLOPS, 4)
            if trec.get("single_launch_256"):
                frac_profiles_256 = round(2.0 * enc_mac * 256 / (trec["single_launch_256"]["kernel_time_us"] * 1e-6) / 1e12
                                          / MFMA_BF16_PEAK_TFLOPS, 4)
        # ... and of the regime the headline runs in (two slice streams): the committed rocprofv3 kernel trace of the engine's
        # steady state reduced to the per-env-step UNION of the encoder kernels' intervals (tools/engine_step_union.py)
        frac_conc, conc_note = None, "no profiles/engine_step_union.json"
        upath = os.path.join(ROOT, "profiles", "engine_step_union.json")
        if a.encoder == "rn50" and os.path.exists(upath):
            urec = json.load(open(upath))
            if urec.get("plan_hash") in (None, "", plan_hash) and urec.get("frames_per_env_step") == per_gpu:
                frac_conc = round(2.0 * enc_mac * urec["frames_per_env_step"] / (urec["union_ms_per_env_step"] * 1e-3) / 1e12
                                  / MFMA_BF16_PEAK_TFLOPS, 4)
                conc_note = (f"profiles/engine_step_union.json: encoder kernels of {urec.get('env_steps')} steady-state env steps, union "
                             f"{urec['union_ms_per_env_step']} ms per env step (sum of kernel durations {urec.get('sum_kernel_ms_per_env_step')} ms, "
                             f"overlap factor {urec.get('overlap_factor')})")
            else:
                conc_note = (f"profiles/engine_step_union.json was collected on plan {urec.get('plan_hash')} at "
                             f"{urec.get('frames_per_env_step')} frames per env step; this run: plan {plan_hash}, {per_gpu} frames -- not quoted")
        # the second roof: L2-miss bytes of the concurrent launches of an env step over their union, against achievable HBM
        hbm_tbs = (traffic * n_conc / (avg_union_ms * 1e-3) / 1e12) if traffic else None
        workload = {"rn50": "RoboTHOR ObjectNav: frozen CLIP-RN50 encoder",
                    "vit": "RoboTHOR ObjectNav: frozen CLIP ViT-B/32 encoder (11 blocks)",
                    "rn50x16": "RoboTHOR ObjectNav: frozen CLIP-RN50x16 encoder (width 96, layers 6/8/18/8, 3072 x 7 x 7 features; "
                               "functional path, not tuned)",
                    "zeroshot": "Zero-shot ObjectNav: frozen CLIP-RN50 trunk + AttentionPool2d image embedding, goal = "
                                "CLIP text-tower embedding table"}[a.encoder]
        out = {
            "metric": "env-frames/sec (CLIP encode + policy fwd/bwd + PPO update)",
            "value": round(value, 1), "unit": "env-frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True, "scaling": a.scaling,
            "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic" if not a.share_gpu else "synthetic; --share-gpu TEST MODE: all ranks on ONE GPU over gloo, not a throughput",
            "config": {"workload": workload + " (bf16 MFMA, fp32 accumulate) + 1-layer GRU actor-critic PPO (fp32), "
                                               "synthetic 224x224 RGB + random goal ids",
                       "actors_per_gpu": per_gpu, "global_actors": global_actors, "rollout": a.rollout,
                       "update_repeats": a.update_repeats, "num_mini_batch": a.num_mini_batch, "encoder_streams": a.encoder_streams,
                       "frames": ("pinned host -> H2D per step, " if a.frames_host else "resident in HBM, ") +
                                 ("uint8 HWC (normalisation fused)" if a.frames_u8 else "fp32 normalised HWC"),
                       "env_order": ("action-synchronous: the sampled actions are copied to the host every env step before the "
                                     "next observation is served" if a.sync_actions else
                                     "free-running: the synthetic env does not read the actions (SURVEY.md 8d); see `sync_actions`"),
                       "parallelism": f"dp{world} (actors sharded; the flat 13.9 MB gradient bucket SUM-all-reduced per optimiser step" +
                                      (": one call after the backward)" if a.no_overlap_allreduce else
                                       ": GRU + heads section (12.8 MB) under the goal encoder's backward, the remaining 1.1 MB after it)"),
                       "flop_per_frame": flop_per_frame,
                       "policy_gemm_mode": (
                           "EC_POLICY_FAST=0: every policy GEMM fp32-exact (fp32 x fp32 as six bf16 products, bf16 x fp32 as three planes)"
                           if os.environ.get("EC_POLICY_FAST", "1") in ("", "0") and os.environ.get("EC_GEMM_BWD3", "0") in ("", "0") else
                           "forward fp32-exact except the compressor conv over the stored features (two leading planes of W1: 16 mantissa "
                           "bits); backward: the large gradient GEMMs on the three leading bf16x3 products, dW1 on two planes of dc1 "
                           "(products accurate to 2^-16 .. 2^-17; parity tests at unchanged tolerances; EC_POLICY_FAST=0 = fp32-exact)")},
            "update_ms": update_ms,      # the 4 PPO epochs (forward, loss, backward, all-reduce, clip + Adam) per iteration: HIP events, main stream
            "rccl_ranks": rccl_ranks, "allreduce_ms_per_rank": None,
            "roofline": {"bound": "mfma", "co_bound": "mfma + hbm (co-bound: see hbm_frac)",
                         # schema 2 (round 5 on): `frac` / `achieved` = the dominant kernel family over the busy union of its
                         # launches (== `frac_union`); the whole-iteration figure of rounds 1-4 is `frac_iteration`
                         "schema": 2,
                         "kernel": ("ec_rn50_forward (conv_bneck whole-bottleneck / conv_igemm / conv_pair / conv3x3_narrow MFMA kernels)"
                                    if a.encoder != "vit" else "ec_vit_forward (conv_igemm GEMMs with LayerNorm folded in + mha kernel)"),
                         "achieved": round(achieved_union, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved_union / MFMA_BF16_PEAK_TFLOPS, 4),
                         "frac_union": round(achieved_union / MFMA_BF16_PEAK_TFLOPS, 4),
                         "frac_profiles_concurrent": frac_conc, "frac_profiles_concurrent_note": conc_note,
                         "frac_iteration": round(achieved_iter / MFMA_BF16_PEAK_TFLOPS, 4), "achieved_iteration": round(achieved_iter, 1),
                         "frac_profiles": frac_profiles, "frac_profiles_single_256_launch": frac_profiles_256,
                         "hbm_frac": round(hbm_tbs / HBM_ACHIEVABLE_TBS, 4) if hbm_tbs else None,
                         "hbm_achieved_tbs": round(hbm_tbs, 3) if hbm_tbs else None, "hbm_peak_tbs": HBM_ACHIEVABLE_TBS,
                         "frac_note": "frac = achieved / peak for the dominant kernel family, ec_rn50_forward: algorithmic_flop_per_launch x "
                                      "concurrent_launches / avg_step_union_ms, the launches' durations measured LIVE with HIP events on the "
                                      "streams they run on: avg_step_union_ms = (UNION of ALL encoder launch intervals of the timed region, "
                                      "i.e. the time during which at least one encoder launch runs) / env steps -- the engine keeps "
                                      "concurrent_launches launches in flight on as many streams; frac_iteration = value x "
                                      "config.flop_per_frame / n_gpus / peak: the WHOLE iteration's algorithmic flop (encoder + act step + "
                                      "4 update epochs, the fp32 policy included) over the timed wall clock, recomputable from value alone; "
                                      "frac_profiles = the committed rocprofv3 kernel trace of ONE engine launch (the plan with "
                                      "this hash) running ALONE on the chip, frac_profiles_single_256_launch = one 256-frame "
                                      "launch (profiles/*_hbm_traffic.json kernel_time_us): both reproducible from profiles/ "
                                      "alone; hbm_frac = traffic x concurrent_launches / avg_step_union_ms / 6.29 TB/s (the "
                                      "path is co-bound: both roofs are shown)",
                         "traffic": traffic, "traffic_kind": "L2-miss (fabric-side) bytes, Infinity-Cache hits included",
                         "traffic_note": tnote, "plan_hash": plan_hash,
                         "avg_launch_ms": round(avg_trunk_ms, 3), "avg_step_union_ms": round(avg_union_ms, 3),
                         "launches_timed": len(trunk_ms), "frames_per_launch": enc_frames,
                         "concurrent_launches": n_conc, "achieved_per_launch": round(achieved_launch, 1),
                         "algorithmic_flop_per_launch": flops_call,
                         "encoder_share_of_step": round(sum(trunk_ms) / (dt * 1e3) / max(1, n_conc), 3)},
            "loss": {k: round(v, 6) for k, v in info.items()},
        }

    # ---- secondary measurements: each fails SOFT (its key carries {"error": ...}) and the whole block runs under a watchdog
    import threading
    out_lock = threading.Lock()          # `out` is written by the main thread (put) and read by the watchdog's timer thread (emit)
    emitted = [False]

    def emit(timeout_leg=None):
        with out_lock:
            if emitted[0]:               # the line goes out exactly once (a timer firing beside the regular emit)
                return
            emitted[0] = True
            if rank == 0 and out is not None:
                line = dict(out)
                if timeout_leg is not None:
                    line.setdefault(timeout_leg, {"error": f"timeout: secondary legs exceeded {a.secondary_budget_s} s"})
                _print_line(line)

    dog = _Watchdog(a.secondary_budget_s, emit)

    def put(key, val):
        if out is not None and val is not None:
            with out_lock:
                out[key] = val

    if a.phase_times:
        def leg_phases():
            torch.cuda.synchronize(); p0 = time.perf_counter()
            w.collect_rollout(); torch.cuda.synchronize(); p1 = time.perf_counter()
            w.compute_returns(); torch.cuda.synchronize(); p2 = time.perf_counter()
            w.update(); w.after_update(); torch.cuda.synchronize(); p3 = time.perf_counter()
            return {"rollout_ms": round((p1 - p0) * 1e3, 1), "gae_ms": round((p2 - p1) * 1e3, 2),
                    "update_ms": round((p3 - p2) * 1e3, 1)}
        dog.leg = "phases"
        put("phases", _soft("phases", leg_phases, agree))

    # the single exchange step, timed alone on every rank (HIP events on the current stream, 20 calls)
    if use_dist:
        def leg_allreduce():
            from embodied_clip_amd.dist import allreduce_flat
            for _ in range(3):
                allreduce_flat(w.grads, force=True)
            barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                allreduce_flat(w.grads, force=True)
            e1.record(); torch.cuda.synchronize()
            whole = e0.elapsed_time(e1) / 20
            # the two sections the overlapped worker sends: GRU + heads (under the goal encoder's backward) and the rest (exposed)
            secs = []
            for sec in [w.rec] + w._other_sections():
                allreduce_flat(w.grads[sec], force=True)
                e0.record()
                for _ in range(20):
                    allreduce_flat(w.grads[sec], force=True)
                e1.record(); torch.cuda.synchronize()
                secs.append(e0.elapsed_time(e1) / 20)
            mine = torch.tensor([whole, secs[0], sum(secs[1:])], dtype=torch.float64, device=dev)
            allv = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allv, mine)
            nopt = a.update_repeats * a.num_mini_batch
            exposed = max(float(v[0 if a.no_overlap_allreduce else 2].item()) for v in allv)
            put("allreduce_sections_ms_per_rank", {"recurrent_section": [round(float(v[1].item()), 4) for v in allv],
                                                   "goal_encoder_section": [round(float(v[2].item()), 4) for v in allv]})
            put("allreduce_exposed", {"ms_per_optimiser_step": round(exposed, 4), "optimiser_steps_per_iteration": nopt,
                                      "share_of_iteration": round(nopt * exposed / (dt / a.steps * 1e3), 5),
                                      "what": ("the whole 13.9-MB bucket after the backward (--no-overlap-allreduce)" if a.no_overlap_allreduce else
                                               "the goal encoder's section, reduced after the backwards; the GRU + heads section (92 % of "
                                               "the bytes) runs on the communication stream under the goal encoder's backward")})
            return [round(float(v[0].item()), 4) for v in allv]
        dog.leg = "allreduce_ms_per_rank"
        put("allreduce_ms_per_rank", _soft("allreduce_ms_per_rank", leg_allreduce, agree))

    def free_worker():
        nonlocal w
        w = None             # free the previous worker's ~11 GB before building the next
        gc.collect(); torch.cuda.empty_cache()

    if not a.no_sync_actions and not a.sync_actions and not a.frames_host:
        def leg_sync():
            free_worker()
            ws_ = Worker(per_gpu, frames_host=False, sync_actions=True, **wkw)
            dts = maxreduce(_time_iterations(ws_, a.sync_steps, 1, barrier))
            r = {"value": round(a.rollout * global_actors * a.sync_steps / dts, 1), "unit": "env-frames/s", "steps": a.sync_steps,
                 "order": "per env step: act(t) on every slice -> the sampled actions of all actors copied D2H and waited for "
                          "(what ONE VectorSampledTasks.step(actions) over all samplers forces) -> observe() -> encode(t+1); "
                          "the free-running headline lets the host issue arbitrarily far ahead"}
            del ws_
            gc.collect(); torch.cuda.empty_cache()
            ws_ = Worker(per_gpu, frames_host=False, sync_actions="slice", **wkw)
            dts = maxreduce(_time_iterations(ws_, a.sync_steps, 1, barrier))
            r["per_slice_envs"] = {
                "value": round(a.rollout * global_actors * a.sync_steps / dts, 1), "unit": "env-frames/s",
                "order": "one vectorised env per actor slice: the host waits for slice s's actions only, steps that slice's env "
                         "and issues its encode(t+1) + act(t+1) while the other slice's encoder is still running (same "
                         "per-actor arithmetic, same round trips per actor)"}
            del ws_
            gc.collect(); torch.cuda.empty_cache()
            # ... and with the frames crossing PCIe every step as well (raw uint8 in pinned host memory): the peer of the plugin
            # route with uint8 sensor frames -- same order, same bytes over the bus, the engine's own storage and kernels
            ws_ = Worker(per_gpu, frames_host=True, sync_actions=True, **{**wkw, "frames_u8": True})
            dts = maxreduce(_time_iterations(ws_, a.sync_steps, 1, barrier))
            r["host_frames_u8"] = {"value": round(a.rollout * global_actors * a.sync_steps / dts, 1), "unit": "env-frames/s",
                                   "order": "action-synchronous (one env for all actors), uint8 frames in pinned host memory copied every env step"}
            del ws_
            return r
        dog.leg = "sync_actions"
        put("sync_actions", _soft("sync_actions", leg_sync, agree))
        gc.collect(); torch.cuda.empty_cache()
    if not a.no_h2d and not a.frames_host:
        def leg_h2d():
            free_worker()
            u8 = a.encoder != "vit"          # (the ViT patch-embed kernel takes the sensor's fp32 frames only)
            wh = Worker(per_gpu, frames_host=True, **{**wkw, "frames_u8": u8})
            dth = maxreduce(_time_iterations(wh, a.h2d_steps, 1, barrier))
            r = {"value": round(a.rollout * global_actors * a.h2d_steps / dth, 1), "unit": "env-frames/s", "steps": a.h2d_steps,
                 "frames": ("uint8 HWC" if u8 else "fp32 normalised HWC") + " in PINNED HOST memory, copied per slice on its "
                           "own copy stream (double-buffered) while the other slice computes" +
                           ("; /255 + CLIP mean/std fused into the stem kernel" if u8 else ""),
                 "h2d_bytes_per_env_step": per_gpu * 224 * 224 * 3 * (1 if u8 else 4)}
            del wh
            return r
        dog.leg = "h2d_inclusive"
        put("h2d_inclusive", _soft("h2d_inclusive", leg_h2d, agree))
        gc.collect(); torch.cuda.empty_cache()
    if world > 1 and a.scaling == "strong" and not a.no_weak:
        def leg_weak():
            free_worker()
            ww = Worker(total, frames_host=a.frames_host, **wkw)
            dtw = maxreduce(_time_iterations(ww, a.steps, a.warmup, barrier))
            r = {"value": round(a.rollout * total * world * a.steps / dtw, 1), "unit": "env-frames/s",
                 "actors_per_gpu": total, "global_actors": total * world, "ms_per_step": round(dtw / a.steps * 1e3, 2)}
            del ww
            return r
        dog.leg = "weak"
        put("weak", _soft("weak", leg_weak, agree))
        gc.collect(); torch.cuda.empty_cache()
    if world == 1 and a.encoder == "rn50" and not a.no_plugin:
        def leg_plugin():
            free_worker()
            from embodied_clip_amd.plugin_path import time_plugin_path
            r = time_plugin_path(per_gpu, a.rollout, dev, steps=a.plugin_steps, warmup=1, update_repeats=a.update_repeats)
            gc.collect(); torch.cuda.empty_cache()
            # the same route when the RGB sensor hands over raw uint8 frames (a quarter of the PCIe bytes)
            r8 = time_plugin_path(per_gpu, a.rollout, dev, steps=a.plugin_steps, warmup=1, update_repeats=a.update_repeats,
                                  frames_u8=True)
            res = {**r, "fraction_of_engine": round(r["value"] / value, 3),
                   "u8_sensor_frames": {"value": r8["value"], "ms_per_step": r8["ms_per_step"],
                                        "fraction_of_engine": round(r8["value"] / value, 3), "route": r8["route"]}}
            peer = ((out or {}).get("sync_actions") or {}).get("host_frames_u8") if out is not None else None
            if isinstance(peer, dict) and peer.get("value"):   # the engine in the same order with the same frames over PCIe
                res["u8_sensor_frames"]["fraction_of_engine_same_order_host_frames"] = round(r8["value"] / peer["value"], 3)
            return res
        dog.leg = "plugin_path"
        put("plugin_path", _soft("plugin_path", leg_plugin))
        gc.collect(); torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        def leg_cpu():
            r = cpu_baseline(a.cpu_actors, a.cpu_rollout, a.update_repeats)
            r["gpu_over_cpu"] = round(value / r["value"], 1)
            return r
        dog.leg = "cpu_baseline"
        put("cpu_baseline", _soft("cpu_baseline", leg_cpu))
    dog.cancel()
    emit()
    if use_dist:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:   # noqa: BLE001 -- the line is out; a teardown hiccup must not turn into a non-zero exit
            pass
    return 0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--actors", type=int, default=256, help="synthetic actors in total (strong) / per GPU (weak)")
    ap.add_argument("--actors-total", type=int, default=0,
                    help="global actor count, overrides --actors (512 = BASELINE config 4: 64 per GPU at 8 GPUs)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="strong (default; the metric's '256 actors, 1/2/4/8 GPUs'): the actor total is sharded over the "
                         "ranks; weak: that many actors PER GPU")
    ap.add_argument("--rollout", type=int, default=128)
    ap.add_argument("--update-repeats", type=int, default=4)
    ap.add_argument("--num-mini-batch", type=int, default=1, help="PPO minibatches per epoch (contiguous actor ranges)")
    ap.add_argument("--encoder-chunk", type=int, default=0)
    ap.add_argument("--encoder-streams", type=int, default=2, help="concurrent HIP streams for the encoder")
    ap.add_argument("--frames-u8", action="store_true",
                    help="raw uint8 frames (normalisation fused into the stem); default is the reference sensor's wire "
                         "form, fp32 normalised HWC")
    ap.add_argument("--frames-host", action="store_true",
                    help="frames live in pinned HOST memory and cross PCIe every env step (the plugin contract); the "
                         "default keeps them resident in HBM (SURVEY.md 8d) and reports this as `h2d_inclusive`")
    ap.add_argument("--encoder", "--config", dest="encoder", choices=("rn50", "vit", "zeroshot", "rn50x16"), default="rn50",
                    help="rn50 = BASELINE headline config; vit = config 3 (ViT-B/32); zeroshot = config 5 "
                         "(RN50 + attnpool image embedding, CLIP-text goal table)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-h2d", action="store_true", help="skip the secondary h2d_inclusive measurement")
    ap.add_argument("--no-weak", action="store_true", help="skip the secondary weak-scaling measurement (N > 1)")
    ap.add_argument("--no-plugin", action="store_true",
                    help="skip the secondary `plugin_path` measurement: the same iteration driven through the drop-in plugin "
                         "classes with the reference's tensor contracts (embodied_clip_amd/plugin_path.py)")
    ap.add_argument("--plugin-steps", type=int, default=3, help="timed iterations of each plugin_path leg (>= 3: a 10 %% move is then not noise)")
    ap.add_argument("--h2d-steps", type=int, default=5, help="timed iterations of the h2d_inclusive measurement")
    ap.add_argument("--sync-actions", nargs="?", const=True, default=False, choices=[True, "slice"],
                    help="HEADLINE run in the action-synchronous order: every env step the sampled actions are copied D2H and "
                         "waited for before the next observation is served (what VectorSampledTasks.step(actions) forces); "
                         "by default this order is a secondary key of the line (`sync_actions`)")
    ap.add_argument("--no-sync-actions", action="store_true", help="skip the secondary `sync_actions` measurement")
    ap.add_argument("--sync-steps", type=int, default=3, help="timed iterations of the sync_actions measurement")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise RCCL (backend nccl) and run the flat-bucket all-reduce also at world size 1 -- the "
                         "first-contact check of the N > 1 path on a 1-GPU box (tests/test_gpu_multi.py)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="TESTS ONLY: every rank runs on cuda:0 and the ranks exchange over gloo (RCCL refuses two ranks on one "
                         "device) -- the N > 1 code path of this file on a 1-GPU box; the line says so in `data` and its value is "
                         "not a throughput of anything (tests/test_gpu_multi.py)")
    ap.add_argument("--no-overlap-allreduce", action="store_true",
                    help="one 13.9-MB all-reduce AFTER the backward instead of the GRU + heads section reduced under the goal "
                         "encoder's backward (engine.Worker(overlap_allreduce=False)); only matters with a collective")
    ap.add_argument("--secondary-budget-s", type=float, default=420.0,
                    help="wall-clock budget of ALL secondary legs together; on expiry the line is printed with what is there")
    ap.add_argument("--no-traffic", action="store_true", help="do not read profiles/*_hbm_traffic.json")
    ap.add_argument("--phase-times", action="store_true", help="extra untimed iteration with per-phase sync timing")
    ap.add_argument("--cpu-actors", type=int, default=32)
    ap.add_argument("--cpu-rollout", type=int, default=8)
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher check only: rendezvous + one flat-bucket all-reduce over gloo on CPU, no GPU work")
    a = ap.parse_args(argv)
    return a


def main(argv=None):
    a = parse_args(argv)
    if os.environ.get("EC_STREAMS_UNVERIFIED") == "1":   # (the profiler-only waiver of the stream-concurrency check: tools/pmc_collect.sh)
        raise SystemExit("bench.py does not run with EC_STREAMS_UNVERIFIED=1: its numbers rest on two launches really in flight")
    env_world = int(os.environ.get("WORLD_SIZE", "0") or 0)
    if env_world >= 1 and "RANK" in os.environ:       # launched one process per GPU by torch.distributed.run
        world, rank = env_world, int(os.environ["RANK"])
        if a.gpus != world:
            raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
        return run_rank(a, rank, int(os.environ.get("LOCAL_RANK", str(rank))), world, None)
    if a.gpus > 1 or a.force_dist:                      # self-launch: one spawned rank per GPU, file-store rendezvous
        import torch.multiprocessing as mp
        fd, store = tempfile.mkstemp(prefix="ec_bench_store_")
        os.close(fd); os.unlink(store)
        mp.spawn(_spawned_entry, args=(a, store), nprocs=a.gpus, join=True)
        return 0
    return run_rank(a, 0, 0, 1, None)


def _spawned_entry(local_rank: int, a, store_path: str):
    os.environ.update(RANK=str(local_rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(a.gpus),
                      MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    rc = run_rank(a, local_rank, local_rank, a.gpus, f"file://{store_path}")
    if rc:
        raise SystemExit(rc)


if __name__ == "__main__":
    sys.exit(main() or 0)
