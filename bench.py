#!/usr/bin/env python
"""Headline benchmark: env-frames/sec of the DD-PPO hot path on MI355X.

One "step" = one full training iteration of one worker per GPU over synthetic
frames already resident in HBM:
  T=128 x [CLIP-RN50 encode N frames + policy act + sample] + GAE
  + 4 x [policy forward over [T,N] + PPO loss + backward + flat-bucket
         all-reduce + clip + Adam]
value = (T * N_total * steps) / wall time, max over ranks, barrier+sync bracketed.

    python bench.py                       # 1 GPU, 256 actors, K=2 W=1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 2 --warmup 1

Scaling is WEAK: every GPU owns --actors (default 256) synthetic actors
(DD-PPO: actors shard over GPUs; the only exchange is one SUM all-reduce of the
13.9 MB flat policy-gradient bucket per optimiser step, RCCL over xGMI).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

TRUNK_MAC_PER_FRAME = 5_367_226_368          # SURVEY.md §8d (RN50 trunk, 224x224)
VIT_MAC_PER_FRAME = 4_050_683_904            # SURVEY.md §8d (ViT-B/32, 11 of 12 blocks)
POLICY_ACT_MAC = 16_846_336
POLICY_UPDATE_MAC = 150_775_808              # 4 x (fwd + bwd)
MFMA_BF16_PEAK_TFLOPS = 2500.0               # MI355X_MICROARCH.md: dense bf16 MFMA
# HBM bytes per ec_rn50_forward launch at N=256 from rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction +
# WRITE_SIZE, separate --pmc passes): profiles/r01_trunk_b256_hbm_traffic.txt.  Algorithmic: 45.7 MB/frame.
TRUNK_HBM_BYTES_PER_LAUNCH_N256 = 1.22e10


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--actors", type=int, default=256, help="synthetic actors per GPU (weak) / in total (strong)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default, DD-PPO convention): --actors per GPU; strong: --actors in total, sharded")
    ap.add_argument("--rollout", type=int, default=128)
    ap.add_argument("--update-repeats", type=int, default=4)
    ap.add_argument("--encoder-chunk", type=int, default=0)
    ap.add_argument("--encoder-streams", type=int, default=2, help="concurrent HIP streams for the RN50 encoder")
    ap.add_argument("--frames-u8", action="store_true",
                    help="raw uint8 frames in HBM (normalisation fused into the stem); default is the reference "
                         "sensor's wire form, fp32 normalised HWC")
    ap.add_argument("--encoder", choices=("rn50", "vit"), default="rn50",
                    help="rn50 = BASELINE headline config; vit = config 3 (ViT-B/32, parity-unpinned fusion)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--phase-times", action="store_true", help="extra untimed iteration with per-phase sync timing")
    ap.add_argument("--cpu-actors", type=int, default=32)
    ap.add_argument("--cpu-rollout", type=int, default=8)
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world == 1 and a.gpus > 1:
        raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nnodes=1 "
                         f"--nproc-per-node {a.gpus} --master-addr 127.0.0.1 --master-port 29500 bench.py --gpus {a.gpus}")
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback for the product path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world)   # RCCL over xGMI

    from embodied_clip_amd.engine import Worker
    if a.scaling == "strong":          # the same global actor list sharded over the ranks (N/G each)
        if a.actors % world != 0:
            raise SystemExit(f"--scaling strong needs --actors ({a.actors}) divisible by the world size ({world})")
        a.actors //= world
    w = Worker(a.actors, T=a.rollout, device=dev, seed=0, rank=rank, world=world, update_repeats=a.update_repeats,
               encoder_chunk=a.encoder_chunk, encoder=a.encoder, encoder_streams=a.encoder_streams,
               frames_u8=a.frames_u8)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        w.iteration()
    w.time_trunk = True
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        w.iteration()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    # dominant kernel family: the RN50 trunk's MFMA implicit-GEMM convs, one ec_rn50_forward per env step
    trunk_ms = [e0.elapsed_time(e1) for e0, e1 in w.trunk_events]
    avg_trunk_ms = sum(trunk_ms) / max(1, len(trunk_ms))
    # The encoder launches of one env step run concurrently, one per HIP stream.  The chip-level rate is therefore
    # taken over the UNION of their [start, end] intervals (first start -> last end of the step's launches), which
    # stays correct whether the launches overlap fully, partly, or (under a serialising profiler) not at all.
    n_conc_ev = max(1, a.actors // max(1, w.encode_frames))
    union_ms = []
    if w.trunk_events:
        ref = w.trunk_events[0][0]
        for i in range(0, len(w.trunk_events) - n_conc_ev + 1, n_conc_ev):
            grp = w.trunk_events[i:i + n_conc_ev]
            union_ms.append(max(ref.elapsed_time(e1) for _, e1 in grp) - min(ref.elapsed_time(e0) for e0, _ in grp))
    avg_union_ms = sum(union_ms) / max(1, len(union_ms)) if union_ms else avg_trunk_ms
    info = w.loss_info()

    phases = None
    if a.phase_times:
        torch.cuda.synchronize(); p0 = time.perf_counter()
        w.collect_rollout(); torch.cuda.synchronize(); p1 = time.perf_counter()
        w.compute_returns(); torch.cuda.synchronize(); p2 = time.perf_counter()
        w.update(); w.after_update(); torch.cuda.synchronize(); p3 = time.perf_counter()
        phases = {"rollout_ms": round((p1 - p0) * 1e3, 1), "gae_ms": round((p2 - p1) * 1e3, 2),
                  "update_ms": round((p3 - p2) * 1e3, 1)}
    if rank == 0:
        frames = a.rollout * a.actors * world * a.steps
        value = frames / dt
        enc_mac = TRUNK_MAC_PER_FRAME if a.encoder == "rn50" else VIT_MAC_PER_FRAME
        flops_call = 2.0 * enc_mac * w.encode_frames     # one timed launch = one (slice of the) encoder forward
        # encoder launches run `n_conc` at a time (one per HIP stream): the chip-level rate is the aggregate
        n_conc = max(1, a.actors // w.encode_frames)
        achieved_launch = flops_call / (avg_trunk_ms * 1e-3) / 1e12
        achieved = flops_call * n_conc / (avg_union_ms * 1e-3) / 1e12
        out = {
            "metric": "env-frames/sec (CLIP encode + policy fwd/bwd + PPO update)",
            "value": round(value, 1), "unit": "env-frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True, "scaling": a.scaling,
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("RoboTHOR ObjectNav: frozen CLIP-RN50 encoder" if a.encoder == "rn50" else
                                    "RoboTHOR ObjectNav: frozen CLIP ViT-B/32 encoder (11 blocks)") +
                                   " (bf16 MFMA, fp32 accumulate) + "
                                   "1-layer GRU actor-critic PPO (fp32), synthetic 224x224 RGB + random goal ids",
                       "actors_per_gpu": a.actors, "global_actors": a.actors * world, "rollout": a.rollout,
                       "update_repeats": a.update_repeats, "num_mini_batch": 1, "encoder_streams": a.encoder_streams,
                       "frames": "uint8 HWC (normalisation fused)" if a.frames_u8 else "fp32 normalised HWC",
                       "parallelism": f"dp{world} (actors sharded; one flat 13.9 MB grad all-reduce per optimiser step)",
                       "flop_per_frame": 2 * ((TRUNK_MAC_PER_FRAME + POLICY_ACT_MAC + POLICY_UPDATE_MAC)
                                             if a.encoder == "rn50" else VIT_MAC_PER_FRAME)},
            "roofline": {"bound": "mfma", "kernel": ("ec_rn50_forward (conv_igemm / conv_pair / conv3x3_narrow MFMA kernels, 50 launches per call)" if a.encoder == "rn50"
                                    else "ec_vit_forward (conv_igemm GEMMs + mha/layernorm kernels)"),
                         "achieved": round(achieved, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4),
                         "traffic": (TRUNK_HBM_BYTES_PER_LAUNCH_N256 * w.encode_frames / 256.0
                                     if (a.actors == 256 and a.encoder == "rn50") else None),
                         "traffic_note": "HBM bytes per launch, PMC-measured offline (profiles/r01_trunk_b256_hbm_traffic.txt); "
                                         "algorithmic bytes 45.7 MB/frame x N (layer by layer; the fused layer-1 boundaries need less)",
                         "avg_launch_ms": round(avg_trunk_ms, 3), "avg_step_union_ms": round(avg_union_ms, 3),
                         "launches_timed": len(trunk_ms),
                         "frames_per_launch": w.encode_frames, "concurrent_launches": n_conc,
                         "achieved_per_launch": round(achieved_launch, 1),
                         "algorithmic_flop_per_launch": flops_call,
                         "encoder_share_of_step": round(sum(trunk_ms) / (dt * 1e3) * w.encode_frames / a.actors, 3)},
            "loss": {k: round(v, 6) for k, v in info.items()},
            **({"phases": phases} if phases else {}),
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.cpu_actors, a.cpu_rollout, a.update_repeats)
            out["cpu_baseline"]["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
