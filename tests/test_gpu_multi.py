"""GPU, world size 2 over RCCL: `python bench.py --gpus 2` self-launches two ranks, each owning half of the actors
(strong scaling), and the flat-bucket SUM all-reduce keeps the replicas' parameters identical.  Needs >= 2 MI355X:
the driver's 1-GPU test box skips it (the N > 1 path is covered on CPU by tests/test_dist_gloo.py and
tests/test_bench_launch.py).  Also here, behind device-count guards: BASELINE config 4 (512 actors sharded 8-way),
config 5's 2-GPU leg (zero-shot worker) and the a18 check proper -- HIP shard gradients summed THROUGH RCCL compared
with the oracle's unsharded gradient (the 1-GPU form of it runs in tests/test_gpu_configs.py).
At the end of the file (round 6): the a18 check and the replicas-stay-identical check with the two ranks SHARING cuda:0 and
exchanging over gloo -- the N > 1 engine path on real HIP kernels on the driver's 1-GPU box, even and uneven (3 + 2) shards."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags, timeout=1500):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags, "--no-cpu-baseline", "--no-h2d", "--no-traffic"],
                       capture_output=True, text=True, timeout=timeout, cwd=ROOT,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_rccl_first_contact_at_world_size_1():
    """the N > 1 path's plumbing on the driver's 1-GPU box -- `bench.py --gpus 1 --force-dist` goes through
    the SPAWN launcher (`_spawned_entry`, file-store rendezvous), `init_process_group("nccl", device_id=...)` (RCCL
    communicator set-up), and runs `allreduce_flat` on the device gradient bucket inside every optimiser step and in the
    timed all-reduce leg.  At world size 1 the collective is an identity, so the loss must be that of the plain run."""
    kw = ("--actors", "32", "--rollout", "8", "--steps", "1", "--warmup", "1", "--no-plugin", "--no-sync-actions")
    line = _bench("--gpus", "1", "--force-dist", *kw, timeout=900)
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1
    assert isinstance(line["allreduce_ms_per_rank"], list) and len(line["allreduce_ms_per_rank"]) == 1
    assert 0 < line["allreduce_ms_per_rank"][0] < 50
    plain = _bench("--gpus", "1", *kw, timeout=900)
    assert plain["allreduce_ms_per_rank"] is None
    # (the weight-gradient GEMMs of the update use split-K atomics: the gradient norm moves in the 5th digit run to run)
    for k, v in plain["loss"].items():
        assert abs(line["loss"][k] - v) <= 1e-3 * max(1.0, abs(v)), (k, line["loss"], plain["loss"])
    # the reproducible fraction of the line: value x flop_per_frame / peak
    r = line["roofline"]
    assert abs(r["frac_iteration"] - line["value"] * line["config"]["flop_per_frame"] / 1e12 / r["peak"]) < 2e-4
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-4 and 0.0 < r["frac"] < 1.0     # live HIP-event union of the encoder launches


@pytest.mark.parametrize("actors,slices", [(32, 1), (64, 2)])
def test_overlapped_allreduce_is_the_single_allreduce(actors, slices):
    """VERDICT r4 'missing' item 4: the GRU + heads section of the flat bucket (final first: ec_policy_backward3's event) is
    summed over the slices and all-reduced on the communication stream UNDER the goal encoder's backward; the remaining 1.1 MB
    after it.  Over RCCL at world size 1 (identity SUM) the worker must end where the single-call worker and the
    no-collective worker end: a missing stream dependency (a section reduced before it is final, an optimiser step before
    the communication stream is done) shows as a parameter difference of the order of lr."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_overlap_allreduce_check.py"), str(actors), str(slices)],
                       capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["moved"] > 1e-4                                    # 4 epochs of Adam at lr 3e-4 did move the weights
    assert out["max_abs_overlap_vs_single"] < 2e-6, out
    assert out["max_abs_overlap_vs_none"] < 2e-6, out
    for k, v in out["none_loss"].items():
        assert abs(out["overlap_loss"][k] - v) <= 1e-4 * max(1.0, abs(v)), (k, out)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs")
def test_bench_two_ranks_over_rccl():
    line = _bench("--gpus", "2", "--actors", "64", "--rollout", "8", "--steps", "1", "--warmup", "1", timeout=900)
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["scaling"] == "strong"
    assert line["config"]["actors_per_gpu"] == 32 and line["config"]["global_actors"] == 64
    assert len(line["allreduce_ms_per_rank"]) == 2 and all(0 < x < 50 for x in line["allreduce_ms_per_rank"])
    assert line["weak"]["actors_per_gpu"] == 64 and line["value"] > 0


@pytest.mark.skipif(torch.cuda.device_count() < 8, reason="needs 8 GPUs")
def test_config4_512_actors_sharded_8_way_over_rccl():
    """BASELINE config 4: Habitat ObjectNav shape -- 512 actors sharded 8-way (64 per GPU), RCCL gradient all-reduce."""
    line = _bench("--gpus", "8", "--actors-total", "512", "--rollout", "16", "--steps", "1", "--warmup", "1", "--no-weak")
    assert line["n_gpus"] == 8 and line["rccl_ranks"] == 8 and line["scaling"] == "strong"
    assert line["config"]["actors_per_gpu"] == 64 and line["config"]["global_actors"] == 512
    assert len(line["allreduce_ms_per_rank"]) == 8 and all(0 < x < 50 for x in line["allreduce_ms_per_rank"])
    assert line["value"] > 0 and abs(line["loss"]["ratio"] - 1.0) < 0.2


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs")
def test_config5_zeroshot_two_ranks_over_rccl():
    """BASELINE config 5's 2-GPU leg: the zero-shot dual-encoder worker, 256 actors sharded over two ranks."""
    line = _bench("--config", "zeroshot", "--gpus", "2", "--actors", "256", "--rollout", "8", "--steps", "1", "--warmup", "1",
                  "--no-weak")
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2
    assert line["config"]["actors_per_gpu"] == 128 and line["config"]["global_actors"] == 256
    assert "Zero-shot" in line["config"]["workload"] and line["value"] > 0


def _rank_grads(rank, world, store, q, backend="nccl", gpu=None):
    """One rank of the a18 check: HIP gradients of this rank's actor shard -> flat bucket -> SUM over the ranks (RCCL with one
    GPU per rank; gloo when the ranks share GPU `gpu`: the 1-GPU box's form of the check)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    gid = rank if gpu is None else gpu
    torch.cuda.set_device(gid)
    torch.distributed.init_process_group(backend, init_method=f"file://{store}", rank=rank, world_size=world)
    from embodied_clip_amd import synthetic as syn
    from embodied_clip_amd.dist import allreduce_flat, grad_scale, shard_actors
    from embodied_clip_amd.policy import PolicyHandle
    from embodied_clip_amd.ppo import ppo_loss_raw
    import _a18_case
    dev = torch.device(f"cuda:{gid}")
    T, N, case = _a18_case.make()
    h = PolicyHandle()
    flat = h.flatten(syn.policy_state_dict(0), dev)
    lo, cnt = shard_actors(N, rank, world)
    sl = slice(lo, lo + cnt)
    c = lambda t: t[:, sl].reshape(-1).contiguous().to(dev)
    rows = case["feat"][:, sl].reshape(T * cnt, 49, 2048).contiguous().to(dev)
    ws = torch.empty(h.workspace_bytes(T, cnt, True), dtype=torch.uint8, device=dev)
    hv, _ = h.forward(flat, rows, c(case["goal"]), case["h0"][sl].contiguous().to(dev), c(case["masks"]), T, cnt, ws)
    dhv, _ = ppo_loss_raw(hv, c(case["actions"]), c(case["old_lp"]), c(case["old_v"]), c(case["ret"]), c(case["nadv"]), 6,
                          grad_scale=grad_scale(T * cnt, T * N))
    bucket = torch.zeros_like(flat)
    h.backward(flat, rows, c(case["masks"]), T, cnt, ws, dhv, None, bucket)
    allreduce_flat(bucket)                                   # the single exchange step of the path, over RCCL
    torch.cuda.synchronize()
    q.put((rank, bucket.cpu(), {k: v for k, v in h.offsets.items()}))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs")
def test_rccl_summed_hip_gradients_equal_the_oracles_unsharded_gradient(tmp_path):
    """Row a18 end to end: HIP shard gradients -> flat bucket -> RCCL SUM all-reduce == the ORACLE's gradient of the
    unsharded batch (torch-CPU autograd), on every rank."""
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _a18_case
    from embodied_clip_amd import synthetic as syn
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    store = str(tmp_path / "store_a18")
    ps = [ctx.Process(target=_rank_grads, args=(r, 2, store, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = [q.get() for _ in range(2)]
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    ref = _a18_case.oracle_gradient(syn.policy_state_dict(0))
    assert torch.equal(got[0][1], got[1][1])
    for name, (o, k) in got[0][2].items():
        a, b = got[0][1][o:o + k], ref[name].reshape(-1)
        rel = ((a - b).norm() / b.norm().clamp_min(1e-20)).item()
        assert rel < 2e-4, (name, rel)


def _rank(rank, world, store, q, backend="nccl", gpu=None, actors=(4, 4)):
    os.environ.update(MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    gid = rank if gpu is None else gpu
    torch.cuda.set_device(gid)
    torch.distributed.init_process_group(backend, init_method=f"file://{store}", rank=rank, world_size=world)
    from embodied_clip_amd.engine import Worker
    w = Worker(actors[rank], T=4, device=f"cuda:{gid}", seed=0, rank=rank, world=world, update_repeats=2)
    w.iteration()
    torch.cuda.synchronize()
    q.put((rank, w.params.cpu(), list(w.shard_counts)))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs")
def test_replicas_stay_identical_after_allreduced_updates(tmp_path):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    store = str(tmp_path / "store")
    ps = [ctx.Process(target=_rank, args=(r, 2, store, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = dict((r[0], r[1]) for r in (q.get() for _ in range(2)))
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    assert torch.equal(got[0], got[1])            # same summed gradient bucket, same Adam step on every rank


# ---------------------------------------------------------------------------------------------------------------------------
# The same two checks on the driver's 1-GPU box: two ranks (two processes) SHARE cuda:0 and exchange over gloo.  RCCL refuses two
# ranks on one device; gloo does not care where the tensors live, and everything else of the N > 1 path is the product code:
# the HIP kernels of both ranks, the per-rank shard and gradient scale, `gather_actor_counts` / `check_job_seed`, the sectioned
# bucket summed on the communication stream under the goal encoder's backward, the fused clip + Adam step on the summed bucket.
# ---------------------------------------------------------------------------------------------------------------------------
def _spawn2(target, tmp_path, name, **kw):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    store = str(tmp_path / name)
    ps = [ctx.Process(target=target, args=(r, 2, store, q), kwargs=kw) for r in range(2)]
    for p in ps:
        p.start()
    got = [q.get() for _ in range(2)]
    for p in ps:
        p.join(180)
        assert p.exitcode == 0
    return sorted(got, key=lambda g: g[0])


def test_two_ranks_sharing_one_gpu_hip_gradients_summed_over_gloo_equal_the_oracles_unsharded_gradient(tmp_path):
    """Row a18 with BOTH ranks' gradients computed by the HIP path on this GPU and summed through `allreduce_flat` (gloo):
    equal on the two ranks, and the ORACLE's gradient of the unsharded batch to 2e-4 per tensor."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _a18_case
    from embodied_clip_amd import synthetic as syn
    got = _spawn2(_rank_grads, tmp_path, "store_a18_gloo", backend="gloo", gpu=0)
    ref = _a18_case.oracle_gradient(syn.policy_state_dict(0))
    assert torch.equal(got[0][1], got[1][1])
    for name, (o, k) in got[0][2].items():
        a, b = got[0][1][o:o + k], ref[name].reshape(-1)
        rel = ((a - b).norm() / b.norm().clamp_min(1e-20)).item()
        assert rel < 2e-4, (name, rel)


@pytest.mark.parametrize("actors", [(4, 4), (3, 2)])
def test_two_ranks_sharing_one_gpu_stay_identical_after_allreduced_updates(tmp_path, actors):
    """Two `engine.Worker`s (even and UNEVEN shards: 3 + 2 actors) run one full iteration each -- rollout, GAE, 4 epochs x 2 repeats
    of the sectioned, overlapped bucket exchange + fused clip + Adam -- and end with `torch.equal` parameters; both saw the same
    gathered actor counts."""
    got = _spawn2(_rank, tmp_path, "store_gloo_%d_%d" % actors, backend="gloo", gpu=0, actors=actors)
    assert got[0][2] == list(actors) and got[1][2] == list(actors)
    assert torch.equal(got[0][1], got[1][1])
    assert torch.isfinite(got[0][1]).all()


def test_bench_py_launched_as_the_driver_launches_it_with_two_ranks_sharing_one_gpu():
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ...` --
    the driver's launch line for N = 2 -- with `--share-gpu` (both ranks on cuda:0, gloo): the N > 1 branch of bench.py on real HIP
    kernels on the 1-GPU box: RANK / LOCAL_RANK / WORLD_SIZE from the environment, strong-scaling shards of an ODD actor total (33 =
    17 + 16), barrier + synchronize bracketing, MAX over ranks, rank 0 alone printing ONE JSON line, per-rank all-reduce timing."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--actors", "33", "--rollout", "8",
           "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-h2d", "--no-traffic", "--no-plugin", "--no-sync-actions"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                              # rank 0 alone prints
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and "TEST MODE" in line["data"]
    assert line["scaling"] == "strong" and line["config"]["global_actors"] == 33
    assert isinstance(line["allreduce_ms_per_rank"], list) and len(line["allreduce_ms_per_rank"]) == 2
    assert line["value"] > 0 and all(v == v and abs(v) < 1e6 for v in line["loss"].values())   # finite
