"""GPU, world size 2 over RCCL: `python bench.py --gpus 2` self-launches two ranks, each owning half of the actors
(strong scaling), and the flat-bucket SUM all-reduce keeps the replicas' parameters identical.  Needs >= 2 MI355X:
the driver's 1-GPU test box skips it (the N > 1 path is covered on CPU by tests/test_dist_gloo.py and
tests/test_bench_launch.py)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs")
def test_bench_two_ranks_over_rccl():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--actors", "64", "--rollout", "8",
                        "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-h2d", "--no-traffic"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["scaling"] == "strong"
    assert line["config"]["actors_per_gpu"] == 32 and line["config"]["global_actors"] == 64
    assert len(line["allreduce_ms_per_rank"]) == 2 and all(0 < x < 50 for x in line["allreduce_ms_per_rank"])
    assert line["weak"]["actors_per_gpu"] == 64 and line["value"] > 0


def _rank(rank, world, store, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    torch.distributed.init_process_group("nccl", init_method=f"file://{store}", rank=rank, world_size=world)
    from embodied_clip_amd.engine import Worker
    w = Worker(4, T=4, device=f"cuda:{rank}", seed=0, rank=rank, world=world, update_repeats=2)
    w.iteration()
    torch.cuda.synchronize()
    q.put((rank, w.params.cpu()))
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
    got = dict(q.get() for _ in range(2))
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    assert torch.equal(got[0], got[1])            # same summed gradient bucket, same Adam step on every rank
