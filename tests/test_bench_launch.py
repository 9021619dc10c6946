"""CPU: `python bench.py --gpus N` launches its own N ranks.  `--dry-run` runs the launcher,
the file-store rendezvous and ONE flat-bucket SUM all-reduce of the policy-gradient bucket (3,480,775 fp32) over
gloo -- the path the driver's `--gpus 2/4/8` runs takes before any GPU work -- and the torchrun-style env launch."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(out: str):
    return json.loads([l for l in out.splitlines() if l.startswith("{")][-1])


@pytest.mark.timeout(300)
def test_bench_self_launches_two_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"],
                       capture_output=True, text=True, timeout=280, cwd=ROOT,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert r.returncode == 0, r.stdout + r.stderr
    line = _last_json(r.stdout)
    assert line == {"dry_run": True, "n_gpus": 2, "rccl_ranks": 2, "bucket_elems": 3480775, "allreduce_ok": True}


@pytest.mark.timeout(300)
def test_bench_single_rank_and_world_mismatch():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run"], capture_output=True, text=True,
                       timeout=120, cwd=ROOT,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert r.returncode == 0 and _last_json(r.stdout)["n_gpus"] == 1, r.stdout + r.stderr
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"], capture_output=True,
                       text=True, timeout=120, cwd=ROOT, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stdout + r.stderr)


def test_bench_defaults_are_the_metric_configuration():
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse_args([])
    assert (a.gpus, a.actors, a.rollout, a.update_repeats, a.scaling, a.encoder) == (1, 256, 128, 4, "strong", "rn50")
    a = bench.parse_args(["--gpus", "8", "--actors-total", "512"])
    assert a.actors_total == 512 and a.scaling == "strong"       # config 4: 64 actors per GPU


def test_secondary_legs_fail_soft_and_under_a_watchdog():
    """nothing a secondary leg raises may cost the headline line, and a hung leg is cut off by the
    watchdog, which prints the line with what is there."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench._soft("weak", lambda: {"value": 1.0}) == {"value": 1.0}
    r = bench._soft("weak", lambda: 1 / 0)
    assert set(r) == {"error"} and "ZeroDivisionError" in r["error"]
    # another rank failed (agree = MIN over ranks of the ok flags): this rank's result is withdrawn
    assert "another rank" in bench._soft("weak", lambda: {"value": 1.0}, agree=lambda ok: 0.0)["error"]
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "out = {'value': 1.0}\n"
            "def emit(leg=None):\n"
            "    if leg: out.setdefault(leg, {'error': 'timeout'})\n"
            "    print(__import__('json').dumps(out), flush=True)\n"
            "d = bench._Watchdog(0.5, emit); d.leg = 'weak'; time.sleep(30); print('not reached')\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "not reached" not in r.stdout
    assert _last_json(r.stdout) == {"value": 1.0, "weak": {"error": "timeout"}}


def test_busy_union_of_launch_intervals():
    """roofline.frac is quoted over the UNION of all encoder launch intervals (bench.busy_union_ms): overlapping launches of two
    streams count once, idle gaps (the update phase) not at all."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.busy_union_ms([]) == 0.0
    assert bench.busy_union_ms([(0.0, 2.0)]) == 2.0
    assert bench.busy_union_ms([(0.0, 2.0), (1.0, 3.0), (2.5, 2.75)]) == 3.0            # two streams overlapping + one nested
    assert bench.busy_union_ms([(5.0, 6.0), (0.0, 1.0), (1.0, 2.0)]) == 3.0             # unsorted, touching, a gap of 3
    assert abs(bench.busy_union_ms([(i * 1.0, i * 1.0 + 0.75) for i in range(8)]) - 6.0) < 1e-12


def test_only_the_json_line_reaches_stdout():
    """Native libraries print to file descriptor 1 behind Python's back (RCCL's version banner goes through C stdio and is
    flushed at exit, i.e. after the line).  bench.py claims the real stdout for the line alone: whatever else writes to fd 1
    afterwards lands on stderr."""
    code = ("import os, sys, json; sys.path.insert(0, %r); import bench\n"
            "bench._print_line({'a': 1})\n"
            "os.write(1, b'banner through fd 1\\n')\n"          # a C library's write(1, ...)
            "print('python print after the claim')\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout == '{"a": 1}\n', r.stdout
    assert "banner through fd 1" in r.stderr and "python print after the claim" in r.stderr
