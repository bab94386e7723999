"""bench.py starts its own ranks for --gpus N > 1 (the driver's command form is `python3 bench.py --gpus N`):
N fresh child processes before anything touches a GPU, rank 0's JSON line passed through, a failing rank
fails the job.  Rehearsed here without a device (WS_BENCH_REHEARSE=dry: gloo, stand-in step)."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def run_bench(*argv, env_extra=None, timeout=300):
    env = dict(os.environ, WS_BENCH_REHEARSE="dry")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=timeout)
    return p


def json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_self_launch_two_ranks_weak_scaling():
    p = run_bench("--gpus", "2", "--steps", "3", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    out = json_line(p.stdout)
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak"
    # a batch of 704 config-2 pairs per rank and step (bench.PAIRS_PER_STEP), alternating over two contexts
    assert out["config"]["pairs_per_step"] == 1408 and out["config"]["in_flight_per_gpu"] == 2
    assert [r["items"] for r in out["per_rank"]] == [704, 704]
    assert [r["rank"] for r in out["per_rank"]] == [0, 1] and all(r["wall_ms"] > 0 for r in out["per_rank"])
    assert out["value"] > 0 and "REHEARSAL" in out["data"]


def test_self_launch_config4_row_bands_over_many_ranks():
    """16 ranks, 15 pairs: the pairs are cut into row bands (sharding.band_items), every rank joins every barrier,
    the bands' map rows add up to the 15 whole maps, every rank reports its own wall time."""
    from stereo_reconstruction_amd.synthetic import TRAINING_H
    p = run_bench("--gpus", "16", "--steps", "2", "--warmup", "1", "--workload", "config4", timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json_line(p.stdout)
    assert out["n_gpus"] == 16 and out["scaling"] == "strong"
    assert sum(r["items"] for r in out["per_rank"]) > 15 and min(r["items"] for r in out["per_rank"]) >= 1
    total = sum(w * h * 256 for _, w, h, _ in TRAINING_H) / 1e6
    assert abs(sum(r["Mdisp_map"] for r in out["per_rank"]) - total) < 1.0
    assert all(r["wall_ms"] > 0 for r in out["per_rank"])


def test_a_failing_rank_fails_the_job():
    p = run_bench("--gpus", "2", "--workload", "no-such-workload")
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
