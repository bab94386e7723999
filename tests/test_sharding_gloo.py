"""The N>1 path on CPU: world_size-2 gloo processes shard independent pairs with no data-path
collective; only the timing bracket (barrier + MAX) and a summary gather communicate."""
import os
import socket
import subprocess
import sys
import textwrap

from conftest import ROOT
from stereo_reconstruction_amd.sharding import lpt_assign
from stereo_reconstruction_amd.synthetic import TRAINING_H


def test_lpt_assign_covers_every_pair_once_and_balances():
    costs = [w * h * 256 for _, w, h, _ in TRAINING_H]
    for world in (1, 2, 4, 8):
        shards = lpt_assign(costs, world)
        assert sorted(i for s in shards for i in s) == list(range(len(costs)))
        loads = [sum(costs[i] for i in s) for s in shards]
        assert max(loads) <= sum(costs) / world + max(costs)
    assert lpt_assign(costs, 2) == lpt_assign(costs, 2)     # deterministic
    assert lpt_assign([], 3) == [[], [], []]


WORKER = textwrap.dedent("""
    import os, sys, time
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from stereo_reconstruction_amd.sharding import lpt_assign, timed_region, gather_objects
    from stereo_reconstruction_amd.synthetic import TRAINING_H
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    costs = [w * h * 256 for _, w, h, _ in TRAINING_H]
    mine = lpt_assign(costs, world)[rank]
    done = []
    def work():
        for i in mine:            # stand-in for the per-pair device call: no collective inside
            done.append(i)
        time.sleep(0.05 * (rank + 1))
    elapsed = timed_region(work, lambda: None, dist, None)
    summary = gather_objects({"rank": rank, "pairs": done, "hyps": sum(costs[i] for i in done)}, dist, world)
    if rank == 0:
        allp = sorted(i for s in summary for i in s["pairs"])
        assert allp == list(range(len(costs))), allp
        assert sum(s["hyps"] for s in summary) == sum(costs)
        assert elapsed >= 0.05 * world - 1e-3, elapsed      # MAX over ranks, not rank 0's own time
        print("OK", elapsed, [len(s["pairs"]) for s in summary])
    dist.barrier()
    dist.destroy_process_group()
""")


def test_two_rank_gloo_job(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "OK" in outs[0]
