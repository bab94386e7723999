"""The N>1 path on CPU: world_size-2 gloo processes shard independent pairs with no data-path
collective; only the timing bracket (barrier + MAX) and a summary gather communicate."""
import os
import socket
import subprocess
import sys
import textwrap

from conftest import ROOT
from stereo_reconstruction_amd.sharding import band_items, lpt_assign
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


def test_band_items_cover_every_row_once_and_balance():
    """Row bands of >= 256 rows (smoothFactor 1: a map row only depends on the image rows under its window,
    BlockSearch.cpp:46-66): every row of every pair in exactly one item, maximum load within 3 % of the mean at 2 / 4 /
    8 ranks on the trainingH shapes (whole pairs: 1.01 / 1.13 / 1.15)."""
    shapes = [(w, h) for _, w, h, _ in TRAINING_H]
    half = 3
    for world in (1, 2, 3, 4, 8, 16, 40):
        items, shards = band_items(shapes, 256, world, 7)
        assert sorted(j for sh in shards for j in sh) == list(range(len(items))) and len(shards) == world
        rows = {}
        for i, y0, y1 in items:
            assert 0 <= y0 < y1 <= shapes[i][1]
            rows.setdefault(i, []).append((y0, y1))
        assert sorted(rows) == list(range(len(shapes)))
        for i, bands in rows.items():
            bands.sort()
            assert bands[0][0] == 0 and bands[-1][1] == shapes[i][1]
            assert all(a[1] == b[0] for a, b in zip(bands, bands[1:]))
            assert len(bands) == 1 or min(y1 - y0 for y0, y1 in bands) >= 256
        cost = [(min(shapes[i][1], y1 + half) - max(0, y0 - half)) * shapes[i][0] * 256.0 for i, y0, y1 in items]
        loads = [sum(cost[j] for j in sh) for sh in shards]
        if world in (2, 4, 8):
            assert max(loads) <= 1.03 * sum(cost) / world, (world, max(loads) / (sum(cost) / world))
    assert band_items(shapes, 256, 4, 7) == band_items(shapes, 256, 4, 7)      # deterministic
    assert band_items([], 256, 3, 7) == ([], [[], [], []])
    items, shards = band_items([(100, 40)], 64, 4, 7)                           # too short to cut: one rank has it all
    assert items == [(0, 0, 40)] and sum(len(sh) for sh in shards) == 1


BAND_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    from stereo_reconstruction_amd.sharding import band_items, gather_objects
    from stereo_reconstruction_amd.synthetic import make_pair
    from oracle import oracle        # (test infrastructure: the stand-in for the per-band device call)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    shapes, bs, D = [(180, 150), (140, 170), (200, 131)], 7, 24
    half = (bs - 1) // 2
    pairs = [make_pair(w, h, D, 40 + i)[:2] for i, (w, h) in enumerate(shapes)]
    items, shards = band_items(shapes, D, world, bs, min_rows=40)
    done = {}
    for j in shards[rank]:
        i, y0, y1 = items[j]
        a, b = max(0, y0 - half), min(shapes[i][1], y1 + half)
        sub = oracle.block_left(pairs[i][0][a:b], pairs[i][1][a:b], bs, 0, D)      # a search on the band's sub-images
        done[j] = sub[y0 - a:y1 - a]
        # a band that touches the image's first / last rows keeps their ring of zeros; elsewhere the sub-image's
        # own ring rows are halo and thrown away
    parts = gather_objects(done, dist, world)
    if rank == 0:
        for i, (w, h) in enumerate(shapes):
            full = np.full((h, w), np.nan)
            for d in parts:
                for j, rows in d.items():
                    if items[j][0] == i:
                        assert np.isnan(full[items[j][1]:items[j][2]]).all()         # disjoint rows
                        full[items[j][1]:items[j][2]] = rows
            want = oracle.block_left(pairs[i][0], pairs[i][1], bs, 0, D)
            assert np.array_equal(full, want), i
        assert len(items) > len(shapes)                                              # something really was cut
        print("OK", [len(sh) for sh in shards])
    dist.barrier()
    dist.destroy_process_group()
""")


def _run_two_ranks(tmp_path, text):
    script = tmp_path / "worker.py"
    script.write_text(text % ROOT)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "OK" in outs[0], outs


def test_two_rank_gloo_job_assembles_identical_maps_from_bands(tmp_path, oracle):
    """World size 2, gloo: the ranks search disjoint row bands (the oracle stands in for the device call), rank 0
    assembles the maps from the gathered bands: identical to the whole-image searches."""
    _run_two_ranks(tmp_path, BAND_WORKER)


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
