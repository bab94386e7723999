#!/usr/bin/env python3
"""bench.py -- Mdisparities/s of the WindowSearch hot path on MI355X (driver contract).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both forms work: started WITHOUT a torch.distributed environment and with --gpus N > 1, this
process only launches the N ranks (fresh child processes, before anything here touches a GPU),
waits for them and passes rank 0's JSON line through.

A "step" is one pass of the hot path (BlockSearch::computeDisparityMapLeft,
BlockSearch.cpp:24-86, through the C-ABI ws_search_device) over one BATCH of synthetic
Middlebury-H-shaped pairs already resident in HBM: BASELINE.json configs[1]
= 1500x1000, 7x7 SSD, D=256, left view, smoothFactor 1.0, 704 pairs per step, every one with its
own bytes in HBM (so that a step is ~0.1 s and the 20 steps the driver times are ~2 s of device
work, visible to its utilisation sampler), consecutive pairs on alternating contexts
(--in-flight 2: one pair's pre-pass overlaps the other's search).  `value_single_pair` is the
like-for-like figure of one pair at a time on one context.
With N ranks every rank owns its own batch (independent pairs shard with no collective:
weak scaling); the barrier / all_reduce(MAX) below only brackets the timing.  `--workload config4` is
BASELINE.json configs[3]: the 15 trainingH-shaped pairs cut into row bands of >= 256 rows and
sharded over the ranks (strong; stereo_reconstruction_amd/sharding.py: band_items).

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel
(ws_march_kernel): algorithmic bytes per launch / its average duration measured
with HIP events on the launch stream, one pair at a time (the kernel alone on the device).  `cpu_baseline` times the CPU oracle
(oracle/, a port: the reference itself cannot be built here) on a bounded row band
of the same workload on this box's host cores, all cores and one core (the reference
is single-threaded).  `e2e` is what a caller of the boundary gets from ONE
ws_search_host call incl. H2D/D2H (never `value`).  `quality` is bad-2.0 (evaldisp,
utils.cpp:123-168) of the device map and of the oracle's map on the two trainingH
scenes whose ground truth the reference tree holds (fixtures under tests/golden/).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
PROFILE_ROUND = "r04"  # profiles/<round>/traffic_<workload>.json: PMC figures of the committed kernels
# the marching kernel's sources: a PMC summary is only replayed into the bench line for the build it was taken from
KERNEL_SOURCES = ["ws_march_kernel.h", "ws_march.hip", "ws_march_nd4.hip", "ws_device.h", "ws_kernels.h"]


def kernel_sources_sha1(read=None):
    """SHA-1 of the marching kernel's sources with comments and whitespace stripped: a reworded comment does not
    orphan the PMC summaries under profiles/, a changed instruction does."""
    import hashlib
    import re
    h = hashlib.sha1()
    for name in KERNEL_SOURCES:
        if read is None:
            with open(os.path.join(ROOT, "stereo_reconstruction_amd", "csrc", name), "rb") as f:
                text = f.read().decode("utf-8", "replace")
        else:
            text = read(name)
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
        text = re.sub(r"//[^\n]*", " ", text)
        h.update(" ".join(text.split()).encode())
    return h.hexdigest()

# A step is one pass of the hot path over one BATCH of pairs of the workload's shape (distinct images, all resident
# in HBM: 10.6 GB at config 2), sized so that a step is ~0.1 s: the 20 steps the driver times after 5 of warm-up are
# then ~2 s of device work -- visible to a utilisation sampler, and long past the clocks coming up (a 0.17 ms pair
# timed 25 times in a row measured them: 2.18e6 Mdisparities/s over 5 + 50 single pairs, 2.33e6 over 200 + 1000).
PAIRS_PER_STEP = {"config2": 704, "config3": 112, "config5": 32, "config1": 4096}
CONFIG4_REPEAT = 100   # config 4: a step is this many passes over the rank's row bands (>= 20 ms a step at 8 ranks)
WORKLOADS = {
    # name: (width, height, block, cost, maxD, seed)
    "config2": (1500, 1000, 7, "ssd", 256, 2),
    "config3": (2964, 1988, 9, "sad", 512, 3),
    "config5": (3840, 2160, 9, "ssd", 1024, 5),
    "config1": (450, 375, 5, "sad", 64, 1),
}
# BASELINE.json configs[3]: the 15 trainingH shapes, 7x7 SSD, D=256, sharded over the ranks
# (strong scaling: the batch is fixed).  `--workload config4`.


def host_cores():
    """CPUs this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS) + ["config4"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the e2e and quality legs (profiling runs)")
    ap.add_argument("--cpu-rows", type=int, default=300,
                    help="rows of the all-cores CPU baseline sample (1/16 of it for the one-core sample): ~4 s together")
    ap.add_argument("--check", action="store_true", help="compare a row band with the oracle")
    ap.add_argument("--pairs-per-step", type=int, default=0,
                    help="pairs a rank searches per step (its batch); 0 = the workload's default")
    ap.add_argument("--in-flight", type=int, default=2,
                    help="contexts (own stream + scratch each) a rank's batch alternates over: with 2, one pair's "
                         "pre-pass and launch gaps run beside the other pair's search")
    return ap.parse_args(argv)


def launch_ranks(n, argv):
    """Start the n ranks as fresh child processes (this process has not touched a GPU and never
    will), one per GPU, wait for them, return the job's exit code.  Rank 0 writes the JSON line to
    the inherited stdout.  A rank that dies takes the others down instead of leaving them at a barrier."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    alive = list(procs)
    while alive:
        time.sleep(0.05)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in alive:       # the exact processes started above
                    q.terminate()
    return rc


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    import numpy as np
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    # WS_BENCH_REHEARSE=1: rehearse the N>1 code path on a box with fewer GPUs than ranks (all
    # ranks share the devices round-robin, gloo instead of RCCL).  =dry: no device at all, the
    # step is a stand-in (CPU test of the launcher and the rank plumbing).  Never for reported numbers.
    rehearse = os.environ.get("WS_BENCH_REHEARSE", "")
    dry = rehearse == "dry"

    import torch
    if not dry and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU path to time)")
    if rehearse and not dry:
        local_rank = local_rank % torch.cuda.device_count()
    if not dry:
        torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import stereo_reconstruction_amd as ws
    from stereo_reconstruction_amd.synthetic import make_pair
    # the extension normally travels with the tree; a tree without it gets it built once per node
    from stereo_reconstruction_amd import build as ws_build
    if not os.path.exists(os.environ.get("WS_STEREO_LIB", ws_build.LIB)):
        if local_rank == 0:
            ws_build.build()
        if dist is not None:
            dist.barrier()
    dev = torch.device("cpu") if dry else torch.device("cuda", local_rank)
    in_flight = max(1, args.in_flight)
    ctxs = [] if dry else [ws.WindowSearch(local_rank) for _ in range(in_flight)]
    ctx = ctxs[0] if ctxs else None
    batch = args.workload == "config4"
    if batch:
        from stereo_reconstruction_amd.sharding import band_items
        from stereo_reconstruction_amd.synthetic import TRAINING_H
        bs, cost, max_d = 7, "ssd", 256
        half = (bs - 1) // 2
        shapes = [(w, h) for _, w, h, _ in TRAINING_H]
        items, shards = band_items(shapes, max_d, world, bs)          # (pair, y0, y1) row bands, >= 256 rows each
        mine = shards[rank]                                           # may be empty (more ranks than bands)
        todo = [(shapes[items[j][0]][0], shapes[items[j][0]][1], 100 + items[j][0]) for j in mine]
        bands = [(items[j][1], items[j][2]) for j in mine]
        hyps_total = float(sum(w * h * max_d for w, h in shapes)) * CONFIG4_REPEAT
        width, height = shapes[0]
        pps = 0
    else:
        width, height, bs, cost, max_d, seed = WORKLOADS[args.workload]
        half = (bs - 1) // 2
        pps = args.pairs_per_step if args.pairs_per_step > 0 else PAIRS_PER_STEP[args.workload]
        mine = [rank * pps + i for i in range(pps)]
        todo = [(width, height, seed + rank)] * pps
        bands = [(0, height)] * pps
        hyps_total = float(width) * height * max_d * pps * world   # the same batch per rank (weak scaling)
    host_pairs, pairs = [], []
    made = {}
    for k, (w, h, sd) in enumerate(todo):
        y0, y1 = bands[k]
        a, b = max(0, y0 - half), min(h, y1 + half)   # the band's sub-images: its map rows + the window's halo
        if dry:
            host_pairs.append((None, None))
            pairs.append((w, b - a))
            continue
        if (w, h, sd) not in made:
            l, r, _ = make_pair(w, h, max_d, sd)
            made[(w, h, sd)] = (l, r, torch.from_numpy(l).to(dev), torch.from_numpy(r).to(dev))
            host_pairs.append((l, r))
            tl, tr = made[(w, h, sd)][2:]
        elif batch:
            tl, tr = made[(w, h, sd)][2:]             # another band of a pair this rank already holds
        else:
            # further pairs of the batch: the same scene shifted down by a few rows (both views alike, so it
            # still is a rectified pair) -- different bytes in every buffer without a second of numpy per pair
            tl, tr = (torch.roll(t, 37 * k, 0).contiguous() for t in made[(w, h, sd)][2:])
        pairs.append((tl[a:b], tr[a:b], torch.empty((b - a, w), dtype=torch.float32, device=dev)))
    params = None if dry else ws.make_params(ws.VIEW_LEFT, bs, 0, max_d, 1.0, cost)

    def step():
        if dry:
            time.sleep(1e-4 * len(pairs))
            return
        # whole pairs per rank, no collective on the data path; consecutive pairs on alternating contexts
        # (each on its own stream, with its own scratch planes)
        for _ in range(CONFIG4_REPEAT if batch else 1):
            for k, (tl, tr, to) in enumerate(pairs):
                ctxs[k % in_flight].search_device(params, tl, tr, to, None)

    def sync():
        if not dry:
            torch.cuda.synchronize()

    def barrier():
        sync()
        if dist is not None:
            dist.barrier()
        sync()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    own_elapsed = time.perf_counter() - t0      # this rank's own work, before it waits for the others
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    value = hyps_total * args.steps / elapsed / 1e6     # H*W*D hypotheses of the whole job

    # dominant kernel, timed alone with HIP events on the launch stream: every pair of this rank
    reps = max(1, min(args.steps, 20))
    sampled = pairs[:32]               # (a sample of the batch: every pair of it has the same shape, except config 4's)
    if batch:
        sampled = pairs
    kernel_ms = [0.0] * len(sampled)   # average duration of the marching kernel, per sampled pair of this rank
    info = {"kernel": "", "threads": 0, "workgroups": 0, "lds_bytes": 0}
    single_pair_ms = None
    if not dry and pairs:
        ctx.set_profiling(True)
        for _ in range(reps):
            for i, (tl, tr, to) in enumerate(sampled):
                ctx.search_device(params, tl, tr, to, None)   # one pair at a time: the kernel alone on the device
                kernel_ms[i] += ctx.last_kernel_ms() / reps
        ctx.set_profiling(False)
        info = ctx.last_launch()
        if not batch:
            # the like-for-like figure of earlier rounds: ONE context, one pair after the other (what --in-flight 1
            # times), after all of the above as warm-up
            n1 = max(64, min(len(pairs), int(0.3 / max(1e-6, sum(kernel_ms) / len(kernel_ms) * 1e-3))))
            sync()
            t1 = time.perf_counter()
            for i in range(n1):
                tl, tr, to = pairs[i % len(pairs)]
                ctx.search_device(params, tl, tr, to, None)
            sync()
            single_pair_ms = (time.perf_counter() - t1) / n1 * 1e3
    # ... and the way the timed region runs it: with the other context's pair beside it (the last launch of
    # every context in each of a few steps; the events sit on the launch streams and nothing waits inside a step)
    kernel_ms_in_flight = None
    if not dry and pairs and in_flight > 1 and len(pairs) >= in_flight:
        for c in ctxs:
            c.set_profiling(True)
        samples = []
        for _ in range(min(reps, 3)):
            step()
            samples.extend(c.last_kernel_ms() for c in ctxs)
        for c in ctxs:
            c.set_profiling(False)
        kernel_ms_in_flight = sum(samples) / len(samples)
    # both images + the f32 map of what was sampled (config 4: the rank's bands incl. their halo rows)
    rows_of = [(min(h, y1 + half) - max(0, y0 - half)) for (w, h, _), (y0, y1) in zip(todo, bands)][:len(sampled)]
    alg_bytes = [10.0 * rows * w for rows, (w, h, _) in zip(rows_of, todo)]
    summary = {"rank": rank, "pairs": list(mine)[:len(sampled)], "kernel_ms_sum": round(sum(kernel_ms), 4),
               "alg_bytes": sum(alg_bytes), "hyps": float(sum(rows * w * max_d for rows, (w, h, _) in zip(rows_of, todo))),
               "items": len(pairs), "wall_ms": round(own_elapsed * 1e3, 3),
               # which device this rank is bound to, what the library sees, which host: a straggler is traced from here
               "device": None if dry else {"local_rank": local_rank, "ws_device_count": ws.device_count(),
                                           "name": torch.cuda.get_device_name(local_rank), "host": os.uname().nodename,
                                           "copy_threads_env": os.environ.get("WS_COPY_THREADS"), "local_world_size": os.environ.get("LOCAL_WORLD_SIZE")},
               # config 4: the row bands this rank searched, (pair, first row, last row + 1)
               "bands": [[int(items[j][0]), int(items[j][1]), int(items[j][2])] for j in mine] if batch else None,
               "map_hyps": float(sum((y1 - y0) * w * max_d for (w, h, _), (y0, y1) in zip(todo, bands)))}
    if dist is not None:
        summaries = [None] * world
        dist.all_gather_object(summaries, summary)
    else:
        summaries = [summary]

    out = None
    if rank == 0:
        # roofline of the dominant kernel over the launches of the busiest rank (config 2: the one launch)
        busiest = max(summaries, key=lambda s: s["kernel_ms_sum"])
        k_ms, k_bytes = busiest["kernel_ms_sum"], busiest["alg_bytes"]
        n_launch = max(1, len(busiest["pairs"]))
        achieved = k_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        # HBM traffic / VALU instructions of the dominant kernel: measured with rocprofv3 --pmc (separate
        # passes), summarised by tools/pmc_to_traffic.py into profiles/<round>/; bench.py does not run the profiler
        traffic, valu = None, None
        tfile = os.path.join(ROOT, "profiles", PROFILE_ROUND, "traffic_%s.json" % args.workload)
        if os.path.exists(tfile) and k_ms > 0:
            tj = json.load(open(tfile))
            # replayed only for the kernel that ran AND the sources it was profiled from (tools/pmc.sh records their hash)
            if tj.get("kernel") == info["kernel"] and tj.get("kernel_sources_sha1") == kernel_sources_sha1():
                traffic = tj["hbm_bytes_per_launch"]
                if "sq_insts_valu" in tj:
                    # the bound that actually applies: VALU instruction issue
                    lane_ops = tj["sq_insts_valu"] * 64.0
                    per_launch_ms = k_ms / n_launch
                    valu = {"from_profiles": True, "lane_ops_per_launch": lane_ops,
                            "lane_ops_per_hypothesis": round(lane_ops / (busiest["hyps"] / n_launch), 2),
                            "achieved_lane_ops_per_s": round(lane_ops / (per_launch_ms * 1e-3), 0),
                            "measured_issue_peak_lane_ops_per_s": tj["valu_issue_peak_lane_ops_per_s"],
                            "frac": round(lane_ops / (per_launch_ms * 1e-3) / tj["valu_issue_peak_lane_ops_per_s"], 3),
                            "source": "profiles/%s/traffic_%s.json" % (PROFILE_ROUND, args.workload)}
        out = {
            "metric": "Mdisparities/s (HxWxD / s) on Middlebury-H pairs",
            "value": round(value, 1),
            "unit": "Mdisparities/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong" if batch else "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic" + ((" (REHEARSAL: %s)" % ("no device, stand-in step" if dry else "ranks share GPUs, gloo"))
                                   if rehearse else ""),
            "config": {"workload": ("config4: 15 trainingH-shaped BGR pairs in row bands of >= 256 rows sharded over the ranks "
                                    "(equal-weight runs of rows), %d passes per step, left view, %dx%d %s, D=%d, "
                                    "smoothFactor 1.0" % (CONFIG4_REPEAT, bs, bs, cost.upper(), max_d)) if batch else
                                   "%s: a batch of %d %dx%d BGR pairs per GPU and step, left view, %dx%d %s, D=%d, smoothFactor 1.0"
                                   % (args.workload, pps, width, height, bs, bs, cost.upper(), max_d),
                       "pairs_per_step": 15 * CONFIG4_REPEAT if batch else pps * world, "in_flight_per_gpu": in_flight,
                       "sharding": "row bands of independent pairs, no collective" if batch else "independent pairs, no collective"},
            # `bound`: what binds the kernel (VALU instruction issue: D/10 hypotheses per compulsory byte); achieved /
            # peak / frac are the HBM figures the contract asks for: algorithmic bytes per launch / kernel time
            "roofline": {"bound": "valu_issue", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "traffic_from_profiles": ("none: no PMC summary under profiles/%s/ for this kernel and these kernel sources"
                                                   % PROFILE_ROUND) if traffic is None else
                                                  "profiles/%s/traffic_%s.json (rocprofv3 --pmc passes of this kernel built from these "
                                                  "sources -- kernel symbol and source hash match --, not collected in this run)"
                                                  % (PROFILE_ROUND, args.workload),
                         "kernel": info["kernel"], "kernel_ms": round(k_ms / n_launch, 4),
                         "kernel_ms_in_flight": None if kernel_ms_in_flight is None else round(kernel_ms_in_flight, 4),
                         "launches": n_launch, "algorithmic_bytes": k_bytes / n_launch, "valu_issue_from_profiles": valu,
                         "note": "stencil/reduction with D/10 hypotheses per compulsory byte: VALU-issue bound, see "
                                 "DESIGN.md for the lane-op ceiling; kernel_ms / achieved: the kernel alone on the "
                                 "device, kernel_ms_in_flight: a launch of the timed region, sharing the chip with "
                                 "the other context's pair (a rocprofv3 average of the default command mixes both)"},
        }
        if single_pair_ms is not None:
            out["value_single_pair"] = round(float(width) * height * max_d / single_pair_ms / 1e3, 1)
            out["single_pair"] = {"ms_per_pair": round(single_pair_ms, 4), "what": "one context, one pair after the other "
                                  "(--in-flight 1 semantics; BENCH_r01 timed this way), after the timed region as warm-up"}
        if batch or world > 1:
            out["per_rank"] = [{"rank": s["rank"], "items": s["items"], "kernel_ms_sum": s["kernel_ms_sum"],
                                "wall_ms": s["wall_ms"], "Mdisp_map": round(s["map_hyps"] / 1e6, 1),
                                "device": s.get("device"), "bands": s.get("bands")} for s in summaries]

    single = rank == 0 and world == 1 and not dry
    left, right = host_pairs[0] if host_pairs else (None, None)
    if args.check and single:
        from oracle import oracle
        y0 = height // 2
        ref = oracle.block_left(left, right, bs, 0, max_d, cost=cost, rows=(y0, y0 + 8), threads=host_cores())
        got = pairs[0][2][y0:y0 + 8].cpu().numpy().astype(np.float64)
        out["check_rows_equal"] = bool(np.array_equal(got, ref[y0:y0 + 8]))

    if single and not args.no_extras and not batch:
        out["e2e"] = e2e_leg(ws, ctx, params, left, right, width, height, max_d)
        q = quality_leg(ws, ctx)
        if q is not None:
            out["quality"] = q

    if single and not args.no_cpu_baseline and not batch:
        out["cpu_baseline"] = cpu_baseline_leg(left, right, bs, cost, max_d, width, height, args.cpu_rows)

    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def e2e_leg(ws, ctx, params, left, right, width, height, max_d):
    """One boundary call as the reference's caller makes it: host images in, host map out
    (ws_search_host; CV_64F as BlockSearch returns it, and f32), PCIe both ways included.  The library cuts
    a call of this size into row bands whose copies overlap the searches; `plain` is the same call unsplit."""
    import numpy as np
    res = {"what": "one ws_search_host call: pageable host images -> host map, H2D + kernels + D2H + sync; "
                   "median of 15, output array kept by the caller"}
    for name, dt in (("f64", np.float64), ("f32", np.float32)):
        keep = np.empty((height, width), dtype=dt)
        row = {}
        for mode, nb in (("ms_per_call", -1), ("plain_ms_per_call", 0)):
            ctx.set_host_bands(nb)
            for _ in range(3):
                ctx.search(params, left, right, dtype=dt, out=keep)
            ts = []
            for _ in range(15):
                t0 = time.perf_counter()
                ctx.search(params, left, right, dtype=dt, out=keep)
                ts.append(time.perf_counter() - t0)
            row[mode] = round(float(np.median(ts)) * 1e3, 4)
        ctx.set_host_bands(-1)
        row["Mdisparities_per_s"] = round(width * height * max_d / row["ms_per_call"] / 1e3, 1)
        res[name] = row
    return res


def quality_leg(ws, ctx):
    """bad-2.0 of the left-view map at the reference's own scale: 7x7 SSD, D = ndisp of calib.txt, on
    Teddy-H (the scene main.cpp:20 runs) and ArtL; device map vs oracle map (also compared bit for bit)."""
    import numpy as np
    from oracle import oracle
    res = {}
    for scene, fn in (("Teddy", "teddyH_pair.npz"), ("ArtL", "artL_pair.npz")):
        path = os.path.join(ROOT, "tests", "golden", fn)
        if not os.path.exists(path):
            return None
        z = np.load(path)
        l, r, gt, mask, nd = z["left"], z["right"], z["gt"], z["mask"], int(z["ndisp"])
        got = ws.BlockSearch(l, r, 7, 0, nd, cost="ssd", context=ctx).computeDisparityMapLeft(1.0)
        want = oracle.block_left(l, r, 7, 0, nd, cost="ssd", threads=host_cores())
        eg = ws.evaldisp(got, gt, mask, 2.0, float(nd))
        ec = oracle.evaldisp(want, gt, mask, 2.0, float(nd))
        res[scene] = {"shape": [int(l.shape[1]), int(l.shape[0])], "ndisp": nd,
                      "bad2.0_gpu": round(eg["bad"], 4), "bad2.0_cpu": round(ec["bad"], 4),
                      "maps_identical": bool(np.array_equal(got, want))}
    res["what"] = "evaldisp(map, disp0GT, mask0nocc, 2.0, ndisp, 0) 'bad' percent, left view 7x7 SSD (utils.cpp:123-168)"
    return res


def cpu_baseline_leg(left, right, bs, cost, max_d, width, height, cpu_rows):
    """The CPU oracle (a port of BlockSearch.cpp:24-86) on row bands of the same pair: all host cores
    (row-parallel, legal for smoothFactor 1) and ONE core, which is what the reference itself uses."""
    from oracle import oracle
    cores = host_cores()
    half = (bs - 1) // 2
    rows = (half, min(height - half, half + cpu_rows))
    t0 = time.perf_counter()
    oracle.block_left(left, right, bs, 0, max_d, cost=cost, rows=rows, threads=cores)
    dt = time.perf_counter() - t0
    rows1 = (half, min(height - half, half + max(4, cpu_rows // 16)))
    t0 = time.perf_counter()
    oracle.block_left(left, right, bs, 0, max_d, cost=cost, rows=rows1, threads=1)
    dt1 = time.perf_counter() - t0
    return {
        "value": round((rows[1] - rows[0]) * width * max_d / dt / 1e6, 2),
        "unit": "Mdisparities/s", "cores": cores, "kind": "port",
        "sample": "rows [%d,%d) of the same pair (%.0f%% of it), oracle/ws_oracle.c row-parallel "
                  "over %d threads, %.1f s" % (rows[0], rows[1], 100.0 * (rows[1] - rows[0]) / height, cores, dt),
        "single_thread": {
            "value": round((rows1[1] - rows1[0]) * width * max_d / dt1 / 1e6, 2), "unit": "Mdisparities/s",
            "cores": 1, "sample": "rows [%d,%d) of the same pair, one thread (the reference is single-threaded, "
                                  "BlockSearch.cpp:36-84), %.1f s" % (rows1[0], rows1[1], dt1)}}


if __name__ == "__main__":
    main()
