#!/usr/bin/env python3
"""bench.py -- Mdisparities/s of the WindowSearch hot path on MI355X (driver contract).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (BlockSearch::computeDisparityMapLeft,
BlockSearch.cpp:24-86, through the C-ABI ws_search_device) over one synthetic
Middlebury-H-shaped pair that is already resident in HBM: BASELINE.json configs[1]
= 1500x1000, 7x7 SSD, D=256, left view, smoothFactor 1.0.  With N ranks every rank
owns its own pair (independent pairs shard with no collective: weak scaling); the
barrier / all_reduce(MAX) below only brackets the timing.

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel
(ws_march_kernel): algorithmic bytes per launch / its average duration measured
with HIP events on the launch stream.  `cpu_baseline` times the CPU oracle
(oracle/, a port: the reference itself cannot be built here) on a bounded row band
of the same workload on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS) + ["config4"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=256, help="rows of the CPU baseline sample")
    ap.add_argument("--check", action="store_true", help="compare a row band with the oracle")
    args = ap.parse_args()

    import torch
    import stereo_reconstruction_amd as ws
    from stereo_reconstruction_amd.synthetic import make_pair

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with that many ranks" % args.gpus)
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU path to time)")
    # WS_BENCH_REHEARSE=1: rehearse the N>1 code path on a box with fewer GPUs than ranks (all
    # ranks share the devices round-robin, gloo instead of RCCL).  Never used for reported numbers.
    rehearse = os.environ.get("WS_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # the extension normally travels with the tree; a tree without it gets it built once per node
    from stereo_reconstruction_amd import build as ws_build
    if not os.path.exists(os.environ.get("WS_STEREO_LIB", ws_build.LIB)):
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            ws_build.build()
        if dist is not None:
            dist.barrier()
    dev = torch.device("cuda", local_rank)
    ctx = ws.WindowSearch(local_rank)
    stream = torch.cuda.current_stream().cuda_stream
    batch = args.workload == "config4"
    if batch:
        from stereo_reconstruction_amd.sharding import lpt_assign
        from stereo_reconstruction_amd.synthetic import TRAINING_H
        bs, cost, max_d = 7, "ssd", 256
        shapes = [(w, h) for _, w, h, _ in TRAINING_H]
        mine = lpt_assign([w * h * max_d for w, h in shapes], world)[rank]
        pairs = []
        for i in mine:
            l, r, _ = make_pair(shapes[i][0], shapes[i][1], max_d, 100 + i)
            pairs.append((torch.from_numpy(l).to(dev), torch.from_numpy(r).to(dev),
                          torch.empty((shapes[i][1], shapes[i][0]), dtype=torch.float32, device=dev)))
        width, height = shapes[mine[0]] if mine else shapes[0]
        left, right = (pairs[0][0].cpu().numpy(), pairs[0][1].cpu().numpy()) if pairs else (None, None)
        t_out = pairs[0][2] if pairs else None
        hyps_total = float(sum(w * h * max_d for w, h in shapes))
    else:
        width, height, bs, cost, max_d, seed = WORKLOADS[args.workload]
        left, right, _gt = make_pair(width, height, max_d, seed + rank)
        t_out = torch.empty((height, width), dtype=torch.float32, device=dev)
        pairs = [(torch.from_numpy(left).to(dev), torch.from_numpy(right).to(dev), t_out)]
        hyps_total = float(width) * height * max_d * world   # one pair per rank (weak scaling)
    params = ws.make_params(ws.VIEW_LEFT, bs, 0, max_d, 1.0, cost)

    def step():
        for tl, tr, to in pairs:   # whole pairs per rank, no collective on the data path
            ctx.search_device(params, tl, tr, to, stream)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    value = hyps_total * args.steps / elapsed / 1e6     # H*W*D hypotheses of the whole job

    # dominant kernel, timed alone with HIP events on the launch stream
    kernel_ms = None
    info = ctx.last_launch()
    ctx.set_profiling(True)
    acc = []
    for _ in range(min(args.steps, 20)):
        step()
        acc.append(ctx.last_kernel_ms())
    ctx.set_profiling(False)
    kernel_ms = float(np.mean(acc))
    # the event pair brackets the LAST pair's marching kernel of a step
    lw, lh = pairs[-1][0].shape[1], pairs[-1][0].shape[0]
    alg_bytes = 3.0 * lh * lw + 3.0 * pairs[-1][1].shape[0] * pairs[-1][1].shape[1] + 4.0 * lh * lw
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9

    # HBM traffic of the dominant kernel: measured separately with rocprofv3 --pmc (one pass per
    # counter) and committed; bench.py does not run the profiler itself
    traffic, valu = None, None
    tfile = os.path.join(ROOT, "profiles", "r01", "traffic_%s.json" % args.workload)
    if os.path.exists(tfile):
        tj = json.load(open(tfile))
        if tj.get("kernel") == info["kernel"]:
            traffic = tj["hbm_bytes_per_launch"]
            if "sq_insts_valu" in tj:
                # the bound that actually applies: VALU instruction issue
                lane_ops = tj["sq_insts_valu"] * 64.0
                valu = {"lane_ops_per_launch": lane_ops,
                        "lane_ops_per_hypothesis": round(lane_ops / (float(lw) * lh * max_d), 2),
                        "achieved_lane_ops_per_s": round(lane_ops / (kernel_ms * 1e-3), 0),
                        "measured_issue_peak_lane_ops_per_s": tj["valu_issue_peak_lane_ops_per_s"],
                        "frac": round(lane_ops / (kernel_ms * 1e-3) / tj["valu_issue_peak_lane_ops_per_s"], 3)}

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
        "data": "synthetic" + (" (REHEARSAL: ranks share GPUs, gloo)" if rehearse else ""),
        "config": {"workload": ("config4: 15 trainingH-shaped BGR pairs sharded over the ranks (LPT), left view, "
                                "%dx%d %s, D=%d, smoothFactor 1.0" % (bs, bs, cost.upper(), max_d)) if batch else
                               "%s: one %dx%d BGR pair per GPU, left view, %dx%d %s, D=%d, smoothFactor 1.0"
                               % (args.workload, width, height, bs, bs, cost.upper(), max_d),
                   "pairs_per_step": 15 if batch else world, "sharding": "independent pairs, no collective"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                     "kernel": info["kernel"], "kernel_ms": round(kernel_ms, 4),
                     "algorithmic_bytes": alg_bytes, "valu_issue": valu,
                     "note": "stencil/reduction with D/10 hypotheses per compulsory byte: "
                             "VALU-issue bound, see DESIGN.md for the lane-op ceiling"},
    }

    if args.check and rank == 0:
        from oracle import oracle
        y0 = height // 2
        ref = oracle.block_left(left, right, bs, 0, max_d, cost=cost, rows=(y0, y0 + 8),
                                threads=host_cores())
        got = t_out[y0:y0 + 8].cpu().numpy().astype(np.float64)
        out["check_rows_equal"] = bool(np.array_equal(got, ref[y0:y0 + 8]))

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle
        cores = host_cores()
        half = (bs - 1) // 2
        rows = (half, min(height - half, half + args.cpu_rows))
        t0 = time.perf_counter()
        oracle.block_left(left, right, bs, 0, max_d, cost=cost, rows=rows, threads=cores)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {
            "value": round((rows[1] - rows[0]) * width * max_d / dt / 1e6, 2),
            "unit": "Mdisparities/s", "cores": cores, "kind": "port",
            "sample": "rows [%d,%d) of the same pair (%.0f%% of it), oracle/ws_oracle.c row-parallel "
                      "over %d threads, %.1f s" % (rows[0], rows[1], 100.0 * (rows[1] - rows[0]) / height,
                                                  cores, dt)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
