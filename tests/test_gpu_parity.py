"""Parity of the HIP path with the oracle -- the tests proper (run with -m gpu on an MI355X).

Everything calls through the C-ABI (ws_search_host / ws_search_device / ws_enqueue_host via the
ctypes layer, or the C++ facade).  Integer SSD / SAD disparity maps must be BIT-IDENTICAL to the
CPU oracle (BASELINE.json north_star); sub-pixel floats within 1e-4 (tolerance stated there).
"""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, golden_cases, load_golden
from stereo_reconstruction_amd.synthetic import make_pair


pytestmark = pytest.mark.gpu

SUBPIXEL_TOL = 1e-4     # north_star: "within 1e-4 for float"


def run(ws, ctx, view, left, right, bs, mind, maxd, cost, subpixel=False):
    b = ws.BlockSearch(left, right, bs, mind, maxd, cost=cost, subpixel=subpixel, context=ctx)
    return b.computeDisparityMapLeft(1.0) if view == "left" else b.computeDisparityMapRight(1.0)


def ref(oracle, view, left, right, bs, mind, maxd, cost, subpixel=False, rows=None):
    f = oracle.block_left if view == "left" else oracle.block_right
    return f(left, right, bs, mind, maxd, cost=cost, subpixel=subpixel, rows=rows,
             threads=8)


@pytest.mark.parametrize("name", golden_cases())
def test_golden_fixtures(wslib, gpu_ctx, name):
    g = load_golden(name)
    s = float(g.get("smooth", 1.0))            # round-2 fixtures (tools/make_golden_r2.py) carry smoothFactor / varBlock
    if g["view"] == "linear":
        out = wslib.LinearSearch(g["left"], g["right"], context=gpu_ctx,
                                 search_range=g["max_disparity"]).computeDisparityMap(s)
    else:
        b = wslib.BlockSearch(g["left"], g["right"], g["block_size"], g["min_disparity"], g["max_disparity"], cost=g["cost"],
                              context=gpu_ctx)
        if g["view"] == "left":
            out = b.computeDisparityMapLeft(s)
        else:
            out = b.computeDisparityMapRight(s, bool(g.get("var_block", 0)), float(g.get("thres", 19.0)))
            if g.get("var_block", 0):
                assert gpu_ctx.last_max_block(g["block_size"]) == g["max_block"]
    assert out.dtype == np.float64
    assert np.array_equal(out, g["expected"].astype(np.float64))


@pytest.mark.parametrize("view", ["left", "right"])
@pytest.mark.parametrize("cost", ["ssd", "sad"])
@pytest.mark.parametrize("bs", [3, 5, 7, 9, 11, 13, 15, 17])
def test_marching_kernel_matches_oracle(wslib, gpu_ctx, oracle, view, cost, bs):
    left, right, _ = make_pair(331, 75, 70, seed=10 * bs + (view == "left"))
    left[30, 100] = 0
    right[31, 101] = 0
    got = run(wslib, gpu_ctx, view, left, right, bs, 0, 70, cost)
    assert "march" in gpu_ctx.last_launch()["kernel"]
    assert np.array_equal(got, ref(oracle, view, left, right, bs, 0, 70, cost))


@pytest.mark.parametrize("view", ["left", "right"])
@pytest.mark.parametrize("bs", [7, 9])
def test_halo_exchange_sad_kernels(wslib, gpu_ctx, oracle, view, bs):
    """Packed SAD, windows 6 .. 9 wide: a thread's chains stop after its own 8 columns and the rest of a window comes
    from the thread to the right (march_pk_halo, 16 disparities per thread); tiles overlap by one run.  The planner
    takes that kernel where the chip is full (a small search is quicker with the plain one's wider workgroups), so the
    images here are large and the oracle checks row bands over the full width: every tile seam, the image's left edge
    (masked candidates), ties (few grey levels), two d-group passes, 8 runs per tile."""
    import torch
    rng = np.random.default_rng(bs)

    def on_device(left, right, bs_, dmin, dmax, cost):
        # (device-resident: the host calls cut a large pair into row bands, each a small search of its own)
        p = wslib.make_params(wslib.VIEW_LEFT if view == "left" else wslib.VIEW_RIGHT, bs_, dmin, dmax, 1.0, cost)
        ref_img = left if view == "left" else right
        to = torch.empty(ref_img.shape[:2], dtype=torch.float32, device="cuda")
        gpu_ctx.search_device(p, torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda(), to, None)
        torch.cuda.synchronize()
        return to.cpu().numpy().astype(np.float64)

    cases = [(2000, 600, 0, 512, 256), (2400, 300, 0, 512, 3), (2000, 600, 0, 500, 2), (3000, 400, 0, 1024, 256), (2600, 500, 40, 552, 256)]
    half = (bs - 1) // 2
    used = 0
    for (w, h, dmin, dmax, levels) in cases:
        if levels == 256:
            left, right, _ = make_pair(w, h, min(dmax, w // 3), seed=w + bs)
        else:
            left = (rng.integers(0, levels, size=(h, w, 3)) * (255 // (levels - 1))).astype(np.uint8)
            right = (rng.integers(0, levels, size=(h, w, 3)) * (255 // (levels - 1))).astype(np.uint8)
        left[h // 2, w // 3] = 0
        got = on_device(left, right, bs, dmin, dmax, "sad")
        used += "halo" in gpu_ctx.last_launch()["kernel"]      # (the planner picks it for three to five of these)
        fn = oracle.block_left if view == "left" else oracle.block_right
        for y0 in (half, h // 2, h - half - 2):
            band = fn(left, right, bs, dmin, dmax, cost="sad", rows=(y0, y0 + 2), threads=8)
            assert np.array_equal(got[y0:y0 + 2], band[y0:y0 + 2]), (w, h, dmin, dmax, levels, y0)
    assert used >= 3, used
    # the narrow windows keep the plain kernels
    left, right, _ = make_pair(2000, 600, 200, seed=3)
    on_device(left, right, 5, 0, 512, "sad")
    assert "halo" not in gpu_ctx.last_launch()["kernel"]


@pytest.mark.parametrize("view", ["left", "right"])
def test_halo_exchange_ssd_kernels(wslib, gpu_ctx, oracle, view):
    """The fused SSD chain with the same exchange (march_fused_ssd_halo; the planner takes it for the 8 / 9 wide windows of
    large searches -- config 5's own test runs that plan): against oracle row bands over the full width, and -- whole maps --
    against the plain kernel; a caller's tile width of 16 or 8 runs selects the halo kernel, any other the plain one."""
    import torch
    vw = wslib.VIEW_LEFT if view == "left" else wslib.VIEW_RIGHT
    for (w, h, bs, dmin, dmax, runs) in ((2400, 700, 9, 0, 512, 16), (3000, 500, 9, 7, 1031, 16), (1500, 400, 7, 0, 256, 16),
                                         (1700, 300, 7, 0, 300, 8)):
        left, right, _ = make_pair(w, h, min(dmax, w // 3), seed=w + bs)
        left[h // 2, w // 3] = 0
        p = wslib.make_params(vw, bs, dmin, dmax, 1.0, "ssd")
        tl, tr = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
        maps = []
        try:
            for x_runs in (runs, 12):                         # 12 runs per tile: never the halo kernel
                gpu_ctx.set_tuning(x_runs_per_tile=x_runs)
                to = torch.empty((h, w), dtype=torch.float32, device="cuda")
                gpu_ctx.search_device(p, tl, tr, to, None)
                torch.cuda.synchronize()
                maps.append((to.cpu().numpy().astype(np.float64), gpu_ctx.last_launch()["kernel"]))
        finally:
            gpu_ctx.set_tuning()
        assert "halo" in maps[0][1] and "halo" not in maps[1][1], (w, h, bs, maps[0][1], maps[1][1])
        assert np.array_equal(maps[0][0], maps[1][0]), (w, h, bs, dmin, dmax)
        half = (bs - 1) // 2
        fn = oracle.block_left if view == "left" else oracle.block_right
        for y0 in (half, h // 2):
            band = fn(left, right, bs, dmin, dmax, cost="ssd", rows=(y0, y0 + 2), threads=8)
            assert np.array_equal(maps[0][0][y0:y0 + 2], band[y0:y0 + 2]), (w, h, bs, y0)


def test_halo_and_plain_sad_kernels_agree_on_random_large_pairs(wslib, gpu_ctx):
    """Differential: the planner's own choice (the halo-exchange kernel wherever its model prefers it) against the plain
    packed kernel forced by a caller's tile width, whole maps, random large shapes / ranges / views / grey levels and
    both workgroup sizes -- no oracle needed, so the pairs can be large; the plain kernel is the one the oracle tests pin."""
    import torch
    rng = np.random.default_rng(2026)
    used = 0
    for case in range(24):
        w, h = int(rng.integers(1300, 3000)), int(rng.integers(300, 1100))
        bs = int(rng.choice([7, 9]))
        view = wslib.VIEW_LEFT if rng.random() < 0.5 else wslib.VIEW_RIGHT
        dmin = int(rng.choice([0, 0, 13]))
        dmax = dmin + int(rng.choice([256, 400, 512, 512, 512, 777, 1024]))
        levels = int(rng.choice([256, 256, 4]))
        threads = int(rng.choice([0, 0, 256]))
        if levels == 256:
            left, right, _ = make_pair(w, h, min(dmax, w // 3), seed=case)
        else:
            left = (rng.integers(0, levels, size=(h, w, 3)) * (255 // (levels - 1))).astype(np.uint8)
            right = (rng.integers(0, levels, size=(h, w, 3)) * (255 // (levels - 1))).astype(np.uint8)
        p = wslib.make_params(view, bs, dmin, dmax, 1.0, "sad")
        tl, tr = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
        maps = []
        try:
            for x_runs in (0, 12):                            # 12 runs per tile: never the halo kernel
                gpu_ctx.set_tuning(x_runs_per_tile=x_runs, threads=threads)
                to = torch.empty((h, w), dtype=torch.float32, device="cuda")
                gpu_ctx.search_device(p, tl, tr, to, None)
                torch.cuda.synchronize()
                maps.append((to.cpu().numpy(), gpu_ctx.last_launch()["kernel"]))
        finally:
            gpu_ctx.set_tuning()
        assert "halo" not in maps[1][1], maps[1][1]
        used += "halo" in maps[0][1]
        assert np.array_equal(maps[0][0], maps[1][0]), (case, w, h, bs, view, dmin, dmax, levels, threads, maps[0][1])
    assert used >= 6, used


def test_right_view_smooth_factor_on_the_halo_cost_kernel(wslib, gpu_ctx, oracle):
    """smoothFactor 0.9 in the right view (main.cpp:40) wants the winners' costs from the search: at 9 x 9 SAD on a large
    pair that is the cost-writing twin of the halo-exchange kernel (window 8 x 8).  The first rows against the oracle (the
    recurrence runs down from the top), the whole map against the plain kernel's (forced by a tile width of the caller's)."""
    import torch
    w, h, maxd = 2400, 160, 512
    left, right, _ = make_pair(w, h, 300, seed=77)
    right[8:12, 500:900] = right[8, 500]                    # a flat patch: equal costs, zeros for the factor to act on
    right[90:94, 1500:1900] = right[90, 1500]
    p = wslib.make_params(wslib.VIEW_RIGHT, 9, 0, maxd, 0.9, "sad")
    tl, tr = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()

    def search():
        to = torch.empty((h, w), dtype=torch.float32, device="cuda")
        gpu_ctx.search_device(p, tl, tr, to, None)
        torch.cuda.synchronize()
        return to.cpu().numpy().astype(np.float64), gpu_ctx.last_launch()["kernel"]

    got, kernel = search()
    assert "halo" in kernel, kernel
    want = oracle.block_right(left, right, 9, 0, maxd, smooth=0.9, cost="sad", threads=8, rows=(0, 20))
    assert np.array_equal(got[:20], want[:20])
    plain = oracle.block_right(left, right, 9, 0, maxd, smooth=1.0, cost="sad", threads=8, rows=(0, 20))
    assert (want[:20] != plain[:20]).any()                    # the factor did change pixels
    gpu_ctx.set_tuning(x_runs_per_tile=12)                    # a caller's tile width other than 8 / 16 runs: the plain kernel
    try:
        other, kernel = search()
    finally:
        gpu_ctx.set_tuning()
    assert "halo" not in kernel, kernel
    assert np.array_equal(got, other)


@pytest.mark.parametrize("view", ["left", "right"])
@pytest.mark.parametrize("levels", [2, 3])
def test_ties_follow_the_reference_order(wslib, gpu_ctx, oracle, view, levels):
    # few grey levels -> many exactly equal costs: left keeps the largest d, right the smallest
    rng = np.random.default_rng(levels)
    left = (rng.integers(0, levels, size=(60, 260, 3)) * (255 // (levels - 1))).astype(np.uint8)
    right = (rng.integers(0, levels, size=(60, 260, 3)) * (255 // (levels - 1))).astype(np.uint8)
    for cost in ("ssd", "sad"):
        got = run(wslib, gpu_ctx, view, left, right, 5, 0, 90, cost)
        assert np.array_equal(got, ref(oracle, view, left, right, 5, 0, 90, cost))


def test_constant_images_give_pure_tie_break_maps(wslib, gpu_ctx, oracle):
    img = np.full((40, 200, 3), 99, dtype=np.uint8)
    for view in ("left", "right"):
        got = run(wslib, gpu_ctx, view, img, img, 7, 0, 64, "ssd")
        assert np.array_equal(got, ref(oracle, view, img, img, 7, 0, 64, "ssd"))
    z = np.zeros((20, 80, 3), dtype=np.uint8)          # all black: every pixel skipped
    assert (run(wslib, gpu_ctx, "left", z, z, 5, 0, 16, "ssd") == 0).all()


@pytest.mark.parametrize("shape2", [(70, 300), (78, 330), (75, 290)])
def test_unequal_image_sizes(wslib, gpu_ctx, oracle, shape2):
    left, right, _ = make_pair(310, 75, 48, seed=77, right_width=shape2[1], right_height=shape2[0])
    got = run(wslib, gpu_ctx, "left", left, right, 7, 0, 48, "ssd")
    assert np.array_equal(got, ref(oracle, "left", left, right, 7, 0, 48, "ssd"))
    if left.shape[0] >= right.shape[0]:
        got = run(wslib, gpu_ctx, "right", left, right, 7, 1, 48, "sad")
        assert np.array_equal(got, ref(oracle, "right", left, right, 7, 1, 48, "sad"))


@pytest.mark.parametrize("case", [("left", 7, 0, 1), ("left", 7, 0, 3), ("right", 7, 0, 1), ("right", 7, 5, 6),
                                  ("right", 7, 9, 9), ("left", 5, 0, 200), ("right", 5, 0, 150)])
def test_edge_disparity_ranges(wslib, gpu_ctx, oracle, case):
    view, bs, mind, maxd = case
    left, right, _ = make_pair(140, 40, 32, seed=3)
    got = run(wslib, gpu_ctx, view, left, right, bs, mind, maxd, "ssd")
    assert np.array_equal(got, ref(oracle, view, left, right, bs, mind, maxd, "ssd"))


def test_tiny_and_ragged_images(wslib, gpu_ctx, oracle):
    for (h, w) in [(1, 1), (1, 30), (30, 1), (3, 3), (7, 9), (8, 65), (9, 513)]:
        left, right, _ = make_pair(w, h, 8, seed=h * 100 + w)
        for view in ("left", "right"):
            got = run(wslib, gpu_ctx, view, left, right, 3, 0, 8, "sad")
            assert np.array_equal(got, ref(oracle, view, left, right, 3, 0, 8, "sad")), (h, w, view)


@pytest.mark.parametrize("bs,view", [(1, "left"), (19, "left"), (23, "right"), (2, "right"), (21, "left")])
def test_window_sizes_on_the_brute_force_kernel(wslib, gpu_ctx, oracle, bs, view):
    left, right, _ = make_pair(120, 48, 24, seed=bs)
    got = run(wslib, gpu_ctx, view, left, right, bs, 0, 24, "ssd")
    assert np.array_equal(got, ref(oracle, view, left, right, bs, 0, 24, "ssd"))


def test_linear_search(wslib, gpu_ctx, oracle):
    left, right, _ = make_pair(300, 40, 64, seed=8, right_width=310)
    left[5, 7] = 0
    got = wslib.LinearSearch(left, right, context=gpu_ctx).computeDisparityMap(1.0)
    assert np.array_equal(got, oracle.linear(left, right))


@pytest.mark.parametrize("shape", [(300, 40, 280, 40), (280, 40, 300, 40), (600, 30, 600, 36), (50, 20, 50, 20)])
@pytest.mark.parametrize("search_range", [1, 7, 200, 700, 5000])
def test_linear_search_shapes_and_ranges(wslib, gpu_ctx, oracle, shape, search_range):
    """Left narrower / wider / shorter than right, ranges beyond the row and beyond the LDS kernel's
    4096 candidates (brute-force fallback), black left pixels, few grey levels (ties -> smallest d)."""
    w1, h1, w2, h2 = shape
    rng = np.random.default_rng(w1 + search_range)
    left = (rng.integers(0, 4, size=(h1, w1, 3)) * 85).astype(np.uint8)
    right = (rng.integers(0, 4, size=(h2, w2, 3)) * 85).astype(np.uint8)
    left[3:6, 10:30] = 0
    got = wslib.LinearSearch(left, right, context=gpu_ctx, search_range=search_range).computeDisparityMap(1.0)
    assert np.array_equal(got, oracle.linear(left, right, search_range=search_range))
    got = wslib.LinearSearch(left, right, context=gpu_ctx, search_range=search_range).computeDisparityMap(0.5)
    assert np.array_equal(got, oracle.linear(left, right, smooth=0.5, search_range=search_range))


@pytest.mark.parametrize("view", ["left", "right"])
@pytest.mark.parametrize("cost", ["ssd", "sad"])
def test_subpixel_within_tolerance(wslib, gpu_ctx, oracle, view, cost):
    left, right, _ = make_pair(300, 64, 64, seed=9)
    got = run(wslib, gpu_ctx, view, left, right, 9, 0, 64, cost, subpixel=True)
    want = ref(oracle, view, left, right, 9, 0, 64, cost, subpixel=True)
    assert np.abs(got - want).max() <= SUBPIXEL_TOL
    # the integer part is still the bit-exact argmin: the refine adds a fraction in [-0.5, 0.5] to it
    want_int = ref(oracle, view, left, right, 9, 0, 64, cost)
    got_int = run(wslib, gpu_ctx, view, left, right, 9, 0, 64, cost)
    assert np.array_equal(got_int, want_int)
    assert np.abs(got - got_int).max() <= 0.5 + SUBPIXEL_TOL
    assert np.array_equal(np.round(got - (want - want_int)), want_int)
    assert ((got - got_int) != 0).mean() > 0.3


@pytest.mark.parametrize("smooth", [0.9, 0.5, 1.7, 0.0])
@pytest.mark.parametrize("levels", [256, 3])
def test_smooth_factor_right_view_and_linear(wslib, gpu_ctx, oracle, smooth, levels):
    """smoothFactor != 1 (the pipeline default is 0.9, main.cpp:40): right view and LinearSearch
    reproduce the reference's raster-order rule exactly (it only ever reaches d = 0)."""
    if levels == 256:
        left, right, _ = make_pair(420, 56, 48, seed=31)
    else:
        rng = np.random.default_rng(7)
        left = (rng.integers(0, levels, size=(56, 420, 3)) * (255 // (levels - 1))).astype(np.uint8)
        right = (rng.integers(0, levels, size=(56, 420, 3)) * (255 // (levels - 1))).astype(np.uint8)
    left[10:14, 20:60] = 0
    right[5:9, 30:90] = 0
    for bs, mind, cost in ((17, 0, "ssd"), (7, 0, "sad"), (7, 2, "ssd")):
        got = wslib.BlockSearch(left, right, bs, mind, 48, cost=cost, context=gpu_ctx).computeDisparityMapRight(smooth)
        want = oracle.block_right(left, right, bs, mind, 48, smooth=smooth, cost=cost)
        assert np.array_equal(got, want), (bs, mind, cost)
    got = wslib.LinearSearch(left, right, context=gpu_ctx).computeDisparityMap(smooth)
    assert np.array_equal(got, oracle.linear(left, right, smooth=smooth))


@pytest.mark.parametrize("width", [31, 62, 63, 1984, 1985, 2500, 3968, 3969])
def test_smooth_factor_right_view_word_layouts(wslib, gpu_ctx, oracle, width):
    """The right view's 0 <= smoothFactor <= 1 resolver packs 31 columns per word, one word per lane up to
    1984 columns, two up to 3968, the other resolvers beyond (ws_smooth.hip): widths on both sides of every
    limit, few levels so that zeros spread over long runs and whole rows (carries through every word and lane),
    and LinearSearch, which shares the resolver."""
    rng = np.random.default_rng(width)
    h = 24 if width > 1000 else 150          # the tall narrow ones cross several LDS chunks of skewed rows
    left = (rng.integers(0, 2, size=(h, width, 3)) * 255).astype(np.uint8)
    right = left.copy()
    right[rng.random((h, width)) < 0.02] = 255   # mostly equal images: d = 0 wins or ties almost everywhere
    left[3:6, :] = right[3:6, :] = 0             # whole rows of exact ties
    for bs, maxd, smooth, cost in ((5, 6, 0.9, "ssd"), (3, 4, 0.5, "sad")):
        got = wslib.BlockSearch(left, right, bs, 0, maxd, cost=cost, context=gpu_ctx).computeDisparityMapRight(smooth)
        want = oracle.block_right(left, right, bs, 0, maxd, smooth=smooth, cost=cost, threads=8)
        assert np.array_equal(got, want), (width, bs, smooth)
        assert (got == 0).mean() > 0.3
    if width <= 2500:
        got = wslib.LinearSearch(left, right, context=gpu_ctx, search_range=12).computeDisparityMap(0.5)
        assert np.array_equal(got, oracle.linear(left, right, smooth=0.5, search_range=12))


@pytest.mark.parametrize("smooth", [0.9, 0.5, 0.0, 1.3, 4.0, -0.5, -2.0, float("inf")])
@pytest.mark.parametrize("levels", [256, 3, 2])
def test_smooth_factor_left_view(wslib, gpu_ctx, oracle, smooth, levels):
    """Left view: the factor reaches whatever d the upper / left neighbour holds (BlockSearch.cpp:68-73),
    a true raster-order dependency; the device walks the anti-diagonals (ws_smooth.hip).  Factors outside
    [0,1] (a penalty, a sign flip) go through the three-best-candidates pass."""
    if levels == 256:
        left, right, _ = make_pair(300, 48, 40, seed=41)
    else:
        rng = np.random.default_rng(levels)
        left = (rng.integers(0, levels, size=(48, 300, 3)) * (255 // (levels - 1))).astype(np.uint8)
        right = (rng.integers(0, levels, size=(48, 300, 3)) * (255 // (levels - 1))).astype(np.uint8)
    left[10:14, 20:60] = 0
    for bs, cost, maxd in ((7, "ssd", 40), (5, "sad", 24), (1, "ssd", 12), (17, "ssd", 30)):
        got = wslib.BlockSearch(left, right, bs, 0, maxd, cost=cost, context=gpu_ctx).computeDisparityMapLeft(smooth)
        want = oracle.block_left(left, right, bs, 0, maxd, smooth=smooth, cost=cost)
        assert np.array_equal(got, want), (bs, cost)


@pytest.mark.parametrize("case", [(5, 19.0, "ssd", 0, 1.0), (7, 10.0, "ssd", 0, 0.9), (3, 60.0, "sad", 1, 1.0),
                                  (9, 35.0, "ssd", 0, 0.5), (17, 10.0, "ssd", 0, 0.9)])
def test_var_block_right_view(wslib, gpu_ctx, oracle, case):
    """varBlock (BlockSearch.cpp:125-145): windows grow by 4 while their centred norm is below thres;
    also together with smoothFactor, and the "max block size" the reference prints (:177)."""
    bs, thres, cost, mind, smooth = case
    left, right, _ = make_pair(170, 52, 24, seed=3)
    right[5:30, 20:70] = (right[5:30, 20:70] // 32) * 32      # weak texture: windows must grow
    right[10:20, 90:110] = 128                                 # none at all
    left[8:25, 30:80] = (left[8:25, 30:80] // 64) * 64
    right[0:4, 0:9] = 0
    want, want_mb = oracle.block_right(left, right, bs, mind, 24, smooth=smooth, var_block=True, thres=thres,
                                       cost=cost, return_max_block=True)
    got = wslib.BlockSearch(left, right, bs, mind, 24, cost=cost, context=gpu_ctx).computeDisparityMapRight(smooth, True, thres)
    assert np.array_equal(got, want)
    assert gpu_ctx.last_max_block(bs) == want_mb
    assert want_mb > bs or bs == 17


def test_reference_pipeline_call_on_teddy_sized_pair(wslib, gpu_ctx, oracle):
    """main.cpp:40: computeDisparityMapRight(17, 0, 200, 0.9) at Teddy-H size (900 x 750)."""
    left, right, _ = make_pair(900, 750, 200, seed=13)
    got = wslib.BlockSearch(left, right, 17, 0, 200, context=gpu_ctx).computeDisparityMapRight(0.9)
    assert "march" in gpu_ctx.last_launch()["kernel"]
    band = oracle.block_right(left, right, 17, 0, 200, smooth=0.9, rows=(0, 12), threads=8)
    assert np.array_equal(got[:12], band[:12])       # raster dependency: the oracle must start at row 0


def test_warp_back_to_the_original_frame(wslib, gpu_ctx, oracle):
    """The warpPerspective(.., H_.inv(), INTER_NEAREST) of rectification.cpp:70-75 on the device."""
    rng = np.random.default_rng(2)
    disp = rng.integers(0, 200, size=(90, 140)).astype(np.float64)
    ident = np.eye(3)
    assert np.array_equal(gpu_ctx.warp_nearest(disp, ident, (90, 140)), disp)      # rectified pairs: a copy
    for m in (np.array([[1.02, 0.01, -3.0], [-0.015, 0.98, 2.5], [1e-5, -2e-5, 1.0]]),
              np.array([[0.5, 0.0, 10.0], [0.0, 0.5, -4.0], [0.0, 0.0, 1.0]]),
              np.array([[1.0, 0.2, 0.0], [0.0, 1.0, 0.0], [0.0, 1e-3, 1.0]])):
        got = gpu_ctx.warp_nearest(disp, m, (100, 150))
        want = oracle.warp_nearest(disp, m, (100, 150))
        assert np.array_equal(got, want)
        assert (got != 0).any()


def test_reconstruction_consumers(wslib, gpu_ctx, oracle, tmp_path):
    """The map's consumers (reconstruction.cpp:5-43, :152-196) as main.cpp:50-64 chains them:
    8-bit disparity -> removeDisparityOutliers(500, 1.5, 0.8) -> depth -> vertices -> OFF mesh."""
    left, right, gt = make_pair(300, 220, 64, seed=17)
    disp8 = np.clip(gt, 0, 255).astype(np.float32)            # what readGrayscaleImageAsDisparityMap yields
    disp8[50:60, 70:90] = 0
    disp8[100, 100] = 250
    for k in (500, 31, 4):
        got = gpu_ctx.remove_disparity_outliers(disp8, k, 1.5, 0.8)
        assert np.array_equal(got, oracle.remove_disparity_outliers(disp8, k, 1.5, 0.8)), k
    filt = gpu_ctx.remove_disparity_outliers(disp8, 500, 1.5, 0.8)
    depth = gpu_ctx.convert_disparity_to_depth(filt, 3000.0, 1.0)
    assert np.array_equal(depth, oracle.convert_disparity_to_depth(filt, 3000.0, 1.0))
    zero = gpu_ctx.convert_disparity_to_depth(disp8, 3000.0, 1.0)
    assert np.isneginf(zero[55, 75])
    K = np.array([[3000, 0, 150], [0, 3000, 110], [0, 0, 1]], dtype=np.float32)
    pos, col = gpu_ctx.back_project(zero, K, right)
    wpos, wcol = oracle.back_project(zero, K, right)
    assert np.array_equal(pos, wpos) and np.array_equal(col, wcol)
    path = str(tmp_path / "mesh.off")
    wslib.write_mesh_off(path, pos[40:70, 60:100], col[40:70, 60:100], 1.0)
    assert open(path).read() == oracle.mesh_off_text(pos[40:70, 60:100], col[40:70, 60:100], 1.0)


@pytest.mark.parametrize("seed", range(int(os.environ.get("WS_FUZZ_CASES", "48"))))
def test_randomised_differential(wslib, gpu_ctx, oracle, seed):
    """Random shapes / windows / ranges / views against the oracle: tile edges, D not a multiple
    of the chunk, D larger than the image, unequal sizes, black patches, few grey levels."""
    rng = np.random.default_rng(1000 + int(os.environ.get("WS_FUZZ_BASE", "0")) + seed)   # (WS_FUZZ_BASE: fresh cases)
    w1, h1 = int(rng.integers(20, 420)), int(rng.integers(12, 90))
    same = rng.random() < 0.5
    w2 = w1 if same else max(8, w1 + int(rng.integers(-40, 41)))
    h2 = h1 if same else max(6, h1 + int(rng.integers(-6, 1)))      # h1 >= h2 keeps the right view legal
    levels = int(rng.choice([256, 256, 4, 2]))
    if levels == 256:
        left = rng.integers(0, 256, size=(h1, w1, 3), dtype=np.uint8)
        right = rng.integers(0, 256, size=(h2, w2, 3), dtype=np.uint8)
        d0 = int(rng.integers(0, 30))
        n = min(w1 - d0, w2)
        if n > 0:
            right[:min(h1, h2), :n] = left[:min(h1, h2), d0:d0 + n]   # something to find
    else:
        left = (rng.integers(0, levels, size=(h1, w1, 3)) * (255 // (levels - 1))).astype(np.uint8)
        right = (rng.integers(0, levels, size=(h2, w2, 3)) * (255 // (levels - 1))).astype(np.uint8)
    y, x = int(rng.integers(0, h1)), int(rng.integers(0, w1))
    left[y:y + 3, x:x + 9] = 0
    y, x = int(rng.integers(0, h2)), int(rng.integers(0, w2))
    right[y:y + 2, x:x + 5] = 0
    view = "left" if rng.random() < 0.5 else "right"
    bs = int(rng.choice([1, 3, 5, 7, 9, 11, 13, 15, 17, 19] if view == "left" else [2, 3, 5, 6, 7, 9, 12, 13, 17, 21]))
    maxd = int(rng.choice([1, 2, 7, 8, 9, 31, 64, 65, 100, 200, 300, 513]))
    mind = int(rng.choice([0, 0, 1, 5])) if view == "right" else 0
    cost = "ssd" if rng.random() < 0.5 else "sad"
    smooth = float(rng.choice([1.0, 1.0, 0.9, 0.3] if view == "right" else [1.0, 1.0, 0.9, 1.6, -0.4]))
    b = wslib.BlockSearch(left, right, bs, mind, maxd, cost=cost, context=gpu_ctx)
    if view == "left":
        got = b.computeDisparityMapLeft(smooth)
        want = oracle.block_left(left, right, bs, mind, maxd, smooth=smooth, cost=cost, threads=8)
    else:
        var_block = bool(rng.random() < 0.25)
        thres = float(rng.choice([10.0, 40.0, 150.0]))
        got = b.computeDisparityMapRight(smooth, var_block, thres)
        want = oracle.block_right(left, right, bs, mind, maxd, smooth=smooth, var_block=var_block, thres=thres,
                                  cost=cost, threads=8, return_max_block=var_block)
        if var_block:
            want, want_mb = want
            assert gpu_ctx.last_max_block(bs) == want_mb
    assert np.array_equal(got, want), (view, bs, mind, maxd, cost, smooth, left.shape, right.shape, levels)


def test_strided_buffers_and_extreme_arguments(wslib, gpu_ctx, oracle):
    """Row strides larger than the row (cv::Mat ROIs), padded output, absurd disparity ranges."""
    import ctypes
    import torch
    big_l, big_r, _ = make_pair(260, 60, 40, seed=71)
    left, right = big_l[5:55, 10:230], big_r[5:55, 10:230]          # non-contiguous views: stride 780 > 3 * 220
    assert not left.flags["C_CONTIGUOUS"]
    want = oracle.block_left(np.ascontiguousarray(left), np.ascontiguousarray(right), 7, 0, 40)
    p = wslib.make_params(wslib.VIEW_LEFT, 7, 0, 40)
    lib = wslib.load_library()
    Li = wslib._Image(left.ctypes.data, 220, 50, left.strides[0])
    Ri = wslib._Image(right.ctypes.data, 220, 50, right.strides[0])
    out = np.full((50, 256), -5.0, dtype=np.float64)                 # out_stride 256 > width 220
    rc = lib.ws_search_host(gpu_ctx._h, ctypes.byref(p), ctypes.byref(Li), ctypes.byref(Ri), out.ctypes.data, 256, 1)
    assert rc == 0 and np.array_equal(out[:, :220], want) and (out[:, 220:] == -5.0).all()
    # the batched entry point with the same strided buffers, three pairs in flight, then a wide ROI
    # (stride > 2 x row: copied row by row) and a row length that is not a multiple of 4 bytes
    outs = [np.full((50, 256), -5.0, dtype=np.float64) for _ in range(3)]
    for o in outs:
        assert lib.ws_enqueue_host(gpu_ctx._h, ctypes.byref(p), ctypes.byref(Li), ctypes.byref(Ri), o.ctypes.data, 256, 1) == 0
    assert lib.ws_wait(gpu_ctx._h) == 0
    for o in outs:
        assert np.array_equal(o[:, :220], want) and (o[:, 220:] == -5.0).all()
    nl, nr = big_l[5:55, 10:81], big_r[5:55, 10:81]                 # 71 columns: 213 bytes per row, stride 780
    nwant = oracle.block_left(np.ascontiguousarray(nl), np.ascontiguousarray(nr), 7, 0, 40)
    NLi = wslib._Image(nl.ctypes.data, 71, 50, nl.strides[0])
    NRi = wslib._Image(nr.ctypes.data, 71, 50, nr.strides[0])
    o1, o2 = np.zeros((50, 71), dtype=np.float32), np.zeros((50, 71), dtype=np.float32)
    assert lib.ws_search_host(gpu_ctx._h, ctypes.byref(p), ctypes.byref(NLi), ctypes.byref(NRi), o1.ctypes.data, 71, 0) == 0
    assert lib.ws_enqueue_host(gpu_ctx._h, ctypes.byref(p), ctypes.byref(NLi), ctypes.byref(NRi), o2.ctypes.data, 71, 0) == 0
    assert lib.ws_wait(gpu_ctx._h) == 0
    assert np.array_equal(o1.astype(np.float64), nwant) and np.array_equal(o2.astype(np.float64), nwant)
    # a big cut-out of a much wider image (span > 32 MB, > 2 x dense): the row-by-row copy path
    rng = np.random.default_rng(5)
    huge_l = rng.integers(1, 255, size=(1900, 6100, 3), dtype=np.uint8)
    huge_r = np.roll(huge_l, -3, axis=1)
    cl, cr = huge_l[:, 100:501], huge_r[:, 100:501]
    cwant = oracle.block_left(np.ascontiguousarray(cl), np.ascontiguousarray(cr), 3, 0, 8, threads=8)
    CLi = wslib._Image(cl.ctypes.data, 401, 1900, cl.strides[0])
    CRi = wslib._Image(cr.ctypes.data, 401, 1900, cr.strides[0])
    p3 = wslib.make_params(wslib.VIEW_LEFT, 3, 0, 8)
    o3 = np.zeros((1900, 401), dtype=np.float32)
    assert lib.ws_search_host(gpu_ctx._h, ctypes.byref(p3), ctypes.byref(CLi), ctypes.byref(CRi), o3.ctypes.data, 401, 0) == 0
    assert np.array_equal(o3.astype(np.float64), cwant)
    # the same through the device entry point with strided device tensors
    tl, tr = torch.from_numpy(big_l).cuda()[5:55, 10:230], torch.from_numpy(big_r).cuda()[5:55, 10:230]
    to = torch.full((50, 256), -5.0, dtype=torch.float32, device="cuda")
    gpu_ctx.search_device(p, tl, tr, to[:, :220], None)
    torch.cuda.synchronize()
    assert np.array_equal(to.cpu().numpy()[:, :220].astype(np.float64), want) and bool((to[:, 220:] == -5.0).all())
    # a disparity range far wider than the image, and the largest window the ABI accepts
    l2, r2 = np.ascontiguousarray(left), np.ascontiguousarray(right)
    for view in ("left", "right"):
        got = run(wslib, gpu_ctx, view, l2, r2, 5, 0, 100000, "sad")
        assert np.array_equal(got, ref(oracle, view, l2, r2, 5, 0, 100000, "sad"))
    got = run(wslib, gpu_ctx, "left", l2, r2, 63, 0, 30, "ssd")
    assert np.array_equal(got, ref(oracle, "left", l2, r2, 63, 0, 30, "ssd"))
    with pytest.raises(wslib.WsError) as e:
        run(wslib, gpu_ctx, "left", l2, r2, 65, 0, 30, "ssd")
    assert e.value.code == -1


def test_very_wide_and_very_tall_images(wslib, gpu_ctx, oracle):
    """More than 4096 columns / rows: many tiles per row, many strips per column, 16-bit-looking sizes."""
    left, right, _ = make_pair(5000, 96, 90, seed=81)
    for view, bs, cost in (("left", 5, "ssd"), ("right", 5, "sad"), ("right", 9, "ssd")):
        got = run(wslib, gpu_ctx, view, left, right, bs, 0, 96, cost)
        assert np.array_equal(got, ref(oracle, view, left, right, bs, 0, 96, cost)), (view, bs, cost)
    left, right, _ = make_pair(96, 4500, 40, seed=82)
    for view, bs, cost in (("left", 7, "sad"), ("right", 7, "ssd")):
        got = run(wslib, gpu_ctx, view, left, right, bs, 0, 40, cost)
        assert np.array_equal(got, ref(oracle, view, left, right, bs, 0, 40, cost)), (view, bs, cost)
    # the right view's smoothFactor passes on a map wider than 64 words of 64 columns
    got = wslib.BlockSearch(left, right, 7, 0, 40, context=gpu_ctx).computeDisparityMapRight(0.9)
    assert np.array_equal(got, oracle.block_right(left, right, 7, 0, 40, smooth=0.9))
    wl, wr, _ = make_pair(4300, 40, 30, seed=83)
    got = wslib.BlockSearch(wl, wr, 5, 0, 30, context=gpu_ctx).computeDisparityMapRight(0.9)
    assert np.array_equal(got, oracle.block_right(wl, wr, 5, 0, 30, smooth=0.9))


def test_two_contexts_from_two_threads(wslib, oracle):
    """One context per host thread on the same device: no shared state."""
    import threading
    left, right, _ = make_pair(400, 120, 48, seed=84)
    want = {"left": oracle.block_left(left, right, 7, 0, 48, threads=4),
            "right": oracle.block_right(left, right, 7, 0, 48, threads=4)}
    errors = []

    def work(view):
        try:
            ctx = wslib.WindowSearch(0)
            b = wslib.BlockSearch(left, right, 7, 0, 48, context=ctx)
            for _ in range(20):
                got = b.computeDisparityMapLeft(1.0) if view == "left" else b.computeDisparityMapRight(1.0)
                if not np.array_equal(got, want[view]):
                    errors.append(view)
                    return
        except Exception as e:          # noqa: BLE001 - reported through the list
            errors.append(repr(e))

    ts = [threading.Thread(target=work, args=(v,)) for v in ("left", "right", "left", "right")]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("shape", [(1, 1), (2, 2), (3, 1), (1, 4), (5, 5), (9, 3), (16, 16), (17, 2)])
def test_tiny_images_every_path(wslib, gpu_ctx, oracle, shape):
    """Images smaller than the window: no interior, everything is border / fallback, in every path."""
    w, h = shape
    rng = np.random.default_rng(w * 31 + h)
    left = rng.integers(0, 3, size=(h, w, 3)).astype(np.uint8) * 100
    right = rng.integers(0, 3, size=(h, w, 3)).astype(np.uint8) * 100
    for bs in (1, 3, 7):
        for s in (1.0, 0.9, 1.5):
            b = wslib.BlockSearch(left, right, bs, 0, 6, context=gpu_ctx)
            assert np.array_equal(b.computeDisparityMapLeft(s), oracle.block_left(left, right, bs, 0, 6, smooth=s)), (bs, s)
            assert np.array_equal(b.computeDisparityMapRight(s), oracle.block_right(left, right, bs, 0, 6, smooth=s)), (bs, s)
        got = wslib.BlockSearch(left, right, bs, 0, 6, context=gpu_ctx).computeDisparityMapRight(0.9, True, 150.0)
        assert np.array_equal(got, oracle.block_right(left, right, bs, 0, 6, smooth=0.9, var_block=True, thres=150.0)), bs
    for s in (1.0, 0.5):
        got = wslib.LinearSearch(left, right, context=gpu_ctx, search_range=5).computeDisparityMap(s)
        assert np.array_equal(got, oracle.linear(left, right, smooth=s, search_range=5))


def test_errors_are_reported_not_computed(wslib, gpu_ctx):
    left, right, _ = make_pair(100, 40, 16, seed=1)
    with pytest.raises(wslib.WsError) as e:
        run(wslib, gpu_ctx, "left", left, right, 6, 0, 16, "ssd")
    assert e.value.code == -2                       # even blockSize: the reference throws
    with pytest.raises(wslib.WsError) as e:
        wslib.BlockSearch(left, right, 7, 0, 16, context=gpu_ctx).computeDisparityMapLeft(float("nan"))
    assert e.value.code == -1
    with pytest.raises(wslib.WsError) as e:
        wslib.BlockSearch(left, right, 7, 0, 16, subpixel=True, context=gpu_ctx).computeDisparityMapRight(0.9)
    assert e.value.code == -3                       # the sub-pixel extension needs smoothFactor 1
    with pytest.raises(wslib.WsError) as e:
        run(wslib, gpu_ctx, "right", left[:30], right, 7, 0, 16, "ssd")
    assert e.value.code == -2


def test_device_resident_path_and_f32_output(wslib, gpu_ctx, oracle):
    import torch
    left, right, _ = make_pair(500, 120, 128, seed=4)
    tl, tr = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    out = torch.full((120, 500), -7.0, dtype=torch.float32, device="cuda")
    p = wslib.make_params(wslib.VIEW_LEFT, 7, 0, 128, 1.0, "ssd")
    gpu_ctx.search_device(p, tl, tr, out, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want = ref(oracle, "left", left, right, 7, 0, 128, "ssd")
    assert np.array_equal(out.cpu().numpy().astype(np.float64), want)
    # deterministic: a second run on the library's own stream gives the same bits
    out2 = torch.empty_like(out)
    gpu_ctx.search_device(p, tl, tr, out2, None)
    torch.cuda.synchronize()
    assert torch.equal(out, out2)


def test_device_entry_point_can_be_captured_into_a_hip_graph(wslib, gpu_ctx):
    """After one warm-up call (scratch buffers exist) ws_search_device only enqueues kernels on the
    caller's stream: a stream capture records it and the replay gives the same bits."""
    import torch
    left, right, _ = make_pair(500, 200, 64, seed=51)
    tl, tr = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    for view, s in ((wslib.VIEW_LEFT, 1.0), (wslib.VIEW_RIGHT, 0.9)):
        p = wslib.make_params(view, 7, 0, 64, s, "ssd")
        want = torch.empty((200, 500), dtype=torch.float32, device="cuda")
        out = torch.empty_like(want)
        gpu_ctx.search_device(p, tl, tr, want, None)
        torch.cuda.synchronize()
        st = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(st):
            gpu_ctx.search_device(p, tl, tr, out, st.cuda_stream)
            st.synchronize()
            with torch.cuda.graph(g, stream=st):
                gpu_ctx.search_device(p, tl, tr, out, st.cuda_stream)
        out.zero_()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want), view


def test_batched_host_path_equals_single_calls(wslib, gpu_ctx):
    pairs = [make_pair(201 + 17 * i, 50 + i, 32, seed=40 + i)[:2] for i in range(7)]   # odd widths, 7 > 2 slots
    p = wslib.make_params(wslib.VIEW_LEFT, 5, 0, 32, 1.0, "sad")
    many = gpu_ctx.search_many(p, pairs, dtype=np.float32)
    for (l, r), m in zip(pairs, many):
        assert np.array_equal(m, gpu_ctx.search(p, l, r, dtype=np.float32))
    many64 = gpu_ctx.search_many(p, pairs[:2], dtype=np.float64)   # job slots are re-used
    assert np.array_equal(many64[0], many[0].astype(np.float64))


def test_tuning_knobs_do_not_change_results(wslib, oracle):
    left, right, _ = make_pair(700, 90, 64, seed=12)
    want = ref(oracle, "left", left, right, 7, 0, 64, "sad")
    with wslib.WindowSearch(0) as ctx:
        for nxr, rows, threads in [(0, 0, 0), (8, 7, 0), (16, 33, 256), (40, 90, 512), (5, 1, 0)]:
            ctx.set_tuning(nxr, rows, threads)
            got = run(wslib, ctx, "left", left, right, 7, 0, 64, "sad")
            assert np.array_equal(got, want), (nxr, rows, threads)


# ---- BASELINE.json sizes: row-band comparison + size-independent properties -------------------
@pytest.mark.parametrize("cfg", [("config2", 1500, 1000, 7, "ssd", 256, 2), ("config3", 2964, 1988, 9, "sad", 512, 3),
                                 ("config5", 3840, 2160, 9, "ssd", 1024, 5)])
def test_full_size_configs(wslib, gpu_ctx, oracle, cfg):
    name, w, h, bs, cost, maxd, seed = cfg
    left, right, gt = make_pair(w, h, maxd, seed)
    got = run(wslib, gpu_ctx, "left", left, right, bs, 0, maxd, cost)
    assert "march" in gpu_ctx.last_launch()["kernel"]
    half = (bs - 1) // 2
    # (1) bit-exact against the oracle on row bands the oracle finishes in seconds
    nb = 3 if name == "config2" else 1
    for y0 in [half, h // 2, h - half - nb][: (3 if name == "config2" else 2)]:
        band = ref(oracle, "left", left, right, bs, 0, maxd, cost, rows=(y0, y0 + nb))
        assert np.array_equal(got[y0:y0 + nb], band[y0:y0 + nb]), (name, y0)
    # (2) properties that hold at any size
    assert (got[:half] == 0).all() and (got[h - half:] == 0).all()
    assert (got[:, :half] == 0).all() and (got[:, w - half:] == 0).all()
    assert (got[half:h - half, half] == half).all()             # x = half: no candidate -> stores x
    inner = got[half:h - half, half:w - half]
    assert inner.min() >= 1 and (inner == np.round(inner)).all()
    # the planted disparity field is recovered on most textured pixels away from band edges
    hit = (got[half:h - half, maxd:w - half] == gt[half:h - half, maxd:w - half]).mean()
    assert hit > 0.6, hit   # occluded and band-edge pixels miss; ~0.7-0.9 observed


@pytest.mark.parametrize("view,cost", [("left", "ssd"), ("left", "sad"), ("right", "ssd")])
def test_wide_disparity_range_runs_in_passes(wslib, gpu_ctx, oracle, view, cost):
    """D beyond what one tile holds: several d-group passes meeting in a key plane."""
    left, right, _ = make_pair(1400, 24, 1100, seed=55)
    p = wslib.make_params(wslib.VIEW_LEFT if view == "left" else wslib.VIEW_RIGHT, 5, 0, 1100, 1.0, cost)
    assert wslib.plan(p, left.shape, right.shape)["passes"] >= 2
    got = run(wslib, gpu_ctx, view, left, right, 5, 0, 1100, cost)
    assert np.array_equal(got, ref(oracle, view, left, right, 5, 0, 1100, cost))


def test_shifted_copy_known_answer_at_full_size(wslib, gpu_ctx):
    rng = np.random.default_rng(11)
    w, h, d0 = 1500, 1000, 137
    left = rng.integers(1, 256, size=(h, w, 3), dtype=np.uint8)
    right = rng.integers(1, 256, size=(h, w, 3), dtype=np.uint8)
    right[:, : w - d0] = left[:, d0:]
    for view, cost in (("left", "ssd"), ("right", "sad")):
        got = run(wslib, gpu_ctx, view, left, right, 7, 0, 256, cost)
        if view == "left":
            assert (got[3:h - 3, 3 + d0: w - 3] == d0).all()
        else:
            assert (got[3:h - 4, 3: w - d0 - 4] == d0).all()


def test_cxx_facade_runs_the_reference_call_surface(wslib, oracle, tmp_path):
    exe = str(tmp_path / "facade_driver")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", ROOT, "-o", exe,
                           os.path.join(ROOT, "tests", "cxx", "facade_driver.cpp"),
                           "-L", os.path.join(ROOT, "stereo_reconstruction_amd"), "-lws_stereo",
                           "-Wl,-rpath," + os.path.join(ROOT, "stereo_reconstruction_amd")])
    left, right, _ = make_pair(260, 70, 48, seed=21)
    lp, rp, op = str(tmp_path / "l.raw"), str(tmp_path / "r.raw"), str(tmp_path / "o.raw")
    left.tofile(lp)
    right.tofile(rp)
    for mode, want in (("left", oracle.block_left(left, right, 7, 0, 48)),
                       ("right", oracle.block_right(left, right, 7, 0, 48)),
                       ("rectifier", oracle.block_left(left, right, 7, 0, 48)),
                       ("linear", oracle.linear(left, right))):
        subprocess.check_call([exe, lp, "260", "70", rp, "260", "70", mode, "7", "0", "48", "ssd", op])
        got = np.fromfile(op, dtype=np.float64).reshape(70, 260)
        assert np.array_equal(got, want), mode
    # what the reference reports as cv::Exception arrives as wsamd::Error (exit code 10 - code)
    rc = subprocess.call([exe, lp, "260", "70", rp, "260", "70", "left", "6", "0", "48", "ssd", op])
    assert rc == 12


def test_example_pipeline_equals_the_python_chain(wslib, gpu_ctx, oracle, tmp_path):
    """examples/pipeline_main.cpp (the reference's main.cpp:13-66 on a rectified pair) end to end."""
    exe = str(tmp_path / "pipeline_main")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", ROOT, "-o", exe, os.path.join(ROOT, "examples", "pipeline_main.cpp"),
                           "-L", os.path.join(ROOT, "stereo_reconstruction_amd"), "-lws_stereo",
                           "-Wl,-rpath," + os.path.join(ROOT, "stereo_reconstruction_amd")])
    left, right, _ = make_pair(240, 140, 60, seed=23)
    wslib.write_ppm(str(tmp_path / "im0.ppm"), left)
    wslib.write_ppm(str(tmp_path / "im1.ppm"), right)
    prefix = str(tmp_path / "out")
    subprocess.check_call([exe, str(tmp_path / "im0.ppm"), str(tmp_path / "im1.ppm"),
                           os.path.join(ROOT, "tests", "golden", "teddy_calib.txt"), prefix])
    disp = oracle.block_right(left, right, 17, 0, 200, smooth=0.9)
    d8 = np.clip(np.rint(disp), 0, 255).astype(np.float32)
    assert np.array_equal(wslib.read_pfm(prefix + "_disparity.pfm"), d8)
    filt = oracle.remove_disparity_outliers(d8, 500, 1.5, 0.8)
    depth = oracle.convert_disparity_to_depth(filt, 3000.0, 1.0)
    K = wslib.read_calib(os.path.join(ROOT, "tests", "golden", "teddy_calib.txt"))["cam1"]
    pos, col = oracle.back_project(depth, K, right)
    head = open(prefix + "_mesh.off").read().split("\n")[:2 + 240 * 3]
    want = oracle.mesh_off_text(pos[:4], col[:4], 1.0).split("\n")
    assert head[0] == "COFF" and head[1].split()[0] == str(240 * 140)
    assert head[2:2 + 240 * 3] == want[2:2 + 240 * 3]        # the first three rows of vertices


def test_teddy_quarter_bad2_same_as_cpu(wslib, gpu_ctx, oracle):
    """bad-2.0 (evaldisp, utils.cpp:123-168) of the device map equals that of the CPU map."""
    g = load_golden("teddy_quarter")
    got = run(wslib, gpu_ctx, "left", g["left"], g["right"], 5, 0, 64, "sad")
    want = oracle.block_left(g["left"], g["right"], 5, 0, 64, cost="sad", threads=8)
    assert np.array_equal(got, want)
    assert wslib.evaldisp(got, g["gt"], g["mask"], 2.0, 64.0) == oracle.evaldisp(want, g["gt"], g["mask"], 2.0, 64.0)


@pytest.mark.parametrize("case", [(3, 1, 1.0), (3, 80, 1.0), (4, 37, 0.9), (7, 1, 1.0), (7, 1, 0.9), (17, 1, 0.5), (2, 5, 1.0)])
def test_right_view_with_a_taller_right_image(wslib, gpu_ctx, oracle, case):
    """h2 > h1 is legal in the right view while no window needs a left-image row >= h1 (BlockSearch.cpp:151-154):
    always for blockSize <= 4, and for h2 == h1 + 1.  Rows >= h1 of the map stay 0 (:94,:100); the border-ring
    kernel must not stage windows there (plane B has no such rows)."""
    bs, extra, smooth = case
    h1 = 40
    left, right, _ = make_pair(200, h1, 24, seed=60 + bs, right_width=190, right_height=h1 + extra)
    right[h1 - 2:h1 + 1, 5:9] = 0
    for cost in ("ssd", "sad"):
        got = wslib.BlockSearch(left, right, bs, 0, 24, cost=cost, context=gpu_ctx).computeDisparityMapRight(smooth)
        want = oracle.block_right(left, right, bs, 0, 24, smooth=smooth, cost=cost)
        assert got.shape == (h1 + extra, 190)
        assert np.array_equal(got, want), (bs, extra, cost)
        assert (got[h1:] == 0).all()
    with pytest.raises(wslib.WsError) as e:      # a grown window may need rows >= h1: defined as an error
        wslib.BlockSearch(left, right, bs, 0, 24, context=gpu_ctx).computeDisparityMapRight(1.0, True, 30.0)
    assert e.value.code == -2
    if bs >= 7:                                   # two extra rows: the reference throws for the plain call too
        _, r2, _ = make_pair(200, h1, 24, seed=61, right_width=190, right_height=h1 + 2)
        with pytest.raises(wslib.WsError) as e:
            wslib.BlockSearch(left, r2, bs, 0, 24, context=gpu_ctx).computeDisparityMapRight(1.0)
        assert e.value.code == -2


@pytest.mark.parametrize("smooth", [float("inf"), float("-inf"), 1e200, -1e200, 0.0])
def test_non_finite_products_refuse_the_first_candidate(wslib, gpu_ctx, oracle, smooth):
    """Right view / LinearSearch with a factor whose product with d = 0's distance is inf or NaN (0 * inf):
    `dist < min` (BlockSearch.cpp:168, LinearSearch.cpp:46) is then false, d = 0 is REFUSED and the best d >= 1
    -- or, without one, minimumCorrespondX = 0, i.e. -x -- is stored.  Images with exact matches at d = 0."""
    rng = np.random.default_rng(12)
    left = (rng.integers(0, 3, size=(36, 150, 3)) * 120).astype(np.uint8)
    right = left[:, :140].copy()                       # d = 0 matches exactly: c0 = 0 everywhere
    right[8:12, 30:60] = (rng.integers(0, 3, size=(4, 30, 3)) * 120).astype(np.uint8)
    right[20, 70:75] = 0
    for bs, cost, maxd in ((7, "ssd", 20), (5, "sad", 1), (3, "ssd", 12), (19, "ssd", 8)):
        got = wslib.BlockSearch(left, right, bs, 0, maxd, cost=cost, context=gpu_ctx).computeDisparityMapRight(smooth)
        want = oracle.block_right(left, right, bs, 0, maxd, smooth=smooth, cost=cost)
        assert np.array_equal(got, want), (bs, cost, maxd)
    for rng_ in (200, 1, 3):
        got = wslib.LinearSearch(left, right, context=gpu_ctx, search_range=rng_).computeDisparityMap(smooth)
        assert np.array_equal(got, oracle.linear(left, right, smooth=smooth, search_range=rng_)), rng_


def test_range_far_wider_than_the_image_stays_on_the_marching_kernel(wslib, gpu_ctx, oracle):
    """maxDisparity = 10 x width: the candidate range is clamped to what the geometry allows, so the search
    needs one d-group pass of up to 512 candidates (8 per thread) or two of up to 256 (4 per thread: ws_march.hip,
    march_nd) instead of six or twelve, and (SAD) no more tie-tag bits than the image is wide."""
    left, right, _ = make_pair(300, 40, 64, seed=91)
    for view, vid in (("left", wslib.VIEW_LEFT), ("right", wslib.VIEW_RIGHT)):
        for cost in ("ssd", "sad"):
            p = wslib.make_params(vid, 7, 0, 3000, 1.0, cost)
            info = wslib.plan(p, left.shape, right.shape)
            assert info["marching"] == 1 and info["passes"] == (1 if info["d_per_thread"] == 8 else 2), (view, cost, info)
            got = run(wslib, gpu_ctx, view, left, right, 7, 0, 3000, cost)
            assert "march" in gpu_ctx.last_launch()["kernel"]
            assert np.array_equal(got, ref(oracle, view, left, right, 7, 0, 3000, cost)), (view, cost)


def test_calls_on_two_streams_share_the_scratch_safely(wslib, gpu_ctx, oracle):
    """The context's scratch planes are shared by all calls: a call on another stream waits (on the device)
    for the previous one.  Alternate two streams and two different pairs without any host sync."""
    import torch
    pairs = [make_pair(640, 200, 96, seed=s)[:2] for s in (101, 102)]
    wants = [oracle.block_left(l, r, 7, 0, 96, threads=8) for l, r in pairs]
    dev = [(torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()) for l, r in pairs]
    p = wslib.make_params(wslib.VIEW_LEFT, 7, 0, 96, 1.0, "ssd")
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [torch.empty((200, 640), dtype=torch.float32, device="cuda") for _ in range(8)]
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        gpu_ctx.search_device(p, dev[i & 1][0], dev[i & 1][1], o, streams[i & 1].cuda_stream)
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        assert np.array_equal(o.cpu().numpy().astype(np.float64), wants[i & 1]), i


def test_var_block_texture_test_is_float32_like_opencv(wslib, gpu_ctx, oracle):
    """cv::subtract(Mat_u8, Scalar) runs in float32.  Where the float mean is exactly k + .5 and the double
    mean is not, the float subtraction ties (here: norm 0 < thres, the window grows once more to 263 x 263)
    and a double one does not (norm 319, the window stays 261 x 261): different windows, different winners.
    The device must follow the oracle (float32)."""
    from test_oracle_construction import float32_tie_scene
    left, right = float32_tie_scene()
    got = wslib.BlockSearch(left, right, 3, 0, 32, context=gpu_ctx).computeDisparityMapRight(1.0, True, 10.0)
    got_mb = gpu_ctx.last_max_block(3)
    want, want_mb = oracle.block_right(left, right, 3, 0, 32, var_block=True, thres=10.0, rows=(129, 132), threads=8,
                                       return_max_block=True)
    assert np.array_equal(got[129:132], want[129:132])
    assert got_mb >= want_mb >= 267


@pytest.mark.parametrize("view", ["left", "right"])
def test_host_call_in_bands_equals_the_plain_call(wslib, oracle, view):
    """ws_search_host cuts big calls into row bands so that copies and searches overlap; a row's result only
    depends on the rows under its window (BlockSearch.cpp:46-66, :120-158), so the maps must not change:
    2..8 bands, odd sizes, both output types, sub-pixel, unequal widths -- and against the oracle."""
    left, right, _ = make_pair(611, 263, 48, seed=95, right_width=590)
    left[100:103, 50:90] = 0
    right[200, 300:310] = 0
    vid = wslib.VIEW_LEFT if view == "left" else wslib.VIEW_RIGHT
    with wslib.WindowSearch(0) as ctx:
        for bs, cost, sub in ((7, "ssd", False), (9, "sad", True), (17, "ssd", False), (3, "sad", False)):
            p = wslib.make_params(vid, bs, 0, 48, 1.0, cost, subpixel=sub)
            ctx.set_host_bands(0)
            plain64 = ctx.search(p, left, right, dtype=np.float64)
            plain32 = ctx.search(p, left, right, dtype=np.float32)
            for nb in (2, 3, 4, 8):
                ctx.set_host_bands(nb)
                assert np.array_equal(ctx.search(p, left, right, dtype=np.float64), plain64), (bs, cost, nb)
                assert np.array_equal(ctx.search(p, left, right, dtype=np.float32), plain32), (bs, cost, nb)
            if not sub:
                f = oracle.block_left if view == "left" else oracle.block_right
                assert np.array_equal(plain64, f(left, right, bs, 0, 48, cost=cost, threads=8))
        # the automatic choice on a map of more than a megapixel
        ctx.set_host_bands(-1)
        big_l, big_r, _ = make_pair(1500, 1000, 64, seed=96)
        p = wslib.make_params(vid, 7, 0, 64, 1.0, "ssd")
        auto = ctx.search(p, big_l, big_r, dtype=np.float64)
        ctx.set_host_bands(0)
        assert np.array_equal(auto, ctx.search(p, big_l, big_r, dtype=np.float64))


@pytest.mark.parametrize("shape", [(60, 1300), (200, 700), (97, 130)])
def test_left_smooth_factor_across_many_row_bands(wslib, gpu_ctx, oracle, shape):
    """The left view's raster pass runs in bands of 64 rows on separate CUs that hand their last row down through
    device-scope words (ws_smooth.hip): tall images (up to 21 bands here), every kind of window line source
    (LDS windows with compile-time and run-time block sizes, global memory for windows without planes), a factor
    inside and one outside [0, 1], tie-heavy content."""
    w, h = shape
    rng = np.random.default_rng(w + h)
    left, right, _ = make_pair(w, h, 40, seed=w)
    tl = (rng.integers(0, 3, size=(h, w, 3)) * 120).astype(np.uint8)
    tr = (rng.integers(0, 3, size=(h, w, 3)) * 120).astype(np.uint8)
    left[h // 3:h // 3 + 5, 10:30] = 0
    for L, R in ((left, right), (tl, tr)):
        for bs, cost, maxd, s in ((7, "ssd", 40, 0.9), (11, "sad", 24, 0.5), (17, "ssd", 30, 0.9), (19, "ssd", 12, 0.9), (5, "ssd", 40, 1.4)):
            got = wslib.BlockSearch(L, R, bs, 0, maxd, cost=cost, context=gpu_ctx).computeDisparityMapLeft(s)
            want = oracle.block_left(L, R, bs, 0, maxd, smooth=s, cost=cost)
            assert np.array_equal(got, want), (shape, bs, cost, s)
