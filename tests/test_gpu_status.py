"""Kernel-side trouble must reach the caller as a status, not as a silently wrong map: the left view's smoothFactor
raster pass (BlockSearch.cpp:68-73 in raster order; ws_smooth_left_bands_kernel) runs 64-row bands on separate CUs
that poll for the band above; a poll that never succeeds gives up after a bounded number of tries.  Round 2 set a flag
nobody read.  WS_BAND_SPIN_LIMIT=-1 (development knob, read once per process) makes every band below the first give up
at its first unsuccessful poll, so the path can be driven on purpose -- in a child process."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CHILD = r"""
import numpy as np, torch
import stereo_reconstruction_amd as ws
from stereo_reconstruction_amd.synthetic import make_pair
left, right, _ = make_pair(300, 200, 32, seed=7)
ctx = ws.WindowSearch(0)
p = ws.make_params(ws.VIEW_LEFT, 7, 0, 32, 0.9)
try:
    ctx.search(p, left, right)
    print("HOST: no error")
except ws.WsError as e:
    print("HOST:", e.code, "gave up" in str(e))
# the next call starts clean (and fails again, the knob is still set); a smoothFactor-1 call is not affected
assert np.array_equal(ctx.search(ws.make_params(ws.VIEW_LEFT, 7, 0, 32, 1.0), left, right),
                      ctx.search(ws.make_params(ws.VIEW_LEFT, 7, 0, 32, 1.0), left, right))
tl, tr = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
out = torch.empty((200, 300), dtype=torch.float32, device="cuda")
ctx.search_device(p, tl, tr, out)
try:
    ctx.device_status()
    print("DEVICE: no error")
except ws.WsError as e:
    print("DEVICE:", e.code, "gave up" in str(e))
ctx.device_status()          # flagged once, reported once
print("DONE")
"""


def test_a_band_that_gives_up_fails_the_call():
    env = dict(os.environ, WS_BAND_SPIN_LIMIT="-1", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "HOST: -4 True" in r.stdout and "DEVICE: -4 True" in r.stdout and "DONE" in r.stdout, r.stdout + r.stderr


def test_no_flag_without_the_knob(wslib, gpu_ctx, oracle):
    import numpy as np
    from stereo_reconstruction_amd.synthetic import make_pair
    left, right, _ = make_pair(300, 200, 32, seed=7)
    got = wslib.BlockSearch(left, right, 7, 0, 32, context=gpu_ctx).computeDisparityMapLeft(0.9)
    gpu_ctx.device_status()
    assert np.array_equal(got, oracle.block_left(left, right, 7, 0, 32, smooth=0.9))


# The raster pass has two forms of its compile-time-window kernel: with a second, row-major copy of the LDS windows (the
# default wherever both copies fit) and without (wide disparity ranges; smoothFactor >= 1).  WS_LEFT_TW=0 (development
# knob, read once per process) takes the second for every call, so that both are held against the oracle on the same
# inputs -- sliding sums included (smoothFactor < 1) -- whatever shapes the other tests happen to use.
CHILD_FORMS = r"""
import sys
import numpy as np
import stereo_reconstruction_amd as ws
from stereo_reconstruction_amd.synthetic import make_pair
from oracle import oracle
ctx = ws.WindowSearch(0)
for (w, h, bs, D, s, cost, seed) in ((260, 150, 7, 48, 0.9, "ssd", 3), (230, 140, 17, 40, 0.5, "ssd", 4), (250, 130, 9, 64, 0.9, "sad", 5),
                                      (240, 120, 5, 32, -0.5, "ssd", 6)):
    left, right, _ = make_pair(w, h, D, seed=seed)
    got = ws.BlockSearch(left, right, bs, 0, D, context=ctx, cost=cost).computeDisparityMapLeft(s)
    want = oracle.block_left(left, right, bs, 0, D, smooth=s, cost=cost)
    print("FORM", bs, s, cost, bool(np.array_equal(got, want)))
print("DONE")
"""


@pytest.mark.parametrize("knob", ["0", "1"])
def test_both_forms_of_the_raster_pass_equal_the_oracle(knob):
    env = dict(os.environ, WS_LEFT_TW=knob, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", CHILD_FORMS], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("FORM")]
    assert len(lines) == 4 and all(l.endswith("True") for l in lines) and "DONE" in r.stdout, r.stdout + r.stderr
