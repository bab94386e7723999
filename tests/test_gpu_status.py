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
