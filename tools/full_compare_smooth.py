"""Whole maps vs the oracle for the smoothFactor paths at the reference's own call shape (900 x 750)."""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereo_reconstruction_amd as ws
from stereo_reconstruction_amd.synthetic import make_pair
from oracle import oracle
ctx = ws.WindowSearch(0)
left, right, _ = make_pair(900, 750, 200, seed=13)
cases = [("right", 17, 0.9, False), ("right", 17, 0.9, True), ("left", 7, 0.9, False), ("left", 7, 1.5, False), ("right", 7, 1.7, False)]
for view, bs, s, vb in cases:
    b = ws.BlockSearch(left, right, bs, 0, 200, context=ctx)
    t0 = time.time()
    got = b.computeDisparityMapLeft(s) if view == "left" else b.computeDisparityMapRight(s, vb, 10.0)
    t1 = time.time()
    print("%s view 900x750 %dx%d SSD D=200 s=%.1f varBlock=%s: device call %.1f ms ..." % (view, bs, bs, s, vb, (t1 - t0) * 1e3), end="", flush=True)
    want = oracle.block_left(left, right, bs, 0, 200, smooth=s) if view == "left" else \
        oracle.block_right(left, right, bs, 0, 200, smooth=s, var_block=vb, thres=10.0)
    print(" identical=%s mismatches=%d sha1=%s (oracle %.0f s)" % (bool(np.array_equal(got, want)), int((got != want).sum()),
          hashlib.sha1(np.ascontiguousarray(got).tobytes()).hexdigest()[:16], time.time() - t1), flush=True)
