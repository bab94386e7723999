"""One-off evidence run: compare WHOLE device maps with the oracle at BASELINE.json's full sizes
(the test-suite compares row bands to stay fast).  Usage on the GPU box:
    python tools/full_compare.py config2 config3 config5 > gpurun_out/full_compare.txt
"""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereo_reconstruction_amd as ws
from stereo_reconstruction_amd.synthetic import make_pair
from oracle import oracle
from bench import WORKLOADS, host_cores

# --sha-only: print only the device maps' hashes (compare with a previous verified run: same hash =
# still identical to the oracle, without the minutes of CPU time)
sha_only = "--sha-only" in sys.argv
ctx = ws.WindowSearch(0)
for name in [a for a in sys.argv[1:] if not a.startswith("--")]:
    w, h, bs, cost, maxd, seed = WORKLOADS[name]
    left, right, _ = make_pair(w, h, maxd, seed)
    for view in (("left",) if "--left-only" in sys.argv else ("left", "right")):
        b = ws.BlockSearch(left, right, bs, 0, maxd, cost=cost, context=ctx)
        t0 = time.time()
        got = b.computeDisparityMapLeft(1.0) if view == "left" else b.computeDisparityMapRight(1.0)
        t1 = time.time()
        if sha_only:
            print("%s %s view sha1(device map)=%s" % (name, view, hashlib.sha1(np.ascontiguousarray(got).tobytes()).hexdigest()[:16]), flush=True)
            continue
        f = oracle.block_left if view == "left" else oracle.block_right
        want = f(left, right, bs, 0, maxd, cost=cost, threads=host_cores())
        t2 = time.time()
        same = bool(np.array_equal(got, want))
        print("%s %s view %dx%d %dx%d %s D=%d: identical=%s  mismatches=%d  sha1(device map)=%s  device call %.1f ms, oracle %.1f s on %d threads"
              % (name, view, w, h, bs, bs, cost.upper(), maxd, same, int((got != want).sum()),
                 hashlib.sha1(np.ascontiguousarray(got).tobytes()).hexdigest()[:16], (t1 - t0) * 1e3, t2 - t1, host_cores()), flush=True)
