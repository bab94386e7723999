"""Developer tool (GPU box): the left view's banded raster pass under UNEVEN load -- its inter-band hand-off is
the one place where workgroups talk to each other.  Thread A repeats a left-view smoothFactor call and compares
every map with the first one (itself checked against the oracle on a band of rows); threads B and C keep the
chip busy with other searches on their own contexts meanwhile."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereo_reconstruction_amd as ws
from stereo_reconstruction_amd.synthetic import make_pair
from oracle import oracle

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
L, R, _ = make_pair(900, 750, 200, seed=13)
L2, R2, _ = make_pair(1500, 1000, 256, seed=2)
stop = False
bad = []

def load(view):
    ctx = ws.WindowSearch(0)
    b = ws.BlockSearch(L2, R2, 7, 0, 256, context=ctx)
    n = 0
    while not stop:
        (b.computeDisparityMapLeft if view == "left" else b.computeDisparityMapRight)(1.0)
        n += 1
    print("load thread (%s view): %d searches meanwhile" % (view, n), flush=True)

ts = [threading.Thread(target=load, args=(v,)) for v in ("left", "right")]
for t in ts:
    t.start()
ctx = ws.WindowSearch(0)
for bs, s in ((7, 0.9), (17, 0.9), (9, 0.5), (7, -0.5)):
    b = ws.BlockSearch(L, R, bs, 0, 200, context=ctx)
    first = b.computeDisparityMapLeft(s)
    band = oracle.block_left(L, R, bs, 0, 200, smooth=s, rows=(0, 3 + bs), threads=1)
    ok = np.array_equal(first[:3 + bs], band[:3 + bs])
    t0 = time.time()
    diff = 0
    for i in range(reps):
        got = b.computeDisparityMapLeft(s)
        diff += int((got != first).sum())
    print("bs=%d s=%.1f: %d repeats under load, %.1f ms per call, first rows equal the oracle: %s, pixels differing from the first map: %d"
          % (bs, s, reps, (time.time() - t0) / reps * 1e3, ok, diff), flush=True)
    if diff or not ok:
        bad.append((bs, s))
stop = True
for t in ts:
    t.join()
print("ALL OK" if not bad else "FAILED %s" % bad)
