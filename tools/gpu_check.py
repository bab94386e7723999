"""Developer script: run a matrix of small cases on the GPU and report mismatches vs the oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereo_reconstruction_amd as ws
from stereo_reconstruction_amd.synthetic import make_pair
from oracle import oracle

def report(name, got, ref):
    if got.shape != ref.shape:
        print("FAIL", name, "shape", got.shape, ref.shape); return False
    bad = np.argwhere(got != ref)
    if len(bad) == 0:
        print("ok  ", name); return True
    ys, xs = bad[:, 0], bad[:, 1]
    print("FAIL", name, "mismatches", len(bad), "of", got.size,
          "y[%d..%d] x[%d..%d]" % (ys.min(), ys.max(), xs.min(), xs.max()))
    for y, x in bad[:8]:
        print("     (y=%d,x=%d) got %s want %s" % (y, x, got[y, x], ref[y, x]))
    return False

ctx = ws.WindowSearch(0)
ok = True
rng = np.random.default_rng(1)
cases = [
    # view, bs, minD, maxD, cost, (w,h), (w2,h2) or None, levels
    ("left", 7, 0, 64, "ssd", (300, 80), None, 256),
    ("left", 7, 0, 64, "sad", (300, 80), None, 256),
    ("left", 5, 0, 40, "sad", (257, 61), (250, 66), 256),
    ("left", 9, 0, 100, "ssd", (420, 70), None, 4),
    ("left", 3, 0, 17, "ssd", (100, 40), None, 2),
    ("right", 7, 0, 64, "ssd", (300, 80), None, 256),
    ("right", 7, 2, 50, "sad", (300, 80), (310, 78), 256),
    ("right", 9, 0, 90, "ssd", (333, 64), None, 4),
    ("right", 5, 0, 33, "sad", (200, 50), None, 2),
    ("left", 11, 0, 30, "ssd", (120, 50), None, 256),
    ("right", 17, 0, 40, "ssd", (150, 60), None, 256),
    ("right", 17, 0, 200, "ssd", (450, 90), None, 256),
    ("left", 17, 0, 100, "ssd", (300, 70), None, 4),
    ("left", 15, 0, 64, "sad", (300, 70), None, 2),
    ("right", 13, 3, 64, "sad", (300, 70), (280, 66), 256),
    ("left", 21, 0, 30, "ssd", (120, 50), None, 256),   # generic path
    ("left", 1, 0, 20, "ssd", (90, 30), None, 256),
]
for view, bs, mind, maxd, cost, (w, h), wh2, levels in cases:
    w2, h2 = wh2 if wh2 else (w, h)
    if levels == 256:
        L, R, _ = make_pair(w, h, maxd, seed=bs * 1000 + maxd, right_width=w2, right_height=h2)
    else:
        L = (rng.integers(0, levels, size=(h, w, 3)) * (255 // (levels - 1))).astype(np.uint8)
        R = (rng.integers(0, levels, size=(h2, w2, 3)) * (255 // (levels - 1))).astype(np.uint8)
    L[h // 2, w // 3] = 0
    R[h2 // 3, w2 // 2] = 0
    name = "%s bs=%d d=[%d,%d] %s %dx%d/%dx%d lv=%d" % (view, bs, mind, maxd, cost, w, h, w2, h2, levels)
    try:
        if view == "left":
            ref = oracle.block_left(L, R, bs, mind, maxd, cost=cost, threads=8)
            got = ws.BlockSearch(L, R, bs, mind, maxd, cost=cost, context=ctx).computeDisparityMapLeft(1.0)
        else:
            ref = oracle.block_right(L, R, bs, mind, maxd, cost=cost, threads=8)
            got = ws.BlockSearch(L, R, bs, mind, maxd, cost=cost, context=ctx).computeDisparityMapRight(1.0)
        print("   ", ctx.last_launch())
        ok &= report(name, got, ref)
    except Exception as e:
        print("EXC ", name, repr(e)); ok = False
# linear
L, R, _ = make_pair(260, 40, 64, seed=5)
ok &= report("linear", ws.LinearSearch(L, R, context=ctx).computeDisparityMap(1.0), oracle.linear(L, R))
# subpixel
L, R, _ = make_pair(300, 60, 64, seed=9)
for view in ("left", "right"):
    for cost in ("ssd", "sad"):
        if view == "left":
            ref = oracle.block_left(L, R, 7, 0, 64, cost=cost, subpixel=True, threads=8)
            got = ws.BlockSearch(L, R, 7, 0, 64, cost=cost, subpixel=True, context=ctx).computeDisparityMapLeft(1.0)
        else:
            ref = oracle.block_right(L, R, 7, 0, 64, cost=cost, subpixel=True, threads=8)
            got = ws.BlockSearch(L, R, 7, 0, 64, cost=cost, subpixel=True, context=ctx).computeDisparityMapRight(1.0)
        err = np.abs(got - ref).max()
        print("subpixel", view, cost, "max err", err, "refined px", int((ref != np.round(ref)).sum()))
        ok &= err <= 1e-4
print("ALL OK" if ok else "SOME FAILED")
sys.exit(0 if ok else 1)
