"""Time the host-buffer entry point (what the C++ facade calls): H2D + search + D2H, pageable numpy buffers."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereo_reconstruction_amd as ws
from stereo_reconstruction_amd.synthetic import make_pair
ctx = ws.WindowSearch(0)
for name, (w, h, bs, maxd, view, s) in {"config2 left": (1500, 1000, 7, 256, "left", 1.0),
                                        "default right s=0.9": (900, 750, 17, 200, "right", 0.9),
                                        "config3 left": (2964, 1988, 9, 512, "left", 1.0)}.items():
    L, R, _ = make_pair(w, h, maxd, seed=3)
    b = ws.BlockSearch(L, R, bs, 0, maxd, context=ctx)
    f = (lambda: b.computeDisparityMapLeft(s)) if view == "left" else (lambda: b.computeDisparityMapRight(s))
    f()
    ts = []
    for _ in range(10):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    print("%-24s host call: min %.3f ms  median %.3f ms" % (name, min(ts) * 1e3, sorted(ts)[5] * 1e3), flush=True)
