"""Developer tool (GPU box): one ws_search_host call at config 2 with 0 / 2 / 3 / 4 / 6 / 8 row bands."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereo_reconstruction_amd as ws
from stereo_reconstruction_amd.synthetic import make_pair
ctx = ws.WindowSearch(0)
for name, (w, h, bs, maxd, cost) in {"config2": (1500, 1000, 7, 256, "ssd"), "config3": (2964, 1988, 9, 512, "sad")}.items():
    L, R, _ = make_pair(w, h, maxd, seed=3)
    p = ws.make_params(ws.VIEW_LEFT, bs, 0, maxd, 1.0, cost)
    for dt in (np.float64, np.float32):
        keep = np.empty((h, w), dtype=dt)
        for nb in (0, 2, 3, 4, 5, 6):
            ctx.set_host_bands(nb)
            res = []
            for out in (None, keep):
                for _ in range(3):
                    ctx.search(p, L, R, dtype=dt, out=out)
                ts = []
                for _ in range(15):
                    t = time.perf_counter(); ctx.search(p, L, R, dtype=dt, out=out); ts.append(time.perf_counter() - t)
                res.append(sorted(ts)[7] * 1e3)
            print("%s %-8s bands=%d  median %.3f ms with a fresh output array per call, %.3f ms into a kept one" % (name, dt.__name__, nb, res[0], res[1]), flush=True)
