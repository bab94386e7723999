"""Developer tool: WS_HOST_TRACE=1 lines of a few ws_search_host calls at config 2 (where the host's time goes)."""
import os, sys, time
os.environ.setdefault("WS_HOST_TRACE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereo_reconstruction_amd as ws
from stereo_reconstruction_amd.synthetic import make_pair
ctx = ws.WindowSearch(0)
L, R, _ = make_pair(1500, 1000, 256, seed=2)
for dtype in (np.float64, np.float32):
    out = np.empty((1000, 1500), dtype=dtype)
    for bands in [int(b) for b in os.environ.get("BANDS", "-1,0,3,4,6").split(",")]:
        ctx.set_host_bands(bands)
        p = ws.make_params(ws.VIEW_LEFT, 7, 0, 256)
        ts = []
        for i in range(int(os.environ.get("REPS", "8"))):
            t = time.perf_counter(); ctx.search(p, L, R, out=out); ts.append(time.perf_counter() - t)
        print("dtype %s bands %d: min %.3f ms median %.3f" % (dtype.__name__, bands, min(ts) * 1e3, sorted(ts)[4] * 1e3), file=sys.stderr, flush=True)
