"""One config-2 ws_search_host call (CV_64F out, automatic bands) a few times, for a rocprofv3 kernel + copy trace:
  cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/e2e_trace -o run -- python3 $R/tools/e2e_trace.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereo_reconstruction_amd as ws
from stereo_reconstruction_amd.synthetic import make_pair
nb = int(sys.argv[1]) if len(sys.argv) > 1 else -1
L, R, _ = make_pair(1500, 1000, 256, 2)
p = ws.make_params(ws.VIEW_LEFT, 7, 0, 256, 1.0, "ssd")
keep = np.empty((1000, 1500), dtype=np.float64)
with ws.WindowSearch(0) as ctx:
    ctx.set_host_bands(nb)
    for _ in range(6):
        t0 = time.perf_counter(); ctx.search(p, L, R, dtype=np.float64, out=keep); dt = time.perf_counter() - t0
    print("last call %.3f ms" % (dt * 1e3))
