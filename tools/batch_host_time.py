"""Host-buffer batch (config 4 shapes): sequential ws_search_host calls vs the batched ws_enqueue_host / ws_wait path."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereo_reconstruction_amd as ws
from stereo_reconstruction_amd.synthetic import make_pair, TRAINING_H
ctx = ws.WindowSearch(0)
pairs = []
for i, (_, w, h, _) in enumerate(TRAINING_H):
    l, r, _ = make_pair(w, h, 256, 100 + i)
    pairs.append((l, r))
p = ws.make_params(ws.VIEW_LEFT, 7, 0, 256, 1.0, "ssd")
hyps = sum(l.shape[0] * l.shape[1] * 256 for l, _ in pairs)
for name, fn in (("sequential ws_search_host (f32 out)", lambda: [ctx.search(p, l, r, dtype=np.float32) for l, r in pairs]),
                 ("sequential ws_search_host (f64 out)", lambda: [ctx.search(p, l, r, dtype=np.float64) for l, r in pairs]),
                 ("ws_enqueue_host x15 + ws_wait (f32 out)", lambda: ctx.search_many(p, pairs, dtype=np.float32))):
    fn()
    ts = []
    for _ in range(5):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    print("%-42s %.2f ms per batch of 15  (%.0f Mdisp/s)" % (name, min(ts) * 1e3, hyps / min(ts) / 1e6), flush=True)
