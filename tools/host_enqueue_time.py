"""Host time to enqueue one ws_search_device call (planning + launches), against the device time per pair."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, stereo_reconstruction_amd as ws
from stereo_reconstruction_amd.synthetic import make_pair
for (w, h, bs, D, cost) in ((1500, 1000, 7, 256, "ssd"), (450, 375, 5, 64, "sad"), (2964, 1988, 9, 512, "sad")):
    L, R, _ = make_pair(w, h, D, 1)
    tl, tr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    p = ws.make_params(ws.VIEW_LEFT, bs, 0, D, 1.0, cost)
    ctxs = [ws.WindowSearch(0) for _ in range(2)]
    outs = [torch.empty((h, w), dtype=torch.float32, device="cuda") for _ in range(2)]
    for i in range(20): ctxs[i & 1].search_device(p, tl, tr, outs[i & 1], None)
    torch.cuda.synchronize()
    n = 400
    t0 = time.perf_counter()
    for i in range(n): ctxs[i & 1].search_device(p, tl, tr, outs[i & 1], None)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%dx%d %dx%d %s D=%d: enqueue %.1f us per call on the host; %.1f us per pair until the device is done"
          % (w, h, bs, bs, cost, D, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6), flush=True)
