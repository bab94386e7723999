import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import stereo_reconstruction_amd as ws
from stereo_reconstruction_amd.synthetic import make_pair
ctx = ws.WindowSearch(0)
st = torch.cuda.current_stream().cuda_stream
for (w, h, bs, maxd, cost) in [(3840, 2160, 9, 1024, "ssd"), (2964, 1988, 9, 512, "sad"), (1500, 1000, 7, 256, "ssd"), (1500,1000,7,512,"ssd")]:
    L, R, _ = make_pair(w, h, maxd, 1)
    tl, tr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    out = torch.empty((h, w), dtype=torch.float32, device="cuda")
    p = ws.make_params(0, bs, 0, maxd, 1.0, cost)
    for _ in range(2): ctx.search_device(p, tl, tr, out, st)
    torch.cuda.synchronize()
    ctx.timer_begin(st)
    for _ in range(5): ctx.search_device(p, tl, tr, out, st)
    ms = ctx.timer_end(st) / 5
    print("WS_MAX_CHUNKS=%s %dx%d D=%d %s: %.3f ms %.0f Mdisp/s plan=%s" % (os.environ.get("WS_MAX_CHUNKS"), w, h, maxd, cost, ms, w*h*maxd/ms/1e3, {k: v for k, v in ws.plan(p, L.shape, R.shape).items() if k in ("x_runs","d_chunks","passes","threads","tiles","strips")}), flush=True)
