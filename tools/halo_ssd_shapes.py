"""Developer tool (GPU box): 7x7 / 9x9 SSD searches at shapes of several rounds, kernel ms of the default plan (run twice:
as it is, and with WS_MARCH_HALO_SSD=0 for the plain kernel)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import stereo_reconstruction_amd as ws
from stereo_reconstruction_amd.synthetic import make_pair
ctx = ws.WindowSearch(0)
st = torch.cuda.current_stream().cuda_stream
for (w, h, bs, D) in ((2964, 1988, 7, 512), (3840, 2160, 7, 256), (2964, 1988, 9, 512), (1500, 1000, 9, 256), (2000, 1500, 7, 256)):
    L, R, _ = make_pair(w, h, D, 1)
    tl, tr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    out = torch.empty((h, w), dtype=torch.float32, device="cuda")
    p = ws.make_params(ws.VIEW_LEFT, bs, 0, D, 1.0, "ssd")
    ctx.set_profiling(True)
    ts = []
    for _ in range(6):
        ctx.search_device(p, tl, tr, out, st)
        ts.append(ctx.last_kernel_ms())
    ctx.set_profiling(False)
    info = ctx.last_launch()
    print("%dx%d %dx%d SSD D=%d: %.4f ms  %s threads %d workgroups %d" % (w, h, bs, bs, D, float(np.median(ts[1:])), info["kernel"], info["threads"], info["workgroups"]), flush=True)
