"""Developer tool (GPU box): wall time of the left view's smoothFactor call at the pipeline's size, per window and factor
(0.9: the pipeline's; 1.5: the three-best-candidates pre-pass, WS_TOP3_WHOLE=1 for its round-2 form)."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import stereo_reconstruction_amd as ws
from stereo_reconstruction_amd.synthetic import make_pair
ctx = ws.WindowSearch(0)
st = torch.cuda.current_stream().cuda_stream
for s in (0.9, 1.5):
    for bs in (7, 17, 9):
        L, R, _ = make_pair(900, 750, 200, 1)
        tl, tr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
        out = torch.empty((750, 900), dtype=torch.float32, device="cuda")
        p = ws.make_params(ws.VIEW_LEFT, bs, 0, 200, s, "ssd")
        for _ in range(2): ctx.search_device(p, tl, tr, out, st)
        torch.cuda.synchronize()
        ctx.timer_begin(st)
        for _ in range(5): ctx.search_device(p, tl, tr, out, st)
        print(os.environ.get("WS_STEREO_LIB", "tree")[-40:], "left smooth %dx%d s=%.1f: %.3f ms" % (bs, bs, s, ctx.timer_end(st) / 5), flush=True)
