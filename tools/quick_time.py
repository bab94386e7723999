"""Developer tool: ms per device-resident call of the BASELINE shapes (HIP events), one line per configuration, plus the
SHA-1 of each map (to compare against profiles/r0N/full_size_parity.txt).  usage: quick_time.py [2,3,5,1] [reps]"""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import stereo_reconstruction_amd as ws
from stereo_reconstruction_amd.synthetic import make_pair
which = sys.argv[1].split(",") if len(sys.argv) > 1 else ["2", "3"]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
CFG = {"1": (450, 375, 5, 64, "sad", 1), "2": (1500, 1000, 7, 256, "ssd", 2), "3": (2964, 1988, 9, 512, "sad", 3),
       "5": (3840, 2160, 9, 1024, "ssd", 5), "r": (900, 750, 17, 200, "ssd", 13)}
ctx = ws.WindowSearch(0)
st = torch.cuda.current_stream().cuda_stream
tag = " ".join("%s=%s" % (k, os.environ[k]) for k in sorted(os.environ) if k.startswith("WS_"))
for c in which:
    w, h, bs, D, cost, seed = CFG[c]
    L, R, _ = make_pair(w, h, D, seed)
    tl, tr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    out = torch.empty((h, w), dtype=torch.float32, device="cuda")
    p = ws.make_params(ws.VIEW_RIGHT if c == "r" else ws.VIEW_LEFT, bs, 0, D, 1.0, cost)
    for _ in range(3): ctx.search_device(p, tl, tr, out, st)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        ctx.timer_begin(st)
        for _ in range(reps): ctx.search_device(p, tl, tr, out, st)
        best = min(best, ctx.timer_end(st) / reps)
    sha = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]
    print("config %s %dx%d %dx%d %s D=%d: %.4f ms  %.0f Mdisp/s  %s  sha1 %s  %s" % (
        c, w, h, bs, bs, cost, D, best, w * h * D / best / 1e3, ctx.last_launch()["kernel"], sha, tag), flush=True)
