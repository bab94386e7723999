import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import stereo_reconstruction_amd as ws
from stereo_reconstruction_amd.synthetic import make_pair
ctx = ws.WindowSearch(0)
st = torch.cuda.current_stream().cuda_stream
def t(name, w, h, bs, maxd, s, n=3):
    L, R, _ = make_pair(w, h, maxd, 1)
    tl, tr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    out = torch.empty((h, w), dtype=torch.float32, device="cuda")
    p = ws.make_params(ws.VIEW_LEFT, bs, 0, maxd, s, "ssd")
    p1 = ws.make_params(ws.VIEW_LEFT, bs, 0, maxd, 1.0, "ssd")
    for _ in range(2): ctx.search_device(p, tl, tr, out, st)
    torch.cuda.synchronize()
    ctx.timer_begin(st)
    for _ in range(n): ctx.search_device(p, tl, tr, out, st)
    ms = ctx.timer_end(st) / n
    ctx.timer_begin(st)
    for _ in range(n): ctx.search_device(p1, tl, tr, out, st)
    ms1 = ctx.timer_end(st) / n
    half = (bs - 1) // 2
    steps = (w - 2 * half) + min(64, h - 2 * half) - 1
    print("%-40s call %.3f ms, s=1 call %.3f ms -> smooth passes %.3f ms = %.2f us per diagonal step (%d steps of the first band)" % (name, ms, ms1, ms - ms1, (ms - ms1) * 1e3 / steps, steps), flush=True)
t("900x70 7x7 s=0.9 (ONE band)", 900, 70, 7, 200, 0.9)
t("900x134 7x7 s=0.9 (two bands)", 900, 134, 7, 200, 0.9)
t("900x750 7x7 s=0.9 (12 bands)", 900, 750, 7, 200, 0.9)
t("900x70 7x7 s=1.5 (ONE band, no sums)", 900, 70, 7, 200, 1.5)
t("900x70 17x17 s=0.9 (ONE band)", 900, 86, 17, 200, 0.9)
