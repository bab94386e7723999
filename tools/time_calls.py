"""Developer tool: device-resident timing of a few representative calls (ms per call, HIP events)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import stereo_reconstruction_amd as ws
from stereo_reconstruction_amd.synthetic import make_pair
ctx = ws.WindowSearch(0)
st = torch.cuda.current_stream().cuda_stream
def t(name, view, w, h, bs, mind, maxd, s, cost, subpixel=False, n=10):
    L, R, _ = make_pair(w, h, maxd, 1)
    tl, tr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    out = torch.empty((h, w), dtype=torch.float32, device="cuda")
    p = ws.make_params(view, bs, mind, maxd, s, cost, subpixel=subpixel)
    for _ in range(2): ctx.search_device(p, tl, tr, out, st)
    torch.cuda.synchronize()
    ctx.timer_begin(st)
    for _ in range(n): ctx.search_device(p, tl, tr, out, st)
    ms = ctx.timer_end(st) / n
    print("%-52s %8.3f ms  %10.0f Mdisp/s  %s" % (name, ms, w * h * maxd / ms / 1e3, ctx.last_launch()["kernel"]), flush=True)
t("left  1500x1000 7x7 SSD D=256 (config2)", ws.VIEW_LEFT, 1500, 1000, 7, 0, 256, 1.0, "ssd")
t("right 1500x1000 7x7 SSD D=256", ws.VIEW_RIGHT, 1500, 1000, 7, 0, 256, 1.0, "ssd")
t("right 900x750 17x17 SSD D=200 s=0.9 (main.cpp:40)", ws.VIEW_RIGHT, 900, 750, 17, 0, 200, 0.9, "ssd")
t("right 900x750 17x17 SSD D=200 s=1.0", ws.VIEW_RIGHT, 900, 750, 17, 0, 200, 1.0, "ssd")
t("left  900x750 17x17 SSD D=200 s=1.0", ws.VIEW_LEFT, 900, 750, 17, 0, 200, 1.0, "ssd")
t("linear 900x750 range 200 s=1.0", ws.VIEW_LINEAR, 900, 750, 1, 0, 200, 1.0, "ssd")
t("left  2964x1988 9x9 SAD D=512 (config3)", ws.VIEW_LEFT, 2964, 1988, 9, 0, 512, 1.0, "sad", n=5)
t("left  3840x2160 9x9 SSD D=1024 subpixel (config5)", ws.VIEW_LEFT, 3840, 2160, 9, 0, 1024, 1.0, "ssd", subpixel=True, n=3)
t("left  450x375 5x5 SAD D=64 (config1 shape)", ws.VIEW_LEFT, 450, 375, 5, 0, 64, 1.0, "sad")
t("left  900x750 7x7 SSD D=200 s=0.9 (left smooth)", ws.VIEW_LEFT, 900, 750, 7, 0, 200, 0.9, "ssd", n=2)
t("left  900x750 17x17 SSD D=200 s=0.9 (left smooth)", ws.VIEW_LEFT, 900, 750, 17, 0, 200, 0.9, "ssd", n=2)
