"""Developer tool (GPU box): where a step of the left view's smoothFactor raster pass spends its time.
Needs the instrumented build (python tools/variants.py build "8,8,512,WS_BAND_STAMPS" here, then on the box
WS_STEREO_LIB=gpurun_variants/libws_8_8_512_WS_BAND_STAMPS.so python tools/band_stamps.py): s_memtime at five points of a step,
128 steps of every band.  Segments: 0-1 flush + window-column fills + the loads of step k+3; 1-2 upper neighbour by DPP / hand-off
word; 2-3 the two sliding sums; 3-4 the decision; 4-0' loop overhead to the next step's first stamp."""
import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import stereo_reconstruction_amd as ws
from stereo_reconstruction_amd.synthetic import make_pair
lib = ctypes.CDLL(os.environ["WS_STEREO_LIB"])
ctx = ws.WindowSearch(0)
st = torch.cuda.current_stream().cuda_stream
for bs in (7, 17):
    L, R, _ = make_pair(900, 750, 200, 1)
    tl, tr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    out = torch.empty((750, 900), dtype=torch.float32, device="cuda")
    p = ws.make_params(ws.VIEW_LEFT, bs, 0, 200, 0.9, "ssd")
    for _ in range(3): ctx.search_device(p, tl, tr, out, st)
    torch.cuda.synchronize()
    ctx.timer_begin(st)
    ctx.search_device(p, tl, tr, out, st)
    ms = ctx.timer_end(st)
    nb, ns, npnt = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    buf = np.zeros(32 * 128 * 8, dtype=np.uint64)
    rc = lib.ws_debug_band_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.byref(nb), ctypes.byref(ns), ctypes.byref(npnt))
    s = buf.reshape(nb.value, ns.value, npnt.value).astype(np.int64)
    print("== %dx%d: call %.3f ms (instrumented), rc %d" % (bs, bs, ms, rc))
    nbands = -(-(750 - 2 * (bs // 2)) // 32)
    print("band   period  seg0-1  seg1-2  seg2-3  seg3-4  seg4-0'   (ticks of s_memtime, medians over 127 steps)   start of step 300")
    for b in range(min(nbands, nb.value)):
        t = s[b]
        if t[0, 0] == 0: continue
        per = np.median(t[1:, 0] - t[:-1, 0])
        segs = [np.median(t[:, i + 1] - t[:, i]) for i in range(4)]
        tail = np.median(t[1:, 0] - t[:-1, 4])
        print("%4d  %7.0f %7.0f %7.0f %7.0f %7.0f %7.0f      %d" % (b, per, *segs, tail, t[0, 0] - s[0, 0, 0]))
    t = s[:min(nbands, nb.value)]
    print("all bands: period mean %.1f  p10 %.0f  p50 %.0f  p90 %.0f  max %.0f" % ((t[:, 1:, 0] - t[:, :-1, 0]).mean(),
          *np.percentile(t[:, 1:, 0] - t[:, :-1, 0], [10, 50, 90, 100])))
    for i, name in enumerate(("fills+loads", "up/hand-off", "sliding sums", "decision")):
        d = t[:, :, i + 1] - t[:, :, i]
        print("   %-14s mean %.1f  p10 %.0f  p50 %.0f  p90 %.0f  max %.0f" % (name, d.mean(), *np.percentile(d, [10, 50, 90, 100])))
    if npnt.value >= 8:  # inside the decision (stamps 5-7 are only written by steps whose lane 0 pixel has a list: zeros otherwise)
        ok = (t[:, :, 5] > t[:, :, 3]) & (t[:, :, 7] > t[:, :, 5]) & (t[:, :, 4] > t[:, :, 7])
        for name, a, b in (("list", 3, 5), ("upper value", 5, 6), ("left value", 6, 7), ("rest", 7, 4)):
            d = (t[:, :, b] - t[:, :, a])[ok]
            print("      %-12s mean %.1f  p10 %.0f  p50 %.0f  p90 %.0f  max %.0f   (%d steps)" % (name, d.mean(), *np.percentile(d, [10, 50, 90, 100]), d.size))
