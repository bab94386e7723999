"""One or two pairs in flight (N contexts taking the pairs alternately, each on its own stream): does one pair's
pre-pass hide behind the other's search, do two searches share the chip?
usage (GPU box): python tools/two_in_flight.py [w h bs D cost view]...   (default: config 2)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, stereo_reconstruction_amd as ws
from stereo_reconstruction_amd.synthetic import make_pair

if sys.argv[1:] == ["--grid"]:   # windows x ranges x costs at config 2's size (the table behind march_nd's rule)
    cases = [["1500", "1000", str(bs), str(D), cost, "left"] for bs in (5, 7, 9, 11, 13, 15, 17) for D in (128, 256, 512)
             for cost in ("ssd", "sad")]
else:
    cases = [a.split(",") for a in sys.argv[1:]] or [["1500", "1000", "7", "256", "ssd", "left"]]
for w, h, bs, D, cost, view in cases:
    w, h, bs, D = int(w), int(h), int(bs), int(D)
    L, R, _ = make_pair(w, h, D, 1)
    tl, tr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
    p = ws.make_params(ws.VIEW_LEFT if view == "left" else ws.VIEW_RIGHT, bs, 0, D, 1.0, cost)
    res = []
    for nctx in (1, 2, 3):
        ctxs = [ws.WindowSearch(0) for _ in range(nctx)]
        outs = [torch.empty((h, w), dtype=torch.float32, device="cuda") for _ in range(nctx)]
        def run(n):
            for i in range(n):
                ctxs[i % nctx].search_device(p, tl, tr, outs[i % nctx], None)
        n = max(12, int(60e-3 / (1e-9 * w * h * D / 2.5)))   # ~60 ms of work per timing
        run(n); torch.cuda.synchronize()
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter(); run(n); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / n)
        assert all(torch.equal(outs[0], o) for o in outs)
        res.append(best * 1e3)
        for c in ctxs: c.close()
    print("%dx%d %dx%d %s D=%d %s: %.4f ms per pair alone, %.4f with 2 in flight, %.4f with 3  (%.0f / %.0f Mdisp/s)" % (
        w, h, bs, bs, cost, D, view, res[0], res[1], res[2], w * h * D / res[0] / 1e3, w * h * D / min(res[1:]) / 1e3), flush=True)
