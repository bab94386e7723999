"""Config 2 with one or two pairs in flight (two contexts, two streams): does one pair's pre-pass hide
behind the other's search?  usage (GPU box): python tools/two_in_flight.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, stereo_reconstruction_amd as ws
from stereo_reconstruction_amd.synthetic import make_pair

w, h, D = 1500, 1000, 256
L, R, _ = make_pair(w, h, D, 1)
tl, tr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
p = ws.make_params(ws.VIEW_LEFT, 7, 0, D, 1.0, "ssd")
for nctx in (1, 2, 3):
    ctxs = [ws.WindowSearch(0) for _ in range(nctx)]
    streams = [torch.cuda.Stream() for _ in range(nctx)]
    outs = [torch.empty((h, w), dtype=torch.float32, device="cuda") for _ in range(nctx)]
    def run(n):
        for i in range(n):
            k = i % nctx
            ctxs[k].search_device(p, tl, tr, outs[k], streams[k].cuda_stream)
    run(10); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); run(60); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 60)
    print("%d in flight: %.4f ms per pair  %.0f Mdisp/s  same map: %s" % (nctx, best * 1e3, w * h * D / best / 1e6,
          all(torch.equal(outs[0], o) for o in outs)))
    for c in ctxs: c.close()
