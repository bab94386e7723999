"""Where does the host path's time go for buffers the runtime has not seen before?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereo_reconstruction_amd as ws
from stereo_reconstruction_amd.synthetic import make_pair
ctx = ws.WindowSearch(0)
p = ws.make_params(ws.VIEW_LEFT, 7, 0, 256, 1.0, "ssd")
L, R, _ = make_pair(1500, 1000, 256, 3)
ctx.search(p, L, R, dtype=np.float32)
def t(fn, n=8):
    ts = []
    for _ in range(n):
        a = time.perf_counter(); fn(); ts.append(time.perf_counter() - a)
    return min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3
print("same inputs every call                 min %.2f  median %.2f ms" % t(lambda: ctx.search(p, L, R, dtype=np.float32)))
keep = []
def fresh_inputs():
    l2, r2 = L.copy(), R.copy(); keep.append((l2, r2))
    ctx.search(p, l2, r2, dtype=np.float32)
print("fresh input arrays every call (kept)   min %.2f  median %.2f ms" % t(fresh_inputs))
outs = []
def fresh_outputs():
    outs.append(ctx.search(p, L, R, dtype=np.float32))
print("same inputs, outputs kept alive         min %.2f  median %.2f ms" % t(fresh_outputs))
pairs = [(L.copy(), R.copy()) for _ in range(8)]
print("search_many, 8 fresh pairs              %.2f ms per pair" % (t(lambda: ctx.search_many(p, pairs, dtype=np.float32), 3)[0] / 8))
from stereo_reconstruction_amd.synthetic import TRAINING_H
prs = []
for i, (_, w, h, _) in enumerate(TRAINING_H[:6]):
    l, r, _ = make_pair(w, h, 256, 100 + i)
    prs.append((l, r))
for rep in range(3):
    line = []
    for l, r in prs:
        a = time.perf_counter(); ctx.search(p, l, r, dtype=np.float32); line.append("%dx%d %.2f" % (l.shape[1], l.shape[0], (time.perf_counter() - a) * 1e3))
    print("varying sizes, per call ms:", "  ".join(line), flush=True)
# the same with the device entry point only (no copies): is it the search or the copies?
import torch
for rep in range(2):
    line = []
    for l, r in prs:
        tl, tr = torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()
        to = torch.empty((l.shape[0], l.shape[1]), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        a = time.perf_counter(); ctx.search_device(p, tl, tr, to, None); torch.cuda.synchronize(); line.append("%.2f" % ((time.perf_counter() - a) * 1e3))
    print("device entry point, per call ms:", "  ".join(line), flush=True)
