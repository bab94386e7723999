import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, time
import stereo_reconstruction_amd as ws
from stereo_reconstruction_amd.synthetic import make_pair
from oracle import oracle
ctx = ws.WindowSearch(0)
rng = np.random.default_rng(5)
ok = True
def cmp(name, got, ref):
    global ok
    bad = np.argwhere(got != ref)
    print(("ok  " if len(bad) == 0 else "FAIL"), name, len(bad), bad[:4].tolist(), [(got[tuple(b)], ref[tuple(b)]) for b in bad[:4]], flush=True)
    ok &= len(bad) == 0
for levels, (w, h) in [(256, (300, 60)), (3, (260, 50)), (2, (500, 30)), (4, (1100, 24)), (256, (97, 41)), (256, (60, 1300)), (3, (4200, 20))]:
    if levels == 256:
        L, R, _ = make_pair(w, h, 48, seed=w)
    else:
        L = (rng.integers(0, levels, size=(h, w, 3)) * (255 // (levels - 1))).astype(np.uint8)
        R = (rng.integers(0, levels, size=(h, w, 3)) * (255 // (levels - 1))).astype(np.uint8)
    L[10:14, 20:60] = 0
    for s in (0.9, 0.5, 0.0, 0.999, 1.2, 3.0, -0.7, -1.5):
        for bs, cost, maxd in ((7, 'ssd', 48), (5, 'sad', 30), (1, 'ssd', 20), (17, 'ssd', 40)):
            ref = oracle.block_left(L, R, bs, 0, maxd, smooth=s, cost=cost)
            got = ws.BlockSearch(L, R, bs, 0, maxd, cost=cost, context=ctx).computeDisparityMapLeft(s)
            cmp("left lv=%d %dx%d s=%.3f bs=%d %s" % (levels, w, h, s, bs, cost), got, ref)
# unequal sizes
L, R, _ = make_pair(310, 75, 48, seed=77, right_width=290, right_height=70)
cmp("left unequal", ws.BlockSearch(L, R, 7, 0, 48, context=ctx).computeDisparityMapLeft(0.9), oracle.block_left(L, R, 7, 0, 48, smooth=0.9))
L, R, _ = make_pair(900, 750, 200, seed=13)
import torch
t = time.time(); got = ws.BlockSearch(L, R, 17, 0, 200, context=ctx).computeDisparityMapLeft(0.9); print("900x750 bs17 D200 s=0.9 host call %.1f ms" % ((time.time() - t) * 1e3))
band = oracle.block_left(L, R, 17, 0, 200, smooth=0.9, rows=(0, 20), threads=1)
cmp("left 900x750 band", got[:20], band[:20])
t = time.time(); got = ws.BlockSearch(L, R, 17, 0, 200, context=ctx).computeDisparityMapLeft(1.5); print("900x750 bs17 D200 s=1.5 host call %.1f ms" % ((time.time() - t) * 1e3))
band = oracle.block_left(L, R, 17, 0, 200, smooth=1.5, rows=(0, 20), threads=1)
cmp("left 900x750 band s=1.5", got[:20], band[:20])
L, R, _ = make_pair(1500, 1000, 256, seed=2)
t = time.time(); got = ws.BlockSearch(L, R, 7, 0, 256, context=ctx).computeDisparityMapLeft(1.5); print("1500x1000 bs7 D256 s=1.5 host call %.1f ms" % ((time.time() - t) * 1e3))
band = oracle.block_left(L, R, 7, 0, 256, smooth=1.5, rows=(0, 24), threads=1)
cmp("left 1500x1000 band s=1.5", got[:24], band[:24])
print("ALL OK" if ok else "SOME FAILED")
