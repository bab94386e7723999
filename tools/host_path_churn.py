"""Alternate banded and plain ws_search_host calls on FRESH numpy buffers, with pageable torch copies of a
megabyte and more in between (the runtime pins those in place): the address churn under which one full test
run once aborted inside a plain call.  Every map is compared.  usage (GPU box): python tools/host_path_churn.py [rounds]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import stereo_reconstruction_amd as ws
from stereo_reconstruction_amd.synthetic import make_pair

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 150
left, right, _ = make_pair(611, 263, 48, seed=95, right_width=590)
big_l, big_r, _ = make_pair(1500, 1000, 64, seed=96)
dev_junk = torch.rand((700, 800), device="cuda")
n = 0
with ws.WindowSearch(0) as ctx:
    for vid in (ws.VIEW_LEFT, ws.VIEW_RIGHT):
        p = ws.make_params(vid, 7, 0, 48, 1.0, "ssd")
        pb = ws.make_params(vid, 7, 0, 64, 1.0, "ssd")
        ctx.set_host_bands(0)
        want64, want32 = ctx.search(p, left, right, dtype=np.float64), ctx.search(p, left, right, dtype=np.float32)
        wantb = ctx.search(pb, big_l, big_r, dtype=np.float64)
        for r in range(rounds):
            for nb in (3, 0, 2, 0, 8, -1):
                ctx.set_host_bands(nb)
                assert np.array_equal(ctx.search(p, left, right, dtype=np.float64), want64), (r, nb)
                junk = dev_junk.cpu().numpy()                       # 2.2 MB pageable D2H into a fresh buffer
                assert np.array_equal(ctx.search(p, left, right, dtype=np.float32), want32), (r, nb)
                del junk
                n += 2
            if r % 10 == 0:
                ctx.set_host_bands(-1 if r % 20 else 0)
                assert np.array_equal(ctx.search(pb, big_l, big_r, dtype=np.float64), wantb), r
                n += 1
            if r % 25 == 0:
                print("view", vid, "round", r, "calls", n, flush=True)
print("host path churn: %d calls, every map identical" % n)
