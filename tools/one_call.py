import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, stereo_reconstruction_amd as ws
from stereo_reconstruction_amd.synthetic import make_pair
view, w, h, bs, mind, maxd, s, cost = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), float(sys.argv[7]), sys.argv[8]
ctx = ws.WindowSearch(0)
L, R, _ = make_pair(w, h, maxd, 1)
tl, tr = torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()
out = torch.empty((h, w), dtype=torch.float32, device="cuda")
sub = len(sys.argv) > 9 and sys.argv[9] == "subpixel"
p = ws.make_params({"left": 0, "right": 1, "linear": 2}[view], bs, mind, maxd, s, cost, subpixel=sub)
for _ in range(int(os.environ.get("WS_CALLS", "5"))): ctx.search_device(p, tl, tr, out, None)
torch.cuda.synchronize()
