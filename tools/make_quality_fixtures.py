"""Generate tests/golden/{teddyH,artL}_pair.npz from the reference's DATA files (build container only).

The two trainingH scenes whose ground truth is in the reference tree (data/MiddEval3/trainingH/Teddy,
.../ArtL: im0.png, im1.png, disp0GT.pfm, mask0nocc.png, calib.txt) at their full half-resolution size --
the scale the reference's own driver runs at (main.cpp:20: scene 13 = Teddy, 900 x 750).  They carry the
quality leg of BASELINE.json's metric: bad-2.0 = evaldisp(disp, disp0GT, mask0nocc, 2.0, ndisp, 0)
(utils.cpp:123-168) of the left-view map, for the device and for the CPU oracle (which must agree).

Fixtures are data: decoded pixels (PIL, RGB -> BGR as cv::imread gives them), the decoded PFM floats, the
mask, ndisp from calib.txt.  No expected disparity map is stored: bad-2.0 is computed at run time from
both implementations, and the maps themselves are compared bit for bit.
"""
import os
import re
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.make_golden import bgr, read_pfm_py  # noqa: E402

REF = "/root/reference/data/MiddEval3/trainingH"
OUT = os.path.join(ROOT, "tests", "golden")


def main():
    for scene, name in (("Teddy", "teddyH_pair"), ("ArtL", "artL_pair")):
        d = os.path.join(REF, scene)
        im0, im1 = bgr(d + "/im0.png"), bgr(d + "/im1.png")
        gt = read_pfm_py(d + "/disp0GT.pfm")
        mask = np.ascontiguousarray(np.asarray(Image.open(d + "/mask0nocc.png")))
        calib = open(d + "/calib.txt").read()
        ndisp = int(re.search(r"ndisp=(\d+)", calib).group(1))
        assert im0.shape == im1.shape and gt.shape == im0.shape[:2] == mask.shape
        np.savez_compressed(os.path.join(OUT, name + ".npz"), left=im0, right=im1, gt=gt, mask=mask,
                            ndisp=ndisp, calib=calib)
        print(name, im0.shape, "ndisp", ndisp, "finite gt %.1f %%" % (100 * np.isfinite(gt).mean()),
              "mask==255 %.1f %%" % (100 * (mask == 255).mean()),
              "%.2f MB" % (os.path.getsize(os.path.join(OUT, name + ".npz")) / 1e6))


if __name__ == "__main__":
    main()
