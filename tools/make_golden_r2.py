"""Round-2 golden fixtures (run in the build container only): the branches that had no fixture yet -- right view
and LinearSearch with smoothFactor (the pipeline's own call is computeDisparityMapRight(17, 0, 200, 0.9),
main.cpp:40), varBlock, the left view's raster dependency -- on crops of the reference's own rectified Teddy
pair (results/Rectified/trainingH/Teddy).  Expected maps come from the literal raster-order witnesses in
oracle/brute.py (block_right_py, linear_py, block_left_smooth_py), NOT from ws_oracle.c, so they pin the C
restatement as well as the HIP path.  Fixtures are pixels + expected maps + parameters."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import brute  # noqa: E402
from tools.make_golden import bgr, OUT, REF  # noqa: E402


def save(name, L, R, exp, view, bs, mind, maxd, cost, smooth, var_block=False, thres=19.0, max_block=0):
    np.savez_compressed(os.path.join(OUT, name + ".npz"), left=L, right=R, expected=exp.astype(np.float32), view=view,
                        block_size=bs, min_disparity=mind, max_disparity=maxd, cost=cost, smooth=smooth,
                        var_block=int(var_block), thres=thres, max_block=max_block)
    print(name, L.shape, R.shape, "nonzero", int((exp != 0).sum()), "zeros", int((exp == 0).sum()))


def main():
    tl = bgr(REF + "/results/Rectified/trainingH/Teddy/rectifiedLeft.png")
    tr = bgr(REF + "/results/Rectified/trainingH/Teddy/rectifiedRight.png")
    # the reference's own call shape on a crop: right view, 17x17, minDisparity 0, smoothFactor 0.9
    L, R = tl[500:540, 200:290], tr[500:540, 200:290]
    exp, _ = brute.block_right_py(L, R, 17, 0, 40, 0.9, "ssd")
    save("teddy_rect_right_ssd17_s09", L, R, exp, "right", 17, 0, 40, "ssd", 0.9)
    exp, _ = brute.block_right_py(L, R, 7, 0, 30, 0.5, "sad")
    save("teddy_rect_right_sad7_s05", L, R, exp, "right", 7, 0, 30, "sad", 0.5)
    # varBlock on a crop with little texture (the wall behind the teddy), threshold as ImageRectifier passes it (10)
    L, R = tl[60:96, 560:640], tr[60:96, 560:640]
    exp, mb = brute.block_right_py(L, R, 9, 0, 24, 1.0, "ssd", var_block=True, thres=10.0)
    save("teddy_rect_right_ssd9_varblock", L, R, exp, "right", 9, 0, 24, "ssd", 1.0, True, 10.0, mb)
    exp, mb = brute.block_right_py(L, R, 5, 0, 24, 0.9, "ssd", var_block=True, thres=60.0)
    save("teddy_rect_right_ssd5_varblock_s09", L, R, exp, "right", 5, 0, 24, "ssd", 0.9, True, 60.0, mb)
    # LinearSearch with smoothFactor (rectification_main.cpp:194-195 calls it with 1.0)
    L, R = tl[300:324, 400:528], tr[300:324, 330:458]
    save("teddy_rect_linear_s05", L, R, brute.linear_py(L, R, 0.5, 200), "linear", 1, 0, 200, "ssd", 0.5)
    # left view with the raster dependency
    L, R = tl[300:330, 400:480], tr[300:330, 360:440]
    save("teddy_rect_left_ssd5_s09", L, R, brute.block_left_smooth_py(L, R, 5, 24, 0.9, "ssd"), "left", 5, 0, 24, "ssd", 0.9)


if __name__ == "__main__":
    main()
