"""The one disparity image the reference itself holds (run in the build container only): a crop of
results/PerceptualWindowSearch/test_result.png with the matching crops of data/MiddEval3/trainingH/Teddy/im{0,1}.png,
for tests/test_oracle_golden.py::test_reference_held_disparity_image_is_plausible.

NOT a pin: the stored image is the output of a related revision of the reference (a centred 17 x 17 window and an
8-pixel zero ring, where BlockSearch.cpp:88-179 as it stands has a 16 x 16 window and no ring), so the oracle agrees
with it on ~94 % of the interior pixels, not bit for bit.  It is the only check of the oracle's output that does not
come from the builder's own hands.  The fixture is data: pixels, the stored map's crop, parameters."""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
BS, DMAX = 17, 200
Y0, Y1 = 330, 386   # rows of the crop (8 rows of window halo each side of 40 compared rows)
X0, W = 150, 216    # right-view columns of the crop (8 columns of halo each side of 200 compared columns)


def main():
    disp = np.array(Image.open(REF + "/results/PerceptualWindowSearch/test_result.png"))
    left = np.array(Image.open(REF + "/data/MiddEval3/trainingH/Teddy/im0.png"))[:, :, ::-1]  # RGB -> BGR (cv::imread)
    right = np.array(Image.open(REF + "/data/MiddEval3/trainingH/Teddy/im1.png"))[:, :, ::-1]
    half = (BS - 1) // 2
    # the right view looks at left columns x + d + [-half, half): the left crop is DMAX + half columns wider
    R = np.ascontiguousarray(right[Y0:Y1, X0:X0 + W])
    L = np.ascontiguousarray(left[Y0:Y1, X0:X0 + W + DMAX + half])
    stored = disp[Y0:Y1, X0:X0 + W]
    got = oracle.block_right(L, R, BS, 0, DMAX, threads=8)
    inner = (slice(half, Y1 - Y0 - half), slice(half, W - half))
    agree = float((got[inner] == stored[inner]).mean())
    near = float((np.abs(got[inner] - stored[inner]) <= 1).mean())
    print("crop rows %d..%d cols %d..%d: exact agreement %.4f, within one disparity %.4f" % (Y0, Y1, X0, X0 + W, agree, near))
    np.savez_compressed(os.path.join(OUT, "teddyH_reference_disparity_crop.npz"), left=L, right=R, stored=stored,
                        block_size=BS, min_disparity=0, max_disparity=DMAX, halo=half,
                        agreement_when_made=agree)
    print(os.path.getsize(os.path.join(OUT, "teddyH_reference_disparity_crop.npz")), "bytes")


if __name__ == "__main__":
    main()
