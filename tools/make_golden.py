"""Generate tests/golden/*.npz from the reference's data files (run in the build container only).

Inputs are small crops of images that live in the reference tree:
  results/Rectified/trainingH/Teddy/rectified{Left,Right}.png   (the exact BlockSearch inputs the
      pipeline produces, rectification_main.cpp:189-192)
  results/Rectified/trainingH/MotorcycleE/rectified*.png        (unequal sizes, black border)
  data/MiddEval3/trainingH/Teddy/im{0,1}.png + disp0GT.pfm + mask0nocc.png + calib.txt
Expected outputs come from oracle/brute.py (the independent NumPy brute force), NOT from
ws_oracle.c, so the fixtures pin the C restatement as well as the HIP path.  PNGs are decoded
with PIL and converted RGB -> BGR (cv::imread order).  Nothing of the reference's source
travels: fixtures are pixels + expected maps.
"""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import brute  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def bgr(path):
    return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])


def read_pfm_py(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"Pf"
        w, h = map(int, f.readline().split())
        scale = float(f.readline())
        a = np.frombuffer(f.read(w * h * 4), dtype="<f4" if scale < 0 else ">f4").reshape(h, w)
    return np.ascontiguousarray(a[::-1]).astype(np.float32)


def write_pfm_py(path, a):
    with open(path, "wb") as f:
        f.write(b"Pf\n%d %d\n-0.003922\n" % (a.shape[1], a.shape[0]))
        f.write(np.ascontiguousarray(a[::-1], dtype="<f4").tobytes())


def case(name, L, R, view, bs, mind, maxd, cost):
    if view == "left":
        exp = brute.block_left(L, R, bs, mind, maxd, cost)
    elif view == "right":
        exp = brute.block_right(L, R, bs, mind, maxd, cost)
    else:
        exp = brute.linear(L, R, maxd)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), left=L, right=R, expected=exp.astype(np.float32),
                        view=view, block_size=bs, min_disparity=mind, max_disparity=maxd, cost=cost)
    print(name, L.shape, R.shape, "nonzero", int((exp != 0).sum()))


def main():
    os.makedirs(OUT, exist_ok=True)
    tl = bgr(REF + "/results/Rectified/trainingH/Teddy/rectifiedLeft.png")
    tr = bgr(REF + "/results/Rectified/trainingH/Teddy/rectifiedRight.png")
    ml = bgr(REF + "/results/Rectified/trainingH/MotorcycleE/rectifiedLeft.png")
    mr = bgr(REF + "/results/Rectified/trainingH/MotorcycleE/rectifiedRight.png")
    print("teddy rect", tl.shape, tr.shape, "motorcycleE rect", ml.shape, mr.shape)
    # Teddy rectified crop (same rows of both images; right crop starts further left so matches exist)
    L = tl[300:364, 400:528]
    R = tr[300:364, 330:458]
    case("teddy_rect_left_ssd7", L, R, "left", 7, 0, 48, "ssd")
    case("teddy_rect_left_sad5", L, R, "left", 5, 0, 64, "sad")
    case("teddy_rect_right_ssd7", L, R, "right", 7, 0, 48, "ssd")
    case("teddy_rect_right_sad9", L, R, "right", 9, 2, 40, "sad")
    case("teddy_rect_linear", L[:24], R[:24], "linear", 1, 0, 200, "ssd")
    # MotorcycleE: unequal sizes + black border (top-left corner region of the warped images)
    L = ml[0:56, 0:120]
    R = mr[0:60, 0:112]
    case("motorcycleE_rect_left_ssd5", L, R, "left", 5, 0, 32, "ssd")
    # (the right view with this pair would make the reference throw: left image shorter than right)
    case("motorcycleE_rect_right_ssd5", ml[0:60, 0:120], mr[0:56, 0:112], "right", 5, 0, 32, "ssd")
    # bottom-right corner: black border on the other side, different sizes again
    L = ml[-48:, -104:]
    R = mr[-52:, -110:]
    case("motorcycleE_corner_left_sad7", L, R, "left", 7, 0, 40, "sad")
    # reference default window (17) on a small crop (right view, as main.cpp:40 calls it)
    L = tl[500:548, 200:296]
    R = tr[500:548, 200:296]
    case("teddy_rect_right_ssd17", L, R, "right", 17, 0, 40, "ssd")

    # config 1 plumbing fixture: Teddy quarter-res pair (2x box down-sampling of the half-res im0/im1)
    im0 = bgr(REF + "/data/MiddEval3/trainingH/Teddy/im0.png").astype(np.uint16)
    im1 = bgr(REF + "/data/MiddEval3/trainingH/Teddy/im1.png").astype(np.uint16)
    q = lambda a: ((a[0::2, 0::2] + a[1::2, 0::2] + a[0::2, 1::2] + a[1::2, 1::2] + 2) // 4).astype(np.uint8)
    q0, q1 = q(im0), q(im1)
    gt = read_pfm_py(REF + "/data/MiddEval3/trainingH/Teddy/disp0GT.pfm")
    mask = np.asarray(Image.open(REF + "/data/MiddEval3/trainingH/Teddy/mask0nocc.png"))
    # quarter-res ground truth: sub-sample and halve the disparities; mask sub-sampled
    gtq = (gt[0::2, 0::2] / 2.0).astype(np.float32)
    maskq = np.ascontiguousarray(mask[0::2, 0::2])
    np.savez_compressed(os.path.join(OUT, "teddy_quarter.npz"), left=q0, right=q1, gt=gtq, mask=maskq)
    write_pfm_py(os.path.join(OUT, "teddy_quarter_disp0GT.pfm"), gtq)
    with open(REF + "/data/MiddEval3/trainingH/Teddy/calib.txt") as f:
        open(os.path.join(OUT, "teddy_calib.txt"), "w").write(f.read())
    print("teddy quarter", q0.shape, gtq.shape, "finite gt", int(np.isfinite(gtq).sum()))
    # a raw crop of the reference's own PFM bytes' decoding: top-left 32x24 of disp0GT as the SDK returns it
    np.save(os.path.join(OUT, "teddy_disp0GT_crop.npy"), gt[100:124, 200:232])
    write_pfm_py(os.path.join(OUT, "teddy_disp0GT_crop.pfm"), gt[100:124, 200:232])


if __name__ == "__main__":
    main()
