"""Synthetic Middlebury-shaped stereo pairs ("Synthetic generator").

Used by bench.py and the tests: there is no network for the real datasets and
the reference tree does not exist on the GPU box.  Deterministic for a given
(seed, shape, max_disparity) under numpy's PCG64.

Left image: i.i.d. uniform[1,255] per channel, 3x3 box low-pass, re-quantised
(never (0,0,0), so no pixel counts as "black", BlockSearch.cpp:41).  Ground
truth: piecewise-constant disparity bands in [D/8, 3D/4].  Right(y, x - d) =
Left(y, x); holes are filled with independent noise.
"""
import numpy as np

# Middlebury trainingH shapes (W, H, ndisp) from each scene's calib.txt
TRAINING_H = [
    ("Adirondack", 1436, 992, 145), ("ArtL", 694, 554, 128), ("Jadeplant", 1318, 994, 320),
    ("Motorcycle", 1482, 994, 140), ("MotorcycleE", 1482, 994, 140), ("Piano", 1414, 962, 130),
    ("PianoL", 1414, 962, 130), ("Pipes", 1470, 970, 150), ("Playroom", 1398, 952, 165),
    ("Playtable", 1360, 926, 145), ("PlaytableP", 1362, 924, 145), ("Recycle", 1440, 972, 130),
    ("Shelves", 1476, 994, 120), ("Teddy", 900, 750, 128), ("Vintage", 1444, 960, 380),
]


def _box3(a):
    p = np.pad(a.astype(np.uint16), ((1, 1), (1, 1), (0, 0)), mode="edge")
    s = sum(p[i:i + a.shape[0], j:j + a.shape[1]] for i in range(3) for j in range(3))
    return np.maximum((s + 4) // 9, 1).astype(np.uint8)


def make_pair(width, height, max_disparity, seed, right_width=None, right_height=None):
    """Return (left, right, gt) with left/right H x W x 3 uint8 (BGR) and gt int32 H x W."""
    rng = np.random.Generator(np.random.PCG64(seed))
    left = _box3(rng.integers(1, 256, size=(height, width, 3), dtype=np.uint8))
    lo = max(1, max_disparity // 8)
    hi = max(lo + 1, (3 * max_disparity) // 4)
    # piecewise-constant field: horizontal bands x vertical stripes
    nby, nbx = max(1, height // 96), max(1, width // 160)
    table = rng.integers(lo, hi, size=(nby + 1, nbx + 1), dtype=np.int32)
    yy = np.minimum(np.arange(height) * (nby + 1) // max(height, 1), nby)
    xx = np.minimum(np.arange(width) * (nbx + 1) // max(width, 1), nbx)
    gt = table[np.ix_(yy, xx)]
    rw = width if right_width is None else right_width
    rh = height if right_height is None else right_height
    right = _box3(rng.integers(1, 256, size=(rh, rw, 3), dtype=np.uint8))  # hole filler
    ys, xs = np.mgrid[0:height, 0:width]
    tx = xs - gt
    ok = (tx >= 0) & (tx < rw) & (ys < rh)
    # far-to-near so that nearer (larger d) surfaces win where they overlap
    order = np.argsort(gt[ok], kind="stable")
    right[ys[ok][order], tx[ok][order]] = left[ys[ok][order], xs[ok][order]]
    return left, right, gt
