"""By-construction known answers for the CPU oracle (SURVEY.md 8c "what pins the restatement").

The reference has no test for WindowSearch, so the oracle (oracle/ws_oracle.c, a restatement of
BlockSearch.cpp:24-179 / LinearSearch.cpp:10-59) is pinned here by inputs whose answer follows
from the reference's rules alone, and by the independent NumPy brute force (oracle/brute.py).
"""
import numpy as np
import pytest

from oracle import brute


def textured(h, w, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(1, 256, size=(h, w, 3), dtype=np.uint8)


@pytest.mark.parametrize("cost", ["ssd", "sad"])
@pytest.mark.parametrize("d0", [1, 7, 13])
def test_left_shifted_copy_gives_constant_disparity(oracle, cost, d0):
    # right(y, x - d0) = left(y, x)  =>  every interior pixel with a full candidate set answers d0
    h, w, bs, maxd = 24, 80, 5, 16
    left = textured(h, w, 1)
    right = textured(h, w, 2)
    right[:, : w - d0] = left[:, d0:]
    out = oracle.block_left(left, right, bs, 0, maxd, cost=cost)
    half = (bs - 1) // 2
    inner = out[half:h - half, half + d0: w - half]
    assert (inner == d0).all()


@pytest.mark.parametrize("cost", ["ssd", "sad"])
def test_right_shifted_copy_gives_constant_disparity(oracle, cost):
    h, w, bs, maxd, d0 = 20, 70, 7, 12, 5
    left = textured(h, w, 3)
    right = textured(h, w, 4)
    right[:, : w - d0] = left[:, d0:]
    out = oracle.block_right(left, right, bs, 0, maxd, cost=cost)
    half = (bs - 1) // 2
    # interior pixels whose window at cx = x + d0 still fits
    assert (out[half:h - half - 1, half: w - d0 - half - 1] == d0).all()


def test_constant_images_show_the_tie_break(oracle):
    # all costs equal: left keeps the first candidate it tries = the largest valid d
    # (BlockSearch.cpp:53,76), right/linear keep the smallest (BlockSearch.cpp:147,168)
    h, w, bs, maxd = 12, 40, 5, 9
    img = np.full((h, w, 3), 77, dtype=np.uint8)
    half = (bs - 1) // 2
    out = oracle.block_left(img, img, bs, 0, maxd)
    for x in range(half, w - half):
        valid = [d for d in range(1, maxd + 1) if half <= x - d < w - half]
        want = max(valid) if valid else x
        assert (out[half:h - half, x] == want).all(), x
    assert (out[:half] == 0).all() and (out[h - half:] == 0).all()
    assert (out[:, :half] == 0).all() and (out[:, w - half:] == 0).all()
    outr = oracle.block_right(img, img, bs, 2, maxd)
    # every pixel with at least one candidate answers minDisparity = 2
    assert (outr[:, : w - 2 - half - 1] == 2).all()
    assert (oracle.linear(img, img, search_range=9) == 0).all()


def test_left_column_half_has_no_candidate_and_stores_x(oracle):
    # x = half: every cx = x - d < half is skipped, minimumCorrespondX stays 0 -> stores x
    left, right = textured(16, 30, 5), textured(16, 30, 6)
    out = oracle.block_left(left, right, 7, 0, 10)
    assert (out[3:13, 3] == 3).all()


def test_right_without_candidates_stores_minus_x(oracle):
    # at the right edge cx + right >= w1 for every d -> the loop breaks at once -> stores -x
    left, right = textured(14, 20, 7), textured(14, 32, 8)
    out = oracle.block_right(left, right, 5, 0, 6)
    for x in range(20, 32):
        assert (out[:14, x] == -x).all()


def test_black_pixels_stay_zero(oracle):
    left, right = textured(16, 40, 9), textured(16, 40, 10)
    left[8, 20] = 0
    right[5, 11] = 0
    assert oracle.block_left(left, right, 5, 0, 8)[8, 20] == 0
    assert oracle.block_right(left, right, 5, 0, 8)[5, 11] == 0
    # LinearSearch tests the LEFT pixel but matches the right one (LinearSearch.cpp:24,30)
    assert oracle.linear(left, right, search_range=8)[8, 20] == 0


def test_even_block_size_is_a_geometry_error_in_the_left_view(oracle):
    left, right = textured(16, 40, 11), textured(16, 40, 12)
    with pytest.raises(oracle.OracleGeometryError):
        oracle.block_left(left, right, 6, 0, 8)
    # the right view has no such problem: window (bs-1)/2*2
    assert np.array_equal(oracle.block_right(left, right, 6, 0, 8),
                          oracle.block_right(left, right, 5, 0, 8))


def test_right_view_with_short_left_image_is_a_geometry_error(oracle):
    left, right = textured(16, 40, 13), textured(20, 40, 14)
    with pytest.raises(oracle.OracleGeometryError):
        oracle.block_right(left, right, 5, 0, 8)


def test_rows_past_the_shorter_image_stay_zero(oracle):
    left, right = textured(20, 40, 15), textured(14, 40, 16)
    out = oracle.block_left(left, right, 5, 0, 8)
    assert out.shape == (20, 40) and (out[12:] == 0).all() and (out[2:12, 2:38] != 0).any()


@pytest.mark.parametrize("seed", range(12))
def test_oracle_equals_independent_brute_force(oracle, seed):
    rng = np.random.default_rng(100 + seed)
    h1, w1 = int(rng.integers(8, 28)), int(rng.integers(14, 48))
    h2, w2 = (h1, w1) if seed % 3 else (h1 + int(rng.integers(0, 4)), w1 + int(rng.integers(-5, 6)))
    levels = [256, 2, 4][seed % 3]          # few grey levels force ties
    scale = 1 if levels == 256 else 255 // (levels - 1)
    left = (rng.integers(0, levels, size=(h1, w1, 3)) * scale).astype(np.uint8)
    right = (rng.integers(0, levels, size=(h2, w2, 3)) * scale).astype(np.uint8)
    left[h1 // 2, w1 // 2] = 0
    bs = int(rng.choice([1, 3, 5, 7, 9]))
    maxd, mind = int(rng.integers(1, 20)), int(rng.integers(0, 3))
    for cost in ("ssd", "sad"):
        assert np.array_equal(oracle.block_left(left, right, bs, mind, maxd, cost=cost),
                              brute.block_left(left, right, bs, mind, maxd, cost))
    if h1 >= h2:
        for cost in ("ssd", "sad"):
            assert np.array_equal(oracle.block_right(left, right, bs, mind, maxd, cost=cost),
                                  brute.block_right(left, right, bs, mind, maxd, cost))
    assert np.array_equal(oracle.linear(left, right, search_range=11), brute.linear(left, right, 11))


def test_smooth_factor_raster_dependency_matches_literal_python(oracle):
    rng = np.random.default_rng(5)
    left = (rng.integers(0, 4, size=(10, 18, 3)) * 85).astype(np.uint8)
    right = (rng.integers(0, 4, size=(10, 18, 3)) * 85).astype(np.uint8)
    for s in (0.9, 0.5):
        assert np.array_equal(oracle.block_left(left, right, 3, 0, 6, smooth=s),
                              brute.block_left_smooth_py(left, right, 3, 6, s))
    # smooth == 1.0 is the identity: the data-parallel case the device path covers
    assert np.array_equal(oracle.block_left(left, right, 3, 0, 6, smooth=1.0),
                          brute.block_left(left, right, 3, 0, 6))


def test_row_band_and_threads_do_not_change_results(oracle):
    left, right = textured(40, 60, 21), textured(40, 60, 22)
    full = oracle.block_left(left, right, 5, 0, 16)
    band = oracle.block_left(left, right, 5, 0, 16, rows=(10, 20), threads=4)
    assert np.array_equal(band[10:20], full[10:20]) and (band[:10] == 0).all() and (band[20:] == 0).all()
    assert np.array_equal(oracle.block_right(left, right, 5, 0, 16, threads=4),
                          oracle.block_right(left, right, 5, 0, 16))


def test_subpixel_is_exact_on_a_parabolic_cost(oracle):
    # refined value = d + (C[d-1]-C[d+1]) / (2 (C[d-1] - 2C[d] + C[d+1])), within half a pixel
    left, right = textured(20, 60, 31), textured(20, 60, 32)
    a = oracle.block_left(left, right, 5, 0, 16)
    b = oracle.block_left(left, right, 5, 0, 16, subpixel=True)
    assert np.abs(a - b).max() <= 0.5 and (a != b).any()
    assert np.array_equal(np.round(b - (b - a)), a)


# ---- second witnesses for the branches the C restatement had alone ---------------
def few_levels(h, w, levels, seed):
    rng = np.random.default_rng(seed)
    return (rng.integers(0, levels, size=(h, w, 3)) * (255 // (levels - 1))).astype(np.uint8)


@pytest.mark.parametrize("smooth", [0.0, 0.9, 1.7, -0.5, float("inf")])
@pytest.mark.parametrize("levels", [256, 3, 2])
def test_right_view_smooth_factor_matches_literal_python(oracle, smooth, levels):
    """Right view with smoothFactor (BlockSearch.cpp:160-165): tie-heavy images, minDisparity 0 and > 0,
    clipped border windows, the -x fallback, exact d = 0 matches (0 * inf = NaN must refuse d = 0)."""
    if levels == 256:
        left, right = textured(11, 22, 40), textured(11, 20, 41)
    else:
        left, right = few_levels(11, 22, levels, 42), few_levels(11, 20, levels, 43)
    right[2:6, 3:9] = left[2:6, 3:9]          # exact matches at d = 0
    right[7, 5] = 0
    for bs, mind, maxd, cost in ((5, 0, 6, "ssd"), (3, 0, 4, "sad"), (7, 2, 7, "ssd"), (2, 0, 3, "ssd"), (5, 0, 1, "sad")):
        want, _ = brute.block_right_py(left, right, bs, mind, maxd, smooth, cost)
        got = oracle.block_right(left, right, bs, mind, maxd, smooth=smooth, cost=cost)
        assert np.array_equal(got, want), (bs, mind, maxd, cost)


@pytest.mark.parametrize("smooth", [0.0, 0.9, 1.7, -0.5, float("inf")])
def test_linear_search_smooth_factor_matches_literal_python(oracle, smooth):
    """LinearSearch with smoothFactor (LinearSearch.cpp:39-44), few grey levels (ties, exact zeros)."""
    for levels, seed in ((256, 50), (3, 51), (2, 52)):
        left = textured(9, 26, seed) if levels == 256 else few_levels(9, 26, levels, seed)
        right = textured(10, 24, seed + 100) if levels == 256 else few_levels(10, 24, levels, seed + 100)
        right[3:5, 4:12] = left[3:5, 4:12]
        left[6, 7] = 0
        for rng_ in (1, 5, 200):
            want = brute.linear_py(left, right, smooth, rng_)
            assert np.array_equal(oracle.linear(left, right, smooth=smooth, search_range=rng_), want), (levels, rng_)


@pytest.mark.parametrize("case", [(3, 40.0, "ssd", 0, 1.0), (5, 19.0, "ssd", 0, 0.9), (3, 150.0, "sad", 1, 1.0),
                                  (7, 10.0, "ssd", 0, 1.7), (5, 1e9, "ssd", 0, 1.0)])
def test_var_block_matches_literal_python(oracle, case):
    """varBlock (BlockSearch.cpp:125-145): windows grow by 4 while the centred norm is below thres; the
    search then uses the grown, re-clipped window; "max block size" (:177); thres = 1e9 grows every window
    until it covers the image (the reference would spin forever; growth is capped where nothing changes)."""
    bs, thres, cost, mind, smooth = case
    left, right = textured(13, 24, 60), textured(13, 22, 61)
    right[2:9, 4:14] = (right[2:9, 4:14] // 64) * 64       # weak texture
    right[5:8, 15:20] = 90                                  # none
    right[0, 0] = 0
    want, want_mb = brute.block_right_py(left, right, bs, mind, 6, smooth, cost, var_block=True, thres=thres)
    got, mb = oracle.block_right(left, right, bs, mind, 6, smooth=smooth, var_block=True, thres=thres, cost=cost,
                                 return_max_block=True)
    assert np.array_equal(got, want)
    assert mb == want_mb and (mb > bs or thres < 20)


def flat_window_where_float32_ties(side=261):
    """side x side (odd count n > 65536), grey 200 with a centred block of (n - 1) / 2 pixels of 201:
    mean = 200.5 - 1/(2n), less than half a float ulp (2^-17) below 200.5."""
    n = side * side
    assert n % 2 == 1 and 1.0 / (2 * n) < 2.0 ** -17
    img = np.full((side, side), 200, dtype=np.uint8)
    k = (n - 1) // 2
    b = int(np.sqrt(k))                     # a centred b x b block of 201s, the rest of them in a ring row
    lo = (side - b) // 2
    img[lo:lo + b, lo:lo + b] = 201
    rest = k - b * b
    ring = [(lo - 1, x) for x in range(lo - 1, lo + b + 1)] + [(lo + b, x) for x in range(lo - 1, lo + b + 1)] + \
           [(y, lo - 1) for y in range(lo, lo + b)] + [(y, lo + b) for y in range(lo, lo + b)]
    for y, x in ring[:rest]:
        img[y, x] = 201
    assert int((img == 201).sum()) == k
    return np.repeat(img[:, :, None], 3, axis=2)


def test_centred_norm_is_the_float32_one(oracle):
    """cv::subtract(Mat_u8, Scalar) with a non-integer Scalar runs in float32 (OpenCV 4.x arithm_op), not
    double.  The two differ exactly when the float mean is k + .5 and the double mean is not -- the mean's
    fraction within half a float ulp (2^-17 for means in [128, 256)) of .5 -- which needs more than 65536
    pixels: a varBlock window grown past 256 x 256 on a near-flat region.  Such a window: 261 x 261, grey
    200 with (n - 1) / 2 pixels of 201 -> mean = 200.5 - 1/(2n); as a float that IS 200.5, the 201s tie
    (0.5 -> 0, half to even) and the norm is 0; in double 201 - mean > 0.5 rounds to 1."""
    img = flat_window_where_float32_ties()
    n = img.shape[0] * img.shape[1]
    f32, f64 = brute.centred_norm_f32(img), brute.centred_norm_f64(img)
    assert f32 == 0.0 and f64 == np.sqrt(3.0 * ((n - 1) // 2))
    assert oracle.centred_norm(img, 0, 0, img.shape[1], img.shape[0]) == f32   # the oracle follows OpenCV's float32
    # small windows (every window the tests and configs use without varBlock growth): no difference
    rng = np.random.default_rng(2)
    for _ in range(200):
        h, w = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        win = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) if rng.random() < 0.5 else \
            (rng.integers(0, 2, size=(h, w, 3)) + int(rng.integers(0, 255))).astype(np.uint8)
        a = brute.centred_norm_f32(win)
        assert a == brute.centred_norm_f64(win) == oracle.centred_norm(win, 0, 0, w, h)


def float32_tie_scene():
    """A right image in which pixel (130, 130) with blockSize 3 grows to the window [0,261) x [0,261) whose
    mean is within half a float ulp of 200.5 (tests/test_oracle_construction.py); columns / rows from 261 on
    are textured, so the NEXT window (blockSize 267) has a large norm."""
    rng = np.random.default_rng(3)
    right = rng.integers(1, 256, size=(280, 300, 3), dtype=np.uint8)
    right[:261, :261] = flat_window_where_float32_ties(261)
    left = rng.integers(1, 256, size=(280, 340, 3), dtype=np.uint8)
    return left, right


def test_float32_texture_test_changes_the_answer_where_it_ties(oracle):
    """The deviation between OpenCV's float32 subtraction and a double one, end to end: pixel (130, 130) of
    float32_tie_scene grows to 267 (float32: the 261 x 261 window's norm is 0) or stops at 263 (double: norm
    319 >= thres), searches with different windows and finds different disparities.  The oracle is float32."""
    left, right = float32_tie_scene()
    a, mba = brute.block_right_py(left, right, 3, 0, 32, var_block=True, thres=10.0, only=(130, 130))
    b, mbb = brute.block_right_py(left, right, 3, 0, 32, var_block=True, thres=10.0, only=(130, 130),
                                  texture=brute.centred_norm_f64)
    assert (mba, mbb) == (267, 263) and a[130, 130] != b[130, 130]
    want, mb = oracle.block_right(left, right, 3, 0, 32, var_block=True, thres=10.0, rows=(130, 131), threads=8,
                                  return_max_block=True)
    assert want[130, 130] == a[130, 130] and mb >= 267
