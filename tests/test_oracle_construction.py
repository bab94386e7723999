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
