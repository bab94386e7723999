"""removeDisparityOutliers (reconstruction.cpp:5-18) on the device: the 32-bit integer box filter that 8-bit maps take
(main.cpp:47-53 reads an 8-bit PNG) and the double one behind it, both against the NumPy restatement."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def run(ctx, oracle, m, k, tf=1.5, tb=0.8):
    got = ctx.remove_disparity_outliers(m, k, tf, tb)
    want = oracle.remove_disparity_outliers(m, k, tf, tb)
    return got, want, ctx.last_outliers_path()


def test_integer_maps_take_the_integer_kernels(gpu_ctx, oracle):
    rng = np.random.default_rng(5)
    for (w, h, k) in ((300, 220, 500), (300, 220, 31), (300, 220, 4), (1, 1, 3), (1, 40, 7), (40, 1, 7), (2, 2, 500),
                      (17, 9, 64), (257, 130, 129), (640, 480, 1), (1500, 1000, 500), (33, 2300, 11), (4100, 3, 9)):
        m = rng.integers(0, 256, size=(h, w)).astype(np.float32)
        m[rng.random((h, w)) < 0.05] = 0
        got, want, path = run(gpu_ctx, oracle, m, k)
        assert path == "integer", (w, h, k, path)
        assert np.array_equal(got, want), (w, h, k)
        assert (got != m).any() or k == 1 or w * h < 100, (w, h, k)


def test_random_small_shapes(gpu_ctx, oracle):
    rng = np.random.default_rng(11)
    for _ in range(150):
        w, h = int(rng.integers(1, 90)), int(rng.integers(1, 90))
        k = int(rng.choice([1, 2, 3, 5, 8, 30, 64, 177, 500, 1200]))
        hi = int(rng.choice([2, 256, 4096]))
        m = rng.integers(0, hi, size=(h, w)).astype(np.float32)
        tf, tb = float(rng.choice([1.5, 1.0, 2.0])), float(rng.choice([0.8, 1.0, 0.5]))
        got, want, path = run(gpu_ctx, oracle, m, k, tf, tb)
        fits = m.max() <= 255                                                    # what an 8-bit PNG holds
        assert path == ("integer" if fits else "integer-then-double"), (w, h, k, hi, path)
        assert np.array_equal(got, want), (w, h, k, hi)


def test_other_values_fall_back_to_the_double_kernels(gpu_ctx, oracle):
    rng = np.random.default_rng(7)
    base = rng.integers(0, 200, size=(120, 160)).astype(np.float32)
    cases = {}
    m = base.copy(); m[60, 80] = 17.25; cases["a quarter"] = m             # exact in double: still equal to the oracle
    m = base.copy(); m[0, 0] = -3.0; cases["a negative value"] = m
    m = base.copy(); m[119, 159] = 3.0e7; cases["above 2^24"] = m
    m = base * 0.5; cases["halves everywhere"] = m
    for name, m in cases.items():
        got, want, path = run(gpu_ctx, oracle, m, 31)
        assert path == "integer-then-double", (name, path)
        assert np.array_equal(got, want), name
    # the flag is per call: an integer map right after goes through the integer kernels again
    got, want, path = run(gpu_ctx, oracle, base, 31)
    assert path == "integer" and np.array_equal(got, want)
    # 16-bit disparities are integers too, but not what the packed intermediate carries
    m = base.copy(); m[5, 5] = 256.0
    got, want, path = run(gpu_ctx, oracle, m, 500)
    assert path == "integer-then-double" and np.array_equal(got, want)
    m[5, 5] = 255.0
    got, want, path = run(gpu_ctx, oracle, m, 500)
    assert path == "integer" and np.array_equal(got, want)
    got, want, path = run(gpu_ctx, oracle, m, 4001)                          # 4001 * 4001 * 255 could pass 2^32
    assert path == "double" and np.array_equal(got, want)
    got, want, path = run(gpu_ctx, oracle, m, 4000)
    assert path == "integer" and np.array_equal(got, want)
    # NaN is not an integer either; the double kernels then do what they always did with it
    m = base.copy(); m[3, 3] = np.nan
    gpu_ctx.remove_disparity_outliers(m, 5, 1.5, 0.8)
    assert gpu_ctx.last_outliers_path() == "integer-then-double"


def test_maps_beyond_the_integer_kernels_limits(gpu_ctx, oracle):
    """Columns too tall for the band's prefix in LDS: the double kernels (and their own direct form) take over."""
    rng = np.random.default_rng(9)
    m = rng.integers(0, 256, size=(9700, 12)).astype(np.float32)
    got, want, path = run(gpu_ctx, oracle, m, 15)
    assert path == "double"
    assert np.array_equal(got, want)


def test_strided_map_in_place(wslib, gpu_ctx, oracle):
    lib = wslib.load_library()
    rng = np.random.default_rng(3)
    m = rng.integers(0, 256, size=(200, 300)).astype(np.float32)
    padded = np.full((200, 320), 7.5, np.float32)
    padded[:, :300] = m
    assert lib.ws_remove_disparity_outliers(gpu_ctx._h, padded.ctypes.data, 300, 200, 320, 9, 2.0, 3.0) == 0
    assert gpu_ctx.last_outliers_path() == "integer"                         # the padding is not part of the map
    assert np.array_equal(padded[:, :300], oracle.remove_disparity_outliers(m, 9, 2.0, 3.0))
    assert (padded[:, 300:] == 7.5).all()
