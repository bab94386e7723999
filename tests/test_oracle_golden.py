"""The oracle against the golden fixtures cut from the reference's own data (tests/golden,
made by tools/make_golden.py with the independent NumPy brute force)."""
import numpy as np
import pytest

from conftest import golden_cases, load_golden


@pytest.mark.parametrize("name", golden_cases())
def test_oracle_reproduces_golden(oracle, name):
    g = load_golden(name)
    args = (g["left"], g["right"], g["block_size"], g["min_disparity"], g["max_disparity"])
    s = float(g.get("smooth", 1.0))            # round-2 fixtures (tools/make_golden_r2.py) carry smoothFactor / varBlock
    if g["view"] == "left":
        out = oracle.block_left(*args, smooth=s, cost=g["cost"])
    elif g["view"] == "right":
        out, mb = oracle.block_right(*args, smooth=s, var_block=bool(g.get("var_block", 0)), thres=float(g.get("thres", 19.0)),
                                     cost=g["cost"], return_max_block=True)
        if g.get("var_block", 0):
            assert mb == g["max_block"]
    else:
        out = oracle.linear(g["left"], g["right"], smooth=s, search_range=g["max_disparity"])
    assert np.array_equal(out, g["expected"].astype(np.float64))


def test_teddy_quarter_bad2_is_plausible(oracle):
    """Config 1 (plumbing): Teddy quarter-res, 5x5 SAD, D=64, scored with evaldisp."""
    g = load_golden("teddy_quarter")
    disp = oracle.block_left(g["left"], g["right"], 5, 0, 64, cost="sad", threads=4)
    res = oracle.evaldisp(disp, g["gt"], g["mask"], 2.0, 64.0)
    assert res["n"] > 100000
    assert res["invalid"] < 5.0
    assert 5.0 < res["bad"] < 60.0      # a 5x5 WTA block matcher: far from good, far from random
