"""Independent NumPy brute force used to pin the C restatement (oracle/ws_oracle.c).

TEST INFRASTRUCTURE ONLY.  Written separately from ws_oracle.c and on purpose
along a different route: a per-disparity difference plane, a 2-D summed-area
table, and vectorised window look-ups -- so that a slip in the C loops
(window geometry, validity, tie-break, fallback) does not repeat here.  Only
smoothFactor == 1.0 is vectorisable; `*_smooth_py` are literal pure-Python
loops for tiny images.

Semantics follow SURVEY.md 8(a) / Appendix A, i.e. the reference's
BlockSearch.cpp:24-179 and LinearSearch.cpp:10-59.
"""
import numpy as np


def _pixel_cost(a, b, cost):
    d = a.astype(np.int64) - b.astype(np.int64)
    return (np.abs(d) if cost == "sad" else d * d).sum(axis=2)


def _sat(p):
    """Summed-area table with a zero row/column in front: window sums by 4 look-ups."""
    s = np.zeros((p.shape[0] + 1, p.shape[1] + 1), dtype=np.int64)
    s[1:, 1:] = p.cumsum(0).cumsum(1)
    return s


def cost_volume_left(L, R, block_size, max_disparity, cost="ssd"):
    """C[d-1, y, x] for d = 1..max_disparity (int64, -1 where the candidate is invalid)."""
    h1, w1 = L.shape[:2]
    h2, w2 = R.shape[:2]
    height = min(h1, h2)
    half = (block_size - 1) // 2
    vol = np.full((max_disparity, h1, w1), -1, dtype=np.int64)
    ys = np.arange(half, height - half)
    for d in range(1, max_disparity + 1):
        # plane[y, x] = pixel cost between L(y,x) and R(y,x-d) for d <= x < min(w1, w2+d)
        x_lo, x_hi = d, min(w1, w2 + d)
        if x_hi <= x_lo:
            continue
        plane = _pixel_cost(L[:height, x_lo:x_hi], R[:height, x_lo - d:x_hi - d], cost)
        s = _sat(plane)
        xs = np.arange(half, w1 - half)
        cx = xs - d
        ok = (cx >= half) & (cx < w2 - half)
        xs = xs[ok]
        if xs.size == 0 or ys.size == 0:
            continue
        a = xs - half - x_lo            # first window column inside `plane`
        b = a + block_size
        top = (ys - half)[:, None]
        bot = top + block_size
        win = s[bot, b[None, :]] - s[top, b[None, :]] - s[bot, a[None, :]] + s[top, a[None, :]]
        vol[d - 1][np.ix_(ys, xs)] = win
    return vol


def block_left(L, R, block_size, min_disparity, max_disparity, cost="ssd"):
    """Left view, smoothFactor 1.0.  Ties go to the largest d; no candidate -> x."""
    del min_disparity  # ignored by the reference's left view
    h1, w1 = L.shape[:2]
    height = min(h1, R.shape[0])
    half = (block_size - 1) // 2
    vol = cost_volume_left(L, R, block_size, max_disparity, cost)
    big = np.iinfo(np.int64).max
    c = np.where(vol < 0, big, vol)
    # largest d on ties == first minimum when d runs downwards
    rev = c[::-1]
    arg = rev.argmin(axis=0)
    dmap = (max_disparity - arg).astype(np.float64)
    none = rev.min(axis=0) == big
    xs = np.arange(w1, dtype=np.float64)[None, :].repeat(h1, 0)
    dmap[none] = xs[none]
    out = np.zeros((h1, w1), dtype=np.float64)
    interior = np.zeros((h1, w1), dtype=bool)
    interior[half:height - half, half:w1 - half] = True
    black = (L == 0).all(axis=2)
    sel = interior & ~black
    out[sel] = dmap[sel]
    return out


def block_right(L, R, block_size, min_disparity, max_disparity, cost="ssd"):
    """Right view, smoothFactor 1.0, varBlock off, min_disparity >= 0.
    Ties go to the smallest d; no candidate -> -x."""
    assert min_disparity >= 0
    h1, w1 = L.shape[:2]
    h2, w2 = R.shape[:2]
    height = min(h1, h2)
    half = (block_size - 1) // 2
    ys, xs = np.mgrid[0:height, 0:w2]
    left = np.minimum(xs, half)
    right = np.minimum(w2 - xs - 1, half)
    up = np.minimum(ys, half)
    down = np.minimum(h2 - ys - 1, half)
    overrun = ys + down > h1            # leftImage_(Rect) would leave the image there
    nonblack = ~(R[:height] == 0).all(axis=2)
    area = (left + right) * (up + down)
    big = np.iinfo(np.int64).max
    best = np.full((height, w2), big, dtype=np.int64)
    bestd = np.full((height, w2), -1, dtype=np.int64)
    rows = min(h1, h2)
    for d in range(min_disparity, max_disparity):
        # plane[y, xr] = pixel cost between L(y, xr+d) and R(y, xr) for xr < min(w2, w1-d)
        n = min(w2, w1 - d)
        if n <= 0:
            break
        plane = np.zeros((rows, w2), dtype=np.int64)
        plane[:, :n] = _pixel_cost(L[:rows, d:d + n], R[:rows, :n], cost)
        s = _sat(plane)
        valid = (xs + d + right < w1)
        if (valid & nonblack & overrun).any():
            raise ValueError("reference would throw (left ROI below the image)")
        valid &= area > 0
        y0 = ys - up
        y1 = np.minimum(ys + down, rows)
        x0 = xs - left
        x1 = xs + right
        win = s[y1, x1] - s[y0, x1] - s[y1, x0] + s[y0, x0]
        upd = valid & (win < best)      # strict <: earlier (smaller) d keeps ties
        best[upd] = win[upd]
        bestd[upd] = d
    dmap = np.where(bestd >= 0, bestd, -xs).astype(np.float64)
    out = np.zeros((h2, w2), dtype=np.float64)
    black = (R[:height] == 0).all(axis=2)
    out[:height][~black] = dmap[~black]
    return out


def linear(L, R, search_range=200):
    """LinearSearch, smoothFactor 1.0, with the build's defined out-of-bounds rule."""
    h1, w1 = L.shape[:2]
    h2, w2 = R.shape[:2]
    rows = min(h1, h2)
    big = np.iinfo(np.int64).max
    best = np.full((rows, w2), big, dtype=np.int64)
    bestk = np.zeros((rows, w2), dtype=np.int64)
    js = np.arange(w2)
    for d in range(search_range):
        n = min(w2, w1 - d)
        if n <= 0:
            break
        c = np.full((rows, w2), big, dtype=np.int64)
        c[:, :n] = _pixel_cost(R[:rows, :n], L[:rows, d:d + n], "ssd")
        upd = c < best
        best[upd] = c[upd]
        bestk[upd] = (js[None, :] + d).repeat(rows, 0)[upd]
    out = np.zeros((h2, w2), dtype=np.float64)
    dmap = (bestk - js[None, :]).astype(np.float64)
    black = np.zeros((rows, w2), dtype=bool)
    m = min(w1, w2)
    black[:, :m] = (L[:rows, :m] == 0).all(axis=2)
    out[:rows][~black] = dmap[~black]
    return out


def block_left_smooth_py(L, R, block_size, max_disparity, smooth, cost="ssd"):
    """Literal raster-order loops with the smoothFactor dependency (tiny images only)."""
    h1, w1 = L.shape[:2]
    h2, w2 = R.shape[:2]
    height = min(h1, h2)
    half = (block_size - 1) // 2
    Li = L.astype(np.int64)
    Ri = R.astype(np.int64)
    out = np.zeros((h1, w1), dtype=np.float64)
    for y in range(half, height - half):
        for x in range(half, w1 - half):
            if not L[y, x].any():
                continue
            lw = Li[y - half:y + half + 1, x - half:x + half + 1]
            best, best_cx = np.finfo(np.float64).max, 0
            for cx in range(x - max_disparity, x):
                if cx < half or cx >= w2 - half:
                    continue
                rw = Ri[y - half:y + half + 1, cx - half:cx + half + 1]
                diff = np.abs(lw - rw)
                dist = float(diff.sum()) if cost == "sad" else float(np.sqrt(np.float64((diff * diff).sum())))
                if y >= 1 and out[y - 1, x] == float(x - cx):
                    dist *= smooth
                if x >= 1 and out[y, x - 1] == float(x - cx):
                    dist *= smooth
                if dist < best:
                    best, best_cx = dist, cx
            out[y, x] = float(x - best_cx)
    return out


# ---- second witnesses for the branches ws_oracle.c had alone: right-view smoothFactor, LinearSearch
# ---- smoothFactor, varBlock.  Literal raster-order loops on NumPy windows (tiny images only).
def centred_norm_f32(win):
    """cv::mean + cv::subtract(window, mean, s) + cv::norm(s, NORM_L2) on a CV_8UC3 window
    (BlockSearch.cpp:125-129), as OpenCV 4.x evaluates them (un-vendored: restated).
      mean     : per-channel double, exact integer sum / pixel count;
      subtract : u8 Mat minus a non-integer Scalar works in FLOAT32 -- the Scalar is narrowed to float,
                 the pixel widened to float, the difference rounded half-to-even and saturated to u8
                 (arithm_op: depth2 = CV_32F for a u8 source, wtype = CV_32F; cvt32f8u = cvRound);
                 an all-integer mean takes the u8 saturating path, which gives the same bytes;
      norm     : sqrt of the exact integer sum of squares."""
    if win.shape[0] == 0 or win.shape[1] == 0:
        return 0.0
    n = win.shape[0] * win.shape[1]
    mean = win.reshape(-1, 3).astype(np.int64).sum(axis=0).astype(np.float64) / np.float64(n)
    diff = win.astype(np.float32) - mean.astype(np.float32)[None, None, :]
    s = np.clip(np.rint(diff), 0, 255).astype(np.int64)
    return float(np.sqrt(np.float64((s * s).sum())))


def centred_norm_f64(win):
    """The same with the subtraction in float64 -- NOT what OpenCV does; kept to show where the two part."""
    if win.shape[0] == 0 or win.shape[1] == 0:
        return 0.0
    n = win.shape[0] * win.shape[1]
    mean = win.reshape(-1, 3).astype(np.int64).sum(axis=0).astype(np.float64) / np.float64(n)
    s = np.clip(np.rint(win.astype(np.float64) - mean[None, None, :]), 0, 255).astype(np.int64)
    return float(np.sqrt(np.float64((s * s).sum())))


def _norm(a, b, cost):
    d = np.abs(a.astype(np.int64) - b.astype(np.int64))
    return float(d.sum()) if cost == "sad" else float(np.sqrt(np.float64((d * d).sum())))


def block_right_py(L, R, block_size, min_disparity, max_disparity, smooth=1.0, cost="ssd",
                   var_block=False, thres=19.0, max_growth=1 << 20, only=None, texture=None):
    """computeDisparityMapRight (BlockSearch.cpp:88-179) with smoothFactor and varBlock, literally.
    Returns (map, max block size).  Raises ValueError where the reference would throw.
    only = (y, x): just that pixel (smoothFactor 1 only); texture: the varBlock norm (default: float32)."""
    texture = texture or centred_norm_f32
    h1, w1 = L.shape[:2]
    h2, w2 = R.shape[:2]
    height = min(h1, h2)
    out = np.zeros((h2, w2), dtype=np.float64)
    max_block = block_size
    fmax = np.finfo(np.float64).max
    for y in range(height):
        for x in range(w2):
            if only is not None and (y, x) != tuple(only):
                continue
            if not R[y, x].any():
                continue
            bs = block_size

            def clip(b):
                hb = (b - 1) // 2
                return min(x, hb), min(w2 - x - 1, hb), min(y, hb), min(h2 - y - 1, hb)

            left, right, up, down = clip(bs)
            if var_block:
                for _ in range(max_growth):
                    if not texture(R[y - up:y + down, x - left:x + right]) < thres:
                        break
                    grown = clip(bs + 4)
                    bs += 4
                    if grown == (left, right, up, down):
                        break       # the reference would spin forever: growth is capped where nothing changes
                    left, right, up, down = grown
            max_block = max(max_block, bs)
            rw = R[y - up:y + down, x - left:x + right]
            area = (left + right) * (up + down)
            best, best_cx = fmax, 0
            for cx in range(x + min_disparity, x + max_disparity):
                if cx + right >= w1:
                    break
                if cx - left < 0 or y + down > h1:
                    raise ValueError("reference would throw (left ROI outside the image)")
                lw = L[y - up:y + down, cx - left:cx + right]
                with np.errstate(invalid="ignore", divide="ignore"):
                    dist = np.float64(_norm(lw, rw, cost)) / np.float64(area)      # 0/0 -> NaN never wins
                    if y >= 1 and out[y - 1, x] == float(x - cx):
                        dist = dist * np.float64(smooth)
                    if x >= 1 and out[y, x - 1] == float(x - cx):
                        dist = dist * np.float64(smooth)
                if dist < best:
                    best, best_cx = dist, cx
            out[y, x] = float(best_cx - x)
    return out, max_block


def linear_py(L, R, smooth=1.0, search_range=200):
    """LinearSearch::computeDisparityMap (LinearSearch.cpp:10-59) with smoothFactor, literally; candidates
    past the left row's end are skipped (the build's definition of the reference's out-of-bounds read)."""
    h1, w1 = L.shape[:2]
    h2, w2 = R.shape[:2]
    out = np.zeros((h2, w2), dtype=np.float64)
    fmax = np.finfo(np.float64).max
    Li, Ri = L.astype(np.int64), R.astype(np.int64)
    for i in range(min(h1, h2)):
        for j in range(w2):
            if j < w1 and not L[i, j].any():
                continue
            best, col = fmax, 0
            for k in range(j, j + search_range):
                if k >= w1:
                    break
                d = Ri[i, j] - Li[i, k]
                with np.errstate(invalid="ignore"):
                    dist = np.sqrt(np.float64((d * d).sum()))
                    if i >= 1 and out[i - 1, j] == float(j - k):
                        dist = dist * np.float64(smooth)
                    if j >= 1 and out[i, j - 1] == float(j - k):
                        dist = dist * np.float64(smooth)
                if dist < best:
                    best, col = dist, k
            out[i, j] = float(col - j)
    return out


def evaldisp_np(disp, gt, mask, badthresh, maxdisp, rounddisp=0):
    """evaldisp (utils.cpp:123-168) vectorised: a third implementation beside the C oracle's loop and the
    library's (those two follow the reference's text line by line and so read alike).  float32 throughout,
    the error sum accumulated in float32 in row-major order like the reference's `serr += err`."""
    d = np.asarray(disp, dtype=np.float32)
    g = np.asarray(gt, dtype=np.float32)
    m = np.asarray(mask)
    known = g != np.float32(np.inf)                      # :137
    valid = d != 0                                       # :140
    dd = np.where(valid, np.maximum(np.float32(0), np.minimum(np.float32(maxdisp), d)), d)   # :142
    if rounddisp:
        dd = np.where(valid, np.floor(np.abs(dd) + np.float32(0.5)) * np.sign(dd), dd).astype(np.float32)    # round(): half away from zero
    err = np.abs(dd - np.where(known, g, np.float32(0))).astype(np.float32)
    counted = known & (m == 255)                         # :146
    n = int(counted.sum())
    bad = int((counted & valid & (err > np.float32(badthresh))).sum())
    invalid = int((counted & ~valid).sum())
    serr = np.float32(0)
    for e in err[counted & valid].ravel():               # sequential float32 sum, as the reference accumulates
        serr = np.float32(serr + e)
    return {"n": n, "bad": float(np.float32(100.0 * bad / n)), "invalid": float(np.float32(100.0 * invalid / n)),
            "total_bad": float(np.float32(100.0 * (bad + invalid) / n)), "avg_err": float(np.float32(serr / np.float32(n - invalid))),
            "valid": 100.0 * n / (d.shape[0] * d.shape[1])}
