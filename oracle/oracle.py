"""ctypes front end of the CPU oracle (oracle/ws_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the product package.  PARITY UNPINNED by
the reference's own tests; see ws_oracle.h.

The functions mirror the reference's call surface:
  block_left   ~ BlockSearch(L,R,bs,minD,maxD).computeDisparityMapLeft(s)    (BlockSearch.cpp:24-86)
  block_right  ~ ...computeDisparityMapRight(s, varBlock, thres)             (BlockSearch.cpp:88-179)
  linear       ~ LinearSearch(L,R).computeDisparityMap(s)                    (LinearSearch.cpp:10-59)
  evaldisp     ~ evaldisp(disp, gt, mask, badthresh, maxdisp, rounddisp)     (utils.cpp:123-168)
Images are H x W x 3 uint8 arrays (BGR), outputs float64 maps.
"""
import ctypes
import hashlib
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libws_oracle.so")

COST = {"ssd": 0, "sad": 1}


class OracleGeometryError(ValueError):
    """The reference would throw a cv::Exception for these arguments."""


class _Image(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("width", ctypes.c_int),
                ("height", ctypes.c_int), ("stride", ctypes.c_int)]


def _host_signature():
    """What -march=native depends on: the CPU model and its feature flags."""
    model, flags = "", ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and not model:
                model = line.split(":", 1)[1].strip()
            elif line.startswith("flags") and not flags:
                flags = " ".join(sorted(line.split(":", 1)[1].split()))
            if model and flags:
                break
    except OSError:
        pass
    return hashlib.sha256((model + "|" + flags).encode()).hexdigest()


def build(force=False):
    """Compile libws_oracle.so with the Makefile next to this file.  The library is built with
    -march=native, so it is rebuilt when it is older than its sources OR was built on another CPU
    (the .so travels to the GPU box with the tree; its build host's signature sits beside it)."""
    src = [os.path.join(_HERE, "ws_oracle.c"), os.path.join(_HERE, "ws_oracle.h"), os.path.join(_HERE, "Makefile")]
    sig_path, sig = _LIB_PATH + ".host", _host_signature()
    try:
        built_on = open(sig_path).read().strip()
    except OSError:
        built_on = ""
    lib_override = os.environ.get("WS_ORACLE_LIB")          # e.g. the sanitizer build (oracle/Makefile: asan)
    if lib_override:
        return lib_override
    if (force or not os.path.exists(_LIB_PATH) or built_on != sig
            or any(os.path.getmtime(_LIB_PATH) < os.path.getmtime(p) for p in src)):
        tmp = "libws_oracle.so.tmp.%d" % os.getpid()       # aside + rename: no loader ever sees half a file
        try:
            subprocess.check_call(["make", "-s", "-B", "-C", _HERE, "OUT=" + tmp, tmp])
            os.replace(os.path.join(_HERE, tmp), _LIB_PATH)
            with open(sig_path, "w") as f:
                f.write(sig + "\n")
        finally:
            if os.path.exists(os.path.join(_HERE, tmp)):
                os.remove(os.path.join(_HERE, tmp))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        P = ctypes.POINTER
        _lib.wso_set_threads.argtypes = [ctypes.c_int]
        _lib.wso_set_threads.restype = None
        _lib.wso_block_left.argtypes = [P(_Image), P(_Image)] + [ctypes.c_int] * 3 + [
            ctypes.c_double] + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int]
        _lib.wso_block_right.argtypes = [P(_Image), P(_Image)] + [ctypes.c_int] * 3 + [
            ctypes.c_double, ctypes.c_int, ctypes.c_double] + [ctypes.c_int] * 4 + [
            ctypes.c_void_p, ctypes.c_int, P(ctypes.c_int)]
        _lib.wso_linear.argtypes = [P(_Image), P(_Image), ctypes.c_int, ctypes.c_double,
                                    ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        _lib.wso_centred_norm.argtypes = [P(_Image)] + [ctypes.c_int] * 4
        _lib.wso_centred_norm.restype = ctypes.c_double
        _lib.wso_evaldisp.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 2 + [
            ctypes.c_float, ctypes.c_float, ctypes.c_int, P(ctypes.c_double)]
    return _lib


def _img(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("expected an H x W x 3 uint8 image")
    return a, _Image(a.ctypes.data, a.shape[1], a.shape[0], a.strides[0])


def _check(rc):
    if rc == -2:
        raise OracleGeometryError("reference would throw cv::Exception (ROI outside image)")
    if rc != 0:
        raise ValueError("oracle rejected the arguments (code %d)" % rc)


def _rows(rows, h):
    return (0, h) if rows is None else (int(rows[0]), int(rows[1]))


def set_threads(n):
    lib().wso_set_threads(int(n))


def block_left(L, R, block_size, min_disparity, max_disparity, smooth=1.0,
               cost="ssd", subpixel=False, rows=None, threads=1):
    La, Li = _img(L)
    Ra, Ri = _img(R)
    out = np.zeros((La.shape[0], La.shape[1]), dtype=np.float64)
    y0, y1 = _rows(rows, La.shape[0])
    set_threads(threads)
    _check(lib().wso_block_left(ctypes.byref(Li), ctypes.byref(Ri), block_size,
                                min_disparity, max_disparity, smooth, COST[cost],
                                int(subpixel), y0, y1, out.ctypes.data, out.shape[1]))
    return out


def block_right(L, R, block_size, min_disparity, max_disparity, smooth=1.0,
                var_block=False, thres=19.0, cost="ssd", subpixel=False, rows=None,
                threads=1, return_max_block=False):
    La, Li = _img(L)
    Ra, Ri = _img(R)
    out = np.zeros((Ra.shape[0], Ra.shape[1]), dtype=np.float64)
    y0, y1 = _rows(rows, Ra.shape[0])
    mb = ctypes.c_int(0)
    set_threads(threads)
    _check(lib().wso_block_right(ctypes.byref(Li), ctypes.byref(Ri), block_size,
                                 min_disparity, max_disparity, smooth, int(var_block),
                                 thres, COST[cost], int(subpixel), y0, y1,
                                 out.ctypes.data, out.shape[1], ctypes.byref(mb)))
    return (out, mb.value) if return_max_block else out


def linear(L, R, smooth=1.0, search_range=200, rows=None, threads=1):
    La, Li = _img(L)
    Ra, Ri = _img(R)
    out = np.zeros((Ra.shape[0], Ra.shape[1]), dtype=np.float64)
    y0, y1 = _rows(rows, Ra.shape[0])
    set_threads(threads)
    _check(lib().wso_linear(ctypes.byref(Li), ctypes.byref(Ri), search_range, smooth,
                            y0, y1, out.ctypes.data, out.shape[1]))
    return out


def centred_norm(image, x0, y0, ww, wh):
    """cv::norm(window - cv::mean(window), NORM_L2) of the varBlock texture test (BlockSearch.cpp:125-129)."""
    a, im = _img(image)
    if x0 < 0 or y0 < 0 or x0 + ww > a.shape[1] or y0 + wh > a.shape[0]:
        raise ValueError("window outside the image")
    return float(lib().wso_centred_norm(ctypes.byref(im), x0, y0, ww, wh))


def evaldisp(disp, gt, mask, badthresh, maxdisp, rounddisp=0):
    d = np.ascontiguousarray(disp, dtype=np.float32)
    g = np.ascontiguousarray(gt, dtype=np.float32)
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    if not (d.shape == g.shape == m.shape):
        raise ValueError("shape mismatch")          # the reference asserts, utils.cpp:128-129
    res = (ctypes.c_double * 6)()
    _check(lib().wso_evaldisp(d.ctypes.data, g.ctypes.data, m.ctypes.data, d.shape[1],
                              d.shape[0], badthresh, maxdisp, int(rounddisp), res))
    return {"n": int(res[0]), "bad": res[1], "invalid": res[2], "total_bad": res[3],
            "avg_err": res[4], "valid": res[5]}


def _inv3(m):
    """3x3 inverse by adjugate / determinant (the closed form cv::invert uses for 3x3)."""
    m = np.asarray(m, dtype=np.float64).reshape(9)
    d = (m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6])
         + m[2] * (m[3] * m[7] - m[4] * m[6]))
    r = 1.0 / d
    return np.array([(m[4] * m[8] - m[5] * m[7]) * r, (m[2] * m[7] - m[1] * m[8]) * r, (m[1] * m[5] - m[2] * m[4]) * r,
                     (m[5] * m[6] - m[3] * m[8]) * r, (m[0] * m[8] - m[2] * m[6]) * r, (m[2] * m[3] - m[0] * m[5]) * r,
                     (m[3] * m[7] - m[4] * m[6]) * r, (m[1] * m[6] - m[0] * m[7]) * r, (m[0] * m[4] - m[1] * m[3]) * r])


def warp_nearest(src, matrix, dst_shape):
    """cv::warpPerspective(src, dst, matrix, dst.size(), INTER_NEAREST) as ImageRectifier calls it with
    matrix = H_.inv() (rectification.cpp:70-75, :82-87).  OpenCV is un-vendored in the reference; this
    restates its 4.x behaviour: invert the matrix, evaluate per 64-column block
    (M0*xb + M1*y + M2 + M0*x1) * (1/W), round half to even, constant 0 outside.  PARITY UNPINNED."""
    src = np.asarray(src, dtype=np.float64)
    m = _inv3(matrix)
    h, w = dst_shape
    out = np.zeros((h, w), dtype=np.float64)
    xs = np.arange(w)
    xb, x1 = (xs & ~63).astype(np.float64), (xs & 63).astype(np.float64)
    for y in range(h):
        X0 = m[0] * xb + m[1] * y + m[2]
        Y0 = m[3] * xb + m[4] * y + m[5]
        W = m[6] * xb + m[7] * y + m[8] + m[6] * x1
        with np.errstate(divide="ignore"):
            W = np.where(W != 0, 1.0 / W, 0.0)
        fx = np.clip((X0 + m[0] * x1) * W, -2147483648.0, 2147483647.0)
        fy = np.clip((Y0 + m[3] * x1) * W, -2147483648.0, 2147483647.0)
        X = np.rint(fx).astype(np.int64)
        Y = np.rint(fy).astype(np.int64)
        ok = (X >= 0) & (X < src.shape[1]) & (Y >= 0) & (Y < src.shape[0])
        out[y, ok] = src[Y[ok], X[ok]]
    return out


# ---- consumers of the map (src/Reconstruction/reconstruction.cpp), restated in NumPy --------------
def remove_disparity_outliers(disparity, kernel_size, thr_front, thr_back):
    """removeDisparityOutliers (reconstruction.cpp:5-18).  cv::blur = normalised k x k box filter,
    anchor k//2, BORDER_REFLECT_101, double sums scaled once by 1/(k*k) (OpenCV 4.x CV_32F path,
    un-vendored: PARITY UNPINNED).  Exact for integer-valued maps (the pipeline reads an 8-bit PNG)."""
    d = np.asarray(disparity, dtype=np.float32)
    k, a = int(kernel_size), int(kernel_size) // 2
    p = np.pad(d.astype(np.float64), ((a, k - 1 - a), (a, k - 1 - a)), mode="reflect")
    c = np.cumsum(np.cumsum(p, axis=0), axis=1)
    c = np.pad(c, ((1, 0), (1, 0)))
    h, w = d.shape
    s = c[k:k + h, k:k + w] - c[0:h, k:k + w] - c[k:k + h, 0:w] + c[0:h, 0:w]
    blurred = (s * (1.0 / (float(k) * float(k)))).astype(np.float32)
    out = d.copy()
    bad = (d > np.float32(thr_front) * blurred) | (d < np.float32(thr_back) * blurred)
    out[bad] = blurred[bad]
    return out


def convert_disparity_to_depth(disparity, focal_length, baseline):
    """convertDisparityToDepth (reconstruction.cpp:30-43): float32 f*b/d, 0 -> MINF (-inf)."""
    d = np.asarray(disparity, dtype=np.float32)
    fb = np.float32(focal_length) * np.float32(baseline)
    with np.errstate(divide="ignore"):
        z = (fb / d).astype(np.float32)
    z[d == 0] = -np.inf
    return z


def back_project(depth, intrinsics, bgr):
    """The vertex loop of reconstruction() (reconstruction.cpp:152-196)."""
    z = np.asarray(depth, dtype=np.float32)
    k = np.asarray(intrinsics, dtype=np.float32).reshape(3, 3)
    h, w = z.shape
    xs = np.arange(w, dtype=np.float32)[None, :].repeat(h, 0)
    ys = np.arange(h, dtype=np.float32)[:, None].repeat(w, 1)
    with np.errstate(invalid="ignore"):
        xc = ((xs * z - k[0, 2] * z) / k[0, 0]).astype(np.float32)
        yc = ((ys * z - k[1, 2] * z) / k[1, 1]).astype(np.float32)
    pos = np.stack([xc, yc, z, np.ones_like(z)], axis=2)
    col = np.concatenate([np.asarray(bgr)[:, :, ::-1], np.full((h, w, 1), 255, np.uint8)], axis=2).astype(np.uint8)
    inv = z == -np.inf
    pos[inv] = -np.inf
    col[inv] = 0
    return pos, col


def mesh_off_text(positions, colors, edge_threshold):
    """WriteMesh + CheckTriangularValidity (reconstruction.cpp:46-149) as a string (small meshes only)."""
    pos = np.asarray(positions, dtype=np.float32)
    col = np.asarray(colors, dtype=np.uint8)
    h, w = pos.shape[:2]
    P = pos.reshape(-1, 4)
    C = col.reshape(-1, 4)

    def ok(a, b, c):
        if P[a, 0] == -np.inf or P[b, 0] == -np.inf or P[c, 0] == -np.inf:
            return False
        def ln(p, q):
            return np.sqrt(np.float32(sum(np.float32(P[p, i] - P[q, i]) ** 2 for i in range(3))), dtype=np.float32)
        return not (ln(a, b) > edge_threshold or ln(a, c) > edge_threshold or ln(b, c) > edge_threshold)

    tris = []
    for y in range(h - 1):
        for x in range(w - 1):
            i00, i10, i01, i11 = y * w + x, (y + 1) * w + x, y * w + x + 1, (y + 1) * w + x + 1
            if ok(i00, i10, i01):
                tris.append((i00, i10, i01))
            if ok(i10, i11, i01):
                tris.append((i10, i11, i01))
    lines = ["COFF", "%d %d 0" % (w * h, len(tris))]
    for n in range(w * h):
        xyz = "0 0 0" if P[n, 0] == -np.inf else " ".join("%.6g" % float(v) for v in P[n, :3])
        lines.append("%s %d %d %d %d" % (xyz, C[n, 0], C[n, 1], C[n, 2], C[n, 3]))
    lines += ["3 %d %d %d" % t for t in tris]
    return "\n".join(lines) + "\n"
