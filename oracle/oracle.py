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


def build(force=False):
    """Compile libws_oracle.so with the Makefile next to this file."""
    src = os.path.join(_HERE, "ws_oracle.c")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libws_oracle.so"])
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
