"""MI355X-native WindowSearch: Python binding of the C-ABI (include/ws_stereo.h).

The product is the shared library `libws_stereo.so` (hand-written gfx950 HIP kernels
behind a C-ABI) and the C++ facade in `host/window_search.hpp`.  This module is the
thin ctypes layer the tests and bench.py call through; it mirrors the reference's
class surface:

    BlockSearch(left, right, blockSize, minDisparity, maxDisparity)   BlockSearch.h:11-15
        .computeDisparityMapLeft(smoothFactor)                         BlockSearch.h:28
        .computeDisparityMapRight(smoothFactor, varBlock, thres)       BlockSearch.h:37
    LinearSearch(left, right).computeDisparityMap(smoothFactor)        LinearSearch.h:13-19

There is no CPU fallback: if the library is missing or no HIP device answers, the calls
raise.  The CPU oracle under oracle/ is test infrastructure and is never imported here.
"""
import ctypes
import importlib.util
import os
import sys

import numpy as np

from . import build as _build

__all__ = ["WindowSearch", "BlockSearch", "LinearSearch", "WsError", "load_library",
           "read_pfm", "write_pfm", "read_ppm", "write_ppm", "write_mesh_off", "read_calib", "evaldisp", "VIEW_LEFT", "VIEW_RIGHT",
           "VIEW_LINEAR", "COST_SSD", "COST_SAD"]

VIEW_LEFT, VIEW_RIGHT, VIEW_LINEAR = 0, 1, 2
COST_SSD, COST_SAD = 0, 1
OUT_F32, OUT_F64 = 0, 1
_COST = {"ssd": COST_SSD, "sad": COST_SAD, COST_SSD: COST_SSD, COST_SAD: COST_SAD}

ERRORS = {-1: "WS_ERR_ARG", -2: "WS_ERR_GEOMETRY", -3: "WS_ERR_UNSUPPORTED", -4: "WS_ERR_HIP",
          -5: "WS_ERR_IO", -6: "WS_ERR_NOMEM"}

# every symbol include/ws_stereo.h declares (tests check the library exports all of them)
EXPORTS = ["ws_version", "ws_params_default", "ws_create", "ws_destroy", "ws_last_error",
           "ws_device_count", "ws_validate", "ws_plan", "ws_search_host", "ws_search_device", "ws_enqueue_host", "ws_wait", "ws_warp_nearest_host",
           "ws_warp_nearest_device", "ws_remove_disparity_outliers", "ws_convert_disparity_to_depth",
           "ws_back_project", "ws_write_mesh_off",
           "ws_timer_begin", "ws_timer_end", "ws_set_profiling", "ws_last_kernel_ms",
           "ws_last_launch_info", "ws_last_max_block", "ws_set_tuning", "ws_set_host_bands", "ws_last_host_paths", "ws_last_wire_format", "ws_last_outliers_path", "ws_device_status",
           "ws_pfm_read", "ws_pfm_write", "ws_free", "ws_ppm_read", "ws_ppm_write", "ws_calib_read", "ws_evaldisp"]


class WsError(RuntimeError):
    def __init__(self, code, message=""):
        self.code = code
        super().__init__("%s (%d): %s" % (ERRORS.get(code, "WS_ERR"), code, message))


class _Image(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("width", ctypes.c_int), ("height", ctypes.c_int),
                ("stride", ctypes.c_int)]


class _Params(ctypes.Structure):
    _fields_ = [("view", ctypes.c_int), ("cost", ctypes.c_int), ("block_size", ctypes.c_int),
                ("min_disparity", ctypes.c_int), ("max_disparity", ctypes.c_int),
                ("smooth_factor", ctypes.c_double), ("var_block", ctypes.c_int),
                ("thres", ctypes.c_double), ("subpixel", ctypes.c_int),
                ("linear_range", ctypes.c_int)]


class _Calib(ctypes.Structure):
    _fields_ = [("cam0", ctypes.c_float * 9), ("cam1", ctypes.c_float * 9),
                ("doffs", ctypes.c_float), ("baseline", ctypes.c_float),
                ("width", ctypes.c_int), ("height", ctypes.c_int), ("ndisp", ctypes.c_int)]


class _PlanInfo(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in (
        "marching", "x_per_thread", "d_per_thread", "x_runs", "d_chunks", "threads", "tiles",
        "strips", "strip_rows", "lds_bytes", "interior_x0", "interior_x1", "interior_y0",
        "interior_y1", "passes", "tile_cols")]


_lib = None


def _preload_hip_runtime():
    """Keep ONE HIP runtime in the process.  PyTorch-ROCm ships its own libamdhip64.so.7 (same
    SONAME as /opt/rocm's); if this library pulled in the system copy first, a later
    `import torch` would start a second HSA runtime and find no device.  So when PyTorch is
    installed but not imported yet, load the runtime it ships before libws_stereo.so."""
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.submodule_search_locations:
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(path):
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)


def load_library(build_if_missing=False):
    """dlopen libws_stereo.so.  Raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("WS_STEREO_LIB", _build.LIB)  # override: tuning variants only
    if not os.path.exists(path):
        if not build_if_missing:
            raise RuntimeError("libws_stereo.so is missing: run `python -c 'import __graft_entry__ as g; "
                               "g.build()'` (the HIP extension is the only compute path)")
        _build.build()
    _preload_hip_runtime()
    lib = ctypes.CDLL(path)
    P, vp, ci = ctypes.POINTER, ctypes.c_void_p, ctypes.c_int
    lib.ws_version.restype = ci
    lib.ws_params_default.argtypes = [P(_Params)]
    lib.ws_params_default.restype = None
    lib.ws_create.argtypes = [ci, P(vp)]
    lib.ws_destroy.argtypes = [vp]
    lib.ws_destroy.restype = None
    lib.ws_last_error.argtypes = [vp]
    lib.ws_last_error.restype = ctypes.c_char_p
    lib.ws_device_count.restype = ci
    lib.ws_validate.argtypes = [P(_Params), P(_Image), P(_Image)]
    lib.ws_plan.argtypes = [P(_Params), P(_Image), P(_Image), ci, P(_PlanInfo)]
    lib.ws_search_host.argtypes = [vp, P(_Params), P(_Image), P(_Image), vp, ci, ci]
    lib.ws_search_device.argtypes = [vp, P(_Params), P(_Image), P(_Image), vp, ci, vp]
    lib.ws_enqueue_host.argtypes = [vp, P(_Params), P(_Image), P(_Image), vp, ci, ci]
    lib.ws_wait.argtypes = [vp]
    cf = ctypes.c_float
    lib.ws_remove_disparity_outliers.argtypes = [vp, vp, ci, ci, ci, ci, cf, cf]
    lib.ws_convert_disparity_to_depth.argtypes = [vp, vp, ci, ci, ci, cf, cf, vp, ci]
    lib.ws_back_project.argtypes = [vp, vp, ci, ci, ci, P(cf), P(_Image), vp, vp]
    lib.ws_write_mesh_off.argtypes = [ctypes.c_char_p, vp, vp, ci, ci, cf]
    lib.ws_warp_nearest_host.argtypes = [vp, vp, ci, ci, ci, P(ctypes.c_double), vp, ci, ci, ci]
    lib.ws_warp_nearest_device.argtypes = [vp, vp, ci, ci, ci, P(ctypes.c_double), vp, ci, ci, ci, vp]
    lib.ws_timer_begin.argtypes = [vp, vp]
    lib.ws_timer_end.argtypes = [vp, vp, P(ctypes.c_float)]
    lib.ws_set_profiling.argtypes = [vp, ci]
    lib.ws_last_kernel_ms.argtypes = [vp, P(ctypes.c_float)]
    lib.ws_last_max_block.argtypes = [vp, ci, P(ci)]
    lib.ws_last_launch_info.argtypes = [vp, ctypes.c_char_p, ci, P(ci), P(ci), P(ci)]
    lib.ws_set_tuning.argtypes = [vp, ci, ci, ci]
    lib.ws_set_host_bands.argtypes = [vp, ci]
    lib.ws_last_host_paths.argtypes = [vp, P(ci)]
    lib.ws_last_wire_format.argtypes = [vp, P(ci)]
    lib.ws_last_outliers_path.argtypes = [vp, P(ci)]
    lib.ws_device_status.argtypes = [vp, vp]
    lib.ws_pfm_read.argtypes = [ctypes.c_char_p, P(P(ctypes.c_float)), P(ci), P(ci)]
    lib.ws_pfm_write.argtypes = [ctypes.c_char_p, vp, ci, ci, ci]
    lib.ws_ppm_read.argtypes = [ctypes.c_char_p, P(P(ctypes.c_uint8)), P(ci), P(ci)]
    lib.ws_ppm_write.argtypes = [ctypes.c_char_p, vp, ci, ci, ci]
    lib.ws_free.argtypes = [vp]
    lib.ws_free.restype = None
    lib.ws_calib_read.argtypes = [ctypes.c_char_p, P(_Calib)]
    lib.ws_evaldisp.argtypes = [vp, vp, vp, ci, ci, ctypes.c_float, ctypes.c_float, ci,
                                P(ctypes.c_double)]
    _lib = lib
    return lib


def device_count():
    """ws_device_count: the HIP devices this process sees (0 without one; never raises)."""
    return int(load_library().ws_device_count())


def make_params(view, block_size=7, min_disparity=0, max_disparity=64, smooth_factor=1.0,
                cost="ssd", var_block=False, thres=19.0, subpixel=False, linear_range=200):
    p = _Params()
    load_library().ws_params_default(ctypes.byref(p))
    p.view, p.cost = view, _COST[cost]
    p.block_size, p.min_disparity, p.max_disparity = block_size, min_disparity, max_disparity
    p.smooth_factor, p.var_block, p.thres = smooth_factor, int(var_block), thres
    p.subpixel, p.linear_range = int(subpixel), linear_range
    return p


def _host_image(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("expected an H x W x 3 uint8 (BGR) image")
    return a, _Image(a.ctypes.data, a.shape[1], a.shape[0], a.strides[0])


class WindowSearch:
    """One ws_context: a HIP device, its stream and scratch memory."""

    def __init__(self, device=0):
        self._lib = load_library()
        h = ctypes.c_void_p()
        rc = self._lib.ws_create(device, ctypes.byref(h))
        if rc != 0:
            raise WsError(rc, self._lib.ws_last_error(None).decode())
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ws_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise WsError(rc, self._lib.ws_last_error(self._h).decode())

    # -- host buffers (numpy in, numpy out) ------------------------------------------------
    def search(self, params, left, right, dtype=np.float64, out=None):
        """ws_search_host.  `out`: a C-contiguous float32 / float64 array of the map's shape to write into
        (a caller that keeps its output buffer); by default a fresh array per call, as the reference returns."""
        La, Li = _host_image(left)
        Ra, Ri = _host_image(right)
        shape = La.shape[:2] if params.view == VIEW_LEFT else Ra.shape[:2]
        if out is None:
            out = np.empty(shape, dtype=dtype)
        elif out.shape != shape or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("out must be a C-contiguous array of shape %s" % (shape,))
        code = OUT_F64 if out.dtype == np.float64 else OUT_F32
        if out.dtype not in (np.float32, np.float64):
            raise ValueError("dtype must be float32 or float64")
        self._check(self._lib.ws_search_host(self._h, ctypes.byref(params), ctypes.byref(Li),
                                             ctypes.byref(Ri), out.ctypes.data, shape[1], code))
        return out

    def search_many(self, params, pairs, dtype=np.float32):
        """Batched host path (ws_enqueue_host / ws_wait) over independent pairs.  Whatever happens, nothing is
        dropped before ws_wait has returned: the library copies from / into these buffers until then."""
        keep, outs = [], []
        try:
            for left, right in pairs:
                La, Li = _host_image(left)
                Ra, Ri = _host_image(right)
                shape = La.shape[:2] if params.view == VIEW_LEFT else Ra.shape[:2]
                out = np.empty(shape, dtype=dtype)
                code = OUT_F64 if out.dtype == np.float64 else OUT_F32
                keep.append((La, Ra, Li, Ri))
                outs.append(out)
                self._check(self._lib.ws_enqueue_host(self._h, ctypes.byref(params), ctypes.byref(Li),
                                                      ctypes.byref(Ri), out.ctypes.data, shape[1], code))
        except BaseException:
            self._lib.ws_wait(self._h)  # the pairs already enqueued still copy from / into keep and outs
            raise
        self._check(self._lib.ws_wait(self._h))
        del keep
        return outs

    # -- device buffers (torch tensors already in HBM) -------------------------------------
    def search_device(self, params, left_t, right_t, out_t, stream=None, check=False):
        """left_t/right_t: uint8 CUDA tensors H x W x 3 (contiguous rows); out_t: float32 H x W.
        Only enqueues.  The one thing a kernel can report after the fact -- the left view's smoothFactor raster pass
        giving up on the band above it, the map is then invalid -- surfaces in the next call that synchronises:
        device_status(stream), or this call with check=True (waits for the stream and raises).  A caller that
        synchronises with torch alone must call device_status() before trusting a left-view smoothFactor != 1 map."""
        Li = _Image(left_t.data_ptr(), left_t.shape[1], left_t.shape[0], left_t.stride(0))
        Ri = _Image(right_t.data_ptr(), right_t.shape[1], right_t.shape[0], right_t.stride(0))
        self._check(self._lib.ws_search_device(self._h, ctypes.byref(params), ctypes.byref(Li),
                                               ctypes.byref(Ri), out_t.data_ptr(), out_t.stride(0),
                                               ctypes.c_void_p(stream or 0)))
        if check:
            self.device_status(stream)

    def warp_nearest(self, src, matrix, dst_shape):
        """cv::warpPerspective(src, dst, matrix, dst_size, INTER_NEAREST) on a float64 map."""
        a = np.ascontiguousarray(src, dtype=np.float64)
        m = (ctypes.c_double * 9)(*np.asarray(matrix, dtype=np.float64).reshape(9))
        out = np.empty(dst_shape, dtype=np.float64)
        self._check(self._lib.ws_warp_nearest_host(self._h, a.ctypes.data, a.shape[1], a.shape[0], a.shape[1],
                                                   m, out.ctypes.data, out.shape[1], out.shape[0], out.shape[1]))
        return out

    # -- consumers (Reconstruction side) ---------------------------------------------------
    def remove_disparity_outliers(self, disparity, kernel_size, thr_front, thr_back):
        m = np.array(disparity, dtype=np.float32, order="C")
        self._check(self._lib.ws_remove_disparity_outliers(self._h, m.ctypes.data, m.shape[1], m.shape[0], m.shape[1],
                                                           kernel_size, thr_front, thr_back))
        return m

    def convert_disparity_to_depth(self, disparity, focal_length, baseline):
        d = np.ascontiguousarray(disparity, dtype=np.float32)
        out = np.empty_like(d)
        self._check(self._lib.ws_convert_disparity_to_depth(self._h, d.ctypes.data, d.shape[1], d.shape[0], d.shape[1],
                                                            focal_length, baseline, out.ctypes.data, out.shape[1]))
        return out

    def back_project(self, depth, intrinsics, bgr):
        z = np.ascontiguousarray(depth, dtype=np.float32)
        img, hdr = _host_image(bgr)
        k = (ctypes.c_float * 9)(*np.asarray(intrinsics, dtype=np.float32).reshape(9))
        pos = np.empty(z.shape + (4,), dtype=np.float32)
        col = np.empty(z.shape + (4,), dtype=np.uint8)
        self._check(self._lib.ws_back_project(self._h, z.ctypes.data, z.shape[1], z.shape[0], z.shape[1], k,
                                              ctypes.byref(hdr), pos.ctypes.data, col.ctypes.data))
        return pos, col

    def timer_begin(self, stream=None):
        self._check(self._lib.ws_timer_begin(self._h, ctypes.c_void_p(stream or 0)))

    def timer_end(self, stream=None):
        ms = ctypes.c_float()
        self._check(self._lib.ws_timer_end(self._h, ctypes.c_void_p(stream or 0), ctypes.byref(ms)))
        return ms.value

    def set_profiling(self, enable):
        self._check(self._lib.ws_set_profiling(self._h, int(bool(enable))))

    def last_kernel_ms(self):
        ms = ctypes.c_float()
        self._check(self._lib.ws_last_kernel_ms(self._h, ctypes.byref(ms)))
        return ms.value

    def last_launch(self):
        name = ctypes.create_string_buffer(128)
        t, w, l = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        self._check(self._lib.ws_last_launch_info(self._h, name, 128, ctypes.byref(t),
                                                  ctypes.byref(w), ctypes.byref(l)))
        return {"kernel": name.value.decode(), "threads": t.value, "workgroups": w.value,
                "lds_bytes": l.value}

    def last_max_block(self, block_size):
        v = ctypes.c_int()
        self._check(self._lib.ws_last_max_block(self._h, block_size, ctypes.byref(v)))
        return v.value

    def device_status(self, stream=None):
        """ws_device_status: wait for the stream and raise if a kernel flagged trouble since the last check."""
        self._check(self._lib.ws_device_status(self._h, ctypes.c_void_p(stream or 0)))

    def last_host_paths(self):
        """How the last host call's (left, right, out) bytes crossed: 'gathered', 'caller-pinned', 'staged'
        ('registered': rounds 2-3 only -- the library registers no caller memory any more)."""
        how = (ctypes.c_int * 3)()
        self._check(self._lib.ws_last_host_paths(self._h, how))
        return tuple(("gathered", "registered", "caller-pinned", "staged")[v] for v in how)

    def last_wire_format(self):
        """The format the last ws_search_host map crossed PCIe in: 'int16' (widened on the host) or 'float32'."""
        v = ctypes.c_int(0)
        self._check(self._lib.ws_last_wire_format(self._h, ctypes.byref(v)))
        return ("same", "int16", "float32")[v.value]

    def last_outliers_path(self):
        """Which box filter the last remove_disparity_outliers ran: 'double', 'integer', 'integer-then-double'."""
        v = ctypes.c_int(0)
        self._check(self._lib.ws_last_outliers_path(self._h, ctypes.byref(v)))
        return ("double", "integer", "integer-then-double")[v.value]

    def set_host_bands(self, bands=-1):
        self._check(self._lib.ws_set_host_bands(self._h, bands))

    def set_tuning(self, x_runs_per_tile=0, strip_rows=0, threads=0):
        self._check(self._lib.ws_set_tuning(self._h, x_runs_per_tile, strip_rows, threads))


def _shape_image(shape):
    """An image header with a non-null dummy pointer: enough for the device-free checks."""
    h, w = shape[:2]
    return _Image(1, w, h, 3 * w)


def validate(params, left_shape, right_shape):
    """Status code the search would return for these arguments (no device needed)."""
    Li, Ri = _shape_image(left_shape), _shape_image(right_shape)
    return load_library().ws_validate(ctypes.byref(params), ctypes.byref(Li), ctypes.byref(Ri))


def plan(params, left_shape, right_shape, num_cus=256):
    """Tiling the library would use (host logic only, no device needed)."""
    Li, Ri = _shape_image(left_shape), _shape_image(right_shape)
    info = _PlanInfo()
    rc = load_library().ws_plan(ctypes.byref(params), ctypes.byref(Li), ctypes.byref(Ri), num_cus,
                                ctypes.byref(info))
    if rc != 0:
        raise WsError(rc, load_library().ws_last_error(None).decode())
    return {n: getattr(info, n) for n, _ in _PlanInfo._fields_}


_default_ctx = None


def _ctx(ctx):
    global _default_ctx
    if ctx is not None:
        return ctx
    if _default_ctx is None:
        _default_ctx = WindowSearch(0)
    return _default_ctx


class BlockSearch:
    """Mirror of the reference's BlockSearch (BlockSearch.h:9-46); `cost`, `subpixel` and
    `context` are the build's additions."""

    def __init__(self, leftImage, rightImage, blockSize, minDisparity, maxDisparity,
                 cost="ssd", subpixel=False, context=None):
        self.leftImage_, self.rightImage_ = leftImage, rightImage
        self.blockSize_, self.minDisparity_, self.maxDisparity_ = blockSize, minDisparity, maxDisparity
        self.cost, self.subpixel, self._context = cost, subpixel, context

    def computeDisparityMapLeft(self, smoothFactor):
        p = make_params(VIEW_LEFT, self.blockSize_, self.minDisparity_, self.maxDisparity_,
                        smoothFactor, self.cost, subpixel=self.subpixel)
        return _ctx(self._context).search(p, self.leftImage_, self.rightImage_)

    def computeDisparityMapRight(self, smoothFactor, varBlock=False, thres=19.0):
        p = make_params(VIEW_RIGHT, self.blockSize_, self.minDisparity_, self.maxDisparity_,
                        smoothFactor, self.cost, varBlock, thres, self.subpixel)
        return _ctx(self._context).search(p, self.leftImage_, self.rightImage_)


class LinearSearch:
    """Mirror of the reference's LinearSearch (LinearSearch.h:9-20)."""

    def __init__(self, leftImage, rightImage, context=None, search_range=200):
        self.leftImage, self.rightImage = leftImage, rightImage
        self._context, self._range = context, search_range

    def computeDisparityMap(self, smoothFactor):
        p = make_params(VIEW_LINEAR, 1, 0, self._range, smoothFactor, "ssd",
                        linear_range=self._range)
        return _ctx(self._context).search(p, self.leftImage, self.rightImage)


# ---- Middlebury plumbing (no GPU needed) -------------------------------------------------
def read_pfm(path):
    lib = load_library()
    data = ctypes.POINTER(ctypes.c_float)()
    w, h = ctypes.c_int(), ctypes.c_int()
    rc = lib.ws_pfm_read(os.fsencode(path), ctypes.byref(data), ctypes.byref(w), ctypes.byref(h))
    if rc != 0:
        raise WsError(rc, "cannot read PFM %s" % path)
    try:
        return np.ctypeslib.as_array(data, shape=(h.value, w.value)).copy()
    finally:
        lib.ws_free(data)


def write_pfm(path, array):
    a = np.ascontiguousarray(array, dtype=np.float32)
    rc = load_library().ws_pfm_write(os.fsencode(path), a.ctypes.data, a.shape[1], a.shape[0], a.shape[1])
    if rc != 0:
        raise WsError(rc, "cannot write PFM %s" % path)


def write_mesh_off(path, positions, colors, edge_threshold):
    pos = np.ascontiguousarray(positions, dtype=np.float32)
    col = np.ascontiguousarray(colors, dtype=np.uint8)
    h, w = pos.shape[:2]
    rc = load_library().ws_write_mesh_off(os.fsencode(path), pos.ctypes.data, col.ctypes.data, w, h, edge_threshold)
    if rc != 0:
        raise WsError(rc, "cannot write mesh %s" % path)


def read_ppm(path):
    """P6 file -> H x W x 3 uint8 in BGR order (what cv::imread(IMREAD_COLOR) would give)."""
    lib = load_library()
    data = ctypes.POINTER(ctypes.c_uint8)()
    w, h = ctypes.c_int(), ctypes.c_int()
    rc = lib.ws_ppm_read(os.fsencode(path), ctypes.byref(data), ctypes.byref(w), ctypes.byref(h))
    if rc != 0:
        raise WsError(rc, "cannot read PPM %s" % path)
    try:
        return np.ctypeslib.as_array(data, shape=(h.value, w.value, 3)).copy()
    finally:
        lib.ws_free(data)


def write_ppm(path, bgr):
    a = np.ascontiguousarray(bgr, dtype=np.uint8)
    rc = load_library().ws_ppm_write(os.fsencode(path), a.ctypes.data, a.shape[1], a.shape[0], a.strides[0])
    if rc != 0:
        raise WsError(rc, "cannot write PPM %s" % path)


def read_calib(path):
    c = _Calib()
    rc = load_library().ws_calib_read(os.fsencode(path), ctypes.byref(c))
    if rc != 0:
        raise WsError(rc, "cannot parse calib %s" % path)
    return {"cam0": np.array(c.cam0[:], dtype=np.float32).reshape(3, 3),
            "cam1": np.array(c.cam1[:], dtype=np.float32).reshape(3, 3),
            "doffs": c.doffs, "baseline": c.baseline, "width": c.width, "height": c.height,
            "ndisp": c.ndisp}


def evaldisp(disp, gt, mask, badthresh, maxdisp, rounddisp=0):
    d = np.ascontiguousarray(disp, dtype=np.float32)
    g = np.ascontiguousarray(gt, dtype=np.float32)
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    if not (d.shape == g.shape == m.shape):
        raise ValueError("shape mismatch (the reference asserts, utils.cpp:128-129)")
    res = (ctypes.c_double * 6)()
    rc = load_library().ws_evaldisp(d.ctypes.data, g.ctypes.data, m.ctypes.data, d.shape[1],
                                    d.shape[0], badthresh, maxdisp, int(rounddisp), res)
    if rc != 0:
        raise WsError(rc, "evaldisp")
    return {"n": int(res[0]), "bad": res[1], "invalid": res[2], "total_bad": res[3],
            "avg_err": res[4], "valid": res[5]}
