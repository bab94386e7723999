"""CPU-side checks of the product library: it loads, exports what include/ws_stereo.h declares,
refuses to run without a device, validates arguments like the reference throws, plans tiles,
and reads/writes the Middlebury files.  No compute is called here."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, load_golden


def declared_functions():
    text = open(os.path.join(ROOT, "include", "ws_stereo.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ws_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(wslib):
    lib = wslib.load_library()
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(wslib.EXPORTS) == names
    assert lib.ws_version() == 100


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "stereo_reconstruction_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "ws_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f


def test_create_fails_loudly_without_a_device(wslib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    with pytest.raises(wslib.WsError) as e:
        wslib.WindowSearch(0)
    assert e.value.code == -4 and "no CPU path" in str(e.value)


def test_validate_mirrors_the_reference_exceptions(wslib):
    ws = wslib
    ok = ws.make_params(ws.VIEW_LEFT, 7, 0, 64)
    assert ws.validate(ok, (100, 200), (100, 200)) == 0
    assert ws.validate(ws.make_params(ws.VIEW_LEFT, 8, 0, 64), (100, 200), (100, 200)) == -2   # even bs
    assert ws.validate(ws.make_params(ws.VIEW_RIGHT, 8, 0, 64), (100, 200), (100, 200)) == 0
    assert ws.validate(ws.make_params(ws.VIEW_RIGHT, 7, 0, 64), (90, 200), (100, 200)) == -2   # h1 < h2
    assert ws.validate(ws.make_params(ws.VIEW_RIGHT, 7, -1, 64), (100, 200), (100, 200)) == -2
    assert ws.validate(ws.make_params(ws.VIEW_LEFT, 7, 0, 64, smooth_factor=0.9), (100, 200), (100, 200)) == 0
    assert ws.validate(ws.make_params(ws.VIEW_LEFT, 7, 0, 64, smooth_factor=1.2), (100, 200), (100, 200)) == 0
    assert ws.validate(ws.make_params(ws.VIEW_LEFT, 7, 0, 64, smooth_factor=0.9), (100, 5000), (100, 5000)) == 0
    assert ws.validate(ws.make_params(ws.VIEW_RIGHT, 17, 0, 200, smooth_factor=0.9), (100, 200), (100, 200)) == 0
    assert ws.validate(ws.make_params(ws.VIEW_LINEAR, 1, 0, 200, smooth_factor=0.9), (100, 200), (100, 200)) == 0
    assert ws.validate(ws.make_params(ws.VIEW_RIGHT, 7, 0, 64, var_block=True), (100, 200), (100, 200)) == 0
    assert ws.validate(ws.make_params(ws.VIEW_RIGHT, 7, 0, 64, var_block=True, subpixel=True), (100, 200), (100, 200)) == -3
    assert ws.validate(ws.make_params(ws.VIEW_LEFT, 0, 0, 64), (100, 200), (100, 200)) == -1
    assert ws.validate(ws.make_params(5, 7, 0, 64), (100, 200), (100, 200)) == -1


def test_plan_covers_the_baseline_configs(wslib):
    ws = wslib
    for (w, h, bs, cost, d) in [(450, 375, 5, "sad", 64), (1500, 1000, 7, "ssd", 256),
                                (2964, 1988, 9, "sad", 512), (3840, 2160, 9, "ssd", 1024)]:
        p = ws.plan(ws.make_params(ws.VIEW_LEFT, bs, 0, d, 1.0, cost), (h, w), (h, w))
        half = (bs - 1) // 2
        assert p["marching"] == 1, (w, h, d)
        assert p["passes"] * p["d_chunks"] * p["d_per_thread"] >= d
        assert p["threads"] % 64 == 0 and p["threads"] <= 1024
        assert p["x_runs"] >= 4
        assert p["threads"] >= p["x_runs"] * p["d_chunks"]
        assert p["tiles"] * p["x_runs"] * p["x_per_thread"] >= w - 2 * half
        assert p["strips"] * p["strip_rows"] >= h - 2 * half
        assert p["lds_bytes"] <= 160 * 1024
        assert (p["interior_x0"], p["interior_x1"], p["interior_y0"], p["interior_y1"]) == (half, w - half, half, h - half)
    wide = ws.plan(ws.make_params(ws.VIEW_LEFT, 7, 0, 4000, 1.0, "ssd"), (500, 6000), (500, 6000))
    assert wide["marching"] == 1 and wide["passes"] > 1       # any disparity range: d-group passes
    # window sizes without a marching instantiation fall back to the brute-force kernel
    assert ws.plan(ws.make_params(ws.VIEW_LEFT, 21, 0, 64), (100, 200), (100, 200))["marching"] == 0


def test_pfm_round_trip_and_orientation(wslib, tmp_path):
    want = np.load(os.path.join(GOLDEN, "teddy_disp0GT_crop.npy"))
    got = wslib.read_pfm(os.path.join(GOLDEN, "teddy_disp0GT_crop.pfm"))
    assert got.dtype == np.float32 and np.array_equal(got, want)       # inf = unknown survives
    assert np.isinf(want).any() or np.isfinite(want).all()
    p = str(tmp_path / "out.pfm")
    wslib.write_pfm(p, want)
    assert np.array_equal(wslib.read_pfm(p), want)
    # independent reader: header + bottom-to-top little-endian rows
    with open(p, "rb") as f:
        assert f.readline().strip() == b"Pf"
        w, h = map(int, f.readline().split())
        assert float(f.readline()) < 0
        raw = np.frombuffer(f.read(), dtype="<f4").reshape(h, w)
    assert np.array_equal(raw[::-1], want)
    # big-endian file
    q = str(tmp_path / "be.pfm")
    with open(q, "wb") as f:
        f.write(b"Pf\n%d %d\n1.0\n" % (want.shape[1], want.shape[0]))
        f.write(np.ascontiguousarray(want[::-1], dtype=">f4").tobytes())
    assert np.array_equal(wslib.read_pfm(q), want)
    with pytest.raises(wslib.WsError):
        wslib.read_pfm(str(tmp_path / "missing.pfm"))


def test_calib_parse(wslib):
    c = wslib.read_calib(os.path.join(GOLDEN, "teddy_calib.txt"))
    assert np.array_equal(c["cam0"], np.array([[3000, 0, 398], [0, 3000, 375], [0, 0, 1]], dtype=np.float32))
    assert c["cam1"][0, 2] == 502 and c["ndisp"] == 128 and c["width"] == 900 and c["height"] == 750
    assert c["doffs"] == 104 and c["baseline"] == 80


def test_evaldisp_matches_the_oracle_restatement(wslib, oracle):
    g = load_golden("teddy_quarter")
    rng = np.random.default_rng(0)
    disp = np.where(np.isfinite(g["gt"]), g["gt"], 0) + rng.normal(0, 1.5, g["gt"].shape).astype(np.float32)
    disp[::7, ::5] = 0          # invalid pixels
    for rounddisp in (0, 1):
        a = wslib.evaldisp(disp, g["gt"], g["mask"], 2.0, 64.0, rounddisp)
        b = oracle.evaldisp(disp, g["gt"], g["mask"], 2.0, 64.0, rounddisp)
        assert a == b and a["n"] > 0 and a["invalid"] > 0
        # the two above follow utils.cpp:123-168 line by line and read alike: a vectorised third witness
        from oracle import brute
        c = brute.evaldisp_np(disp, g["gt"], g["mask"], 2.0, 64.0, rounddisp)
        assert c == a, (c, a)


def test_cxx_facade_compiles_and_links(wslib, tmp_path):
    exe = str(tmp_path / "facade_driver")
    cmd = ["g++", "-std=c++17", "-O1", "-I", ROOT, "-o", exe, os.path.join(ROOT, "tests", "cxx", "facade_driver.cpp"),
           "-L", os.path.join(ROOT, "stereo_reconstruction_amd"), "-lws_stereo",
           "-Wl,-rpath," + os.path.join(ROOT, "stereo_reconstruction_amd")]
    subprocess.check_call(cmd)
    assert os.path.exists(exe)


def test_opencv_adapters_compile_against_a_stub(tmp_path):
    """The WSAMD_WITH_OPENCV block of the facade (wsamd::view(cv::Mat), wsamd::to_cv) -- the code a maintainer of the
    reference compiles -- has no OpenCV to be compiled against in this image.  It is compiled
    and run here against tests/cxx/opencv_stub/opencv2/core.hpp, a dozen declarations with cv::Mat's names and
    signatures: a syntax-and-types check, NOT a check against OpenCV."""
    exe = str(tmp_path / "opencv_adapter_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "tests", "cxx", "opencv_stub"), "-I", ROOT,
                           "-o", exe, os.path.join(ROOT, "tests", "cxx", "opencv_adapter_check.cpp")])
    assert b"opencv adapters ok" in subprocess.check_output([exe])


def test_mesh_writer_matches_the_restatement(wslib, oracle, tmp_path):
    """WriteMesh (reconstruction.cpp:72-149) is host-only: text identical to the Python restatement."""
    rng = np.random.default_rng(4)
    h, w = 9, 13
    z = rng.uniform(20, 22, size=(h, w)).astype(np.float32)
    z[2, 3] = -np.inf
    bgr = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    pos, col = oracle.back_project(z, [[100, 0, 6], [0, 100, 4], [0, 0, 1]], bgr)
    path = str(tmp_path / "m.off")
    for thr in (0.5, 5.0):
        wslib.write_mesh_off(path, pos, col, thr)
        assert open(path).read() == oracle.mesh_off_text(pos, col, thr)
    assert open(path).read().startswith("COFF\n117 ")


def test_ppm_round_trip(wslib, tmp_path):
    rng = np.random.default_rng(6)
    img = rng.integers(0, 256, size=(11, 17, 3), dtype=np.uint8)          # BGR
    p = str(tmp_path / "a.ppm")
    wslib.write_ppm(p, img)
    raw = open(p, "rb").read()
    assert raw.startswith(b"P6\n17 11\n255\n") and raw[-3:] == bytes(img[-1, -1, ::-1])   # file is RGB
    assert np.array_equal(wslib.read_ppm(p), img)
    q = str(tmp_path / "c.ppm")
    open(q, "wb").write(b"P6\n# a comment\n17 11\n255\n" + raw[len(b"P6\n17 11\n255\n"):])
    assert np.array_equal(wslib.read_ppm(q), img)


def test_example_pipeline_compiles(wslib, tmp_path):
    exe = str(tmp_path / "pipeline_main")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", ROOT, "-o", exe, os.path.join(ROOT, "examples", "pipeline_main.cpp"),
                           "-L", os.path.join(ROOT, "stereo_reconstruction_amd"), "-lws_stereo",
                           "-Wl,-rpath," + os.path.join(ROOT, "stereo_reconstruction_amd")])


def test_planner_choices_without_a_device(wslib):
    """ws_plan is host logic: the tiling of the BASELINE.json configs, the strip count that minimises
    rounds x (rows + warm-up), and the candidate range clamped to what the geometry allows."""
    ws = wslib
    p = ws.make_params(ws.VIEW_LEFT, 7, 0, 256, 1.0, "ssd")
    c2 = ws.plan(p, (1000, 1500, 3), (1000, 1500, 3))
    assert c2["marching"] == 1 and c2["passes"] == 1 and c2["threads"] == 512
    assert c2["tiles"] * c2["strips"] <= 256 and c2["tiles"] * c2["strips"] >= 240      # one round on 256 CUs
    assert c2["strip_rows"] * c2["strips"] >= 994
    p = ws.make_params(ws.VIEW_LEFT, 9, 0, 512, 1.0, "sad")
    c3 = ws.plan(p, (1988, 2964, 3), (1988, 2964, 3))
    cap = 256            # 8 disparities per thread: one workgroup per CU at a time (march_slots_per_cu)
    rounds = -(-c3["tiles"] * c3["strips"] // cap)
    assert c3["strip_rows"] >= 64 and rounds * cap - c3["tiles"] * c3["strips"] < 64    # tall strips, full rounds
    # packed SAD 9x9 runs the halo-exchange kernel: tiles of 16 runs that hand out 15 (the last one feeds its neighbour),
    # the range in d-group passes of 16 runs x (threads / 16) chunks; SSD keeps whole tiles
    assert c3["x_runs"] == 16 and c3["tile_cols"] == 15 * 8 and c3["d_per_thread"] == 16
    assert c3["passes"] == 1 and c3["d_chunks"] == 32 and c3["threads"] == 512
    assert c2["tile_cols"] == c2["x_runs"] * 8
    p = ws.make_params(ws.VIEW_LEFT, 9, 0, 1024, 1.0, "ssd")
    c5 = ws.plan(p, (2160, 3840, 3), (2160, 3840, 3))                                   # D = 1024: several d-group passes
    assert c5["passes"] >= 2 and c5["passes"] * c5["d_chunks"] * c5["d_per_thread"] >= 1024
    # the thread shape follows the planner's own cost model (ws_march.hip: march_shape): 8 disparities per thread
    # wherever both shapes fill the chip alike, 4 where the range is narrow for the image or the image small
    assert c2["d_per_thread"] == 8 and c5["d_per_thread"] == 8
    p = ws.make_params(ws.VIEW_LEFT, 17, 0, 200, 1.0, "ssd")
    assert ws.plan(p, (750, 900, 3), (750, 900, 3))["d_per_thread"] == 4
    p = ws.make_params(ws.VIEW_LEFT, 5, 0, 64, 1.0, "sad")
    assert ws.plan(p, (375, 450, 3), (375, 450, 3))["d_per_thread"] == 4               # config 1's shape
    for view in (ws.VIEW_LEFT, ws.VIEW_RIGHT):
        for cost in ("ssd", "sad"):
            p = ws.make_params(view, 7, 0, 3000, 1.0, cost)                              # 10 x the width
            info = ws.plan(p, (40, 300, 3), (40, 300, 3))
            # the range is clamped to what the geometry allows (< 300 candidates): one pass of <= 512 of them at 8
            # per thread, two passes of <= 256 at 4 per thread -- not the 6..12 passes 3000 candidates would take
            assert info["marching"] == 1 and info["passes"] == (1 if info["d_per_thread"] == 8 else 2), (view, cost, info)
            assert info["passes"] * info["d_chunks"] * info["d_per_thread"] < 2 * 300
    # what the reference rejects is rejected without a device too
    assert ws.validate(ws.make_params(ws.VIEW_LEFT, 6, 0, 16), (40, 100, 3), (40, 100, 3)) == -2
    assert ws.validate(ws.make_params(ws.VIEW_RIGHT, 7, 0, 16), (30, 100, 3), (40, 100, 3)) == -2
    assert ws.validate(ws.make_params(ws.VIEW_RIGHT, 3, 0, 16), (30, 100, 3), (90, 100, 3)) == 0     # bs <= 4: legal
    assert ws.validate(ws.make_params(ws.VIEW_RIGHT, 7, 0, 16), (39, 100, 3), (40, 100, 3)) == 0     # h2 = h1 + 1: legal
    assert ws.validate(ws.make_params(ws.VIEW_RIGHT, 3, 0, 16, var_block=True), (30, 100, 3), (90, 100, 3)) == -2
