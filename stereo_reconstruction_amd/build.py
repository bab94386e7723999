"""In-tree build of libws_stereo.so (HIP kernels + C-ABI) for gfx950.

hipcc cross-compiles without a GPU; the .so stays next to this file so that it
travels with the tree (it is git-ignored, not gpurun-ignored).
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libws_stereo.so")
SOURCES = ["ws_march.hip", "ws_march_nd4.hip", "ws_prepass.hip", "ws_border.hip", "ws_smooth.hip", "ws_consumers.hip", "ws_capi.cpp"]
HEADERS = [os.path.join(CSRC, "ws_kernels.h"), os.path.join(CSRC, "ws_device.h"), os.path.join(CSRC, "ws_march_kernel.h"),
           os.path.join(HERE, "..", "include", "ws_stereo.h")]
ARCH = "gfx950"


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built")
    return exe


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


OBJ = os.path.join(HERE, "build")  # per-source objects (git-ignored): only what changed is compiled again
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"]
# per-source flags: the 8-disparities-per-thread marching kernels keep their prefix chains interleaved under the
# compiler's "max-ilp" scheduling strategy (ws_march_kernel.h)
EXTRA = {"ws_march.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}


def _object_stale(src, obj):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in [src] + HEADERS if os.path.exists(d))


def build(force=False, verbose=False):
    """Compile the shared library if it is missing or older than its sources."""
    if not force and not stale():
        return LIB
    # -ffp-contract=off: the few floating-point kernels (warp, depth, back-projection, smoothFactor)
    # must round every product like the reference's x86-64 build does; the hot kernels are integer
    os.makedirs(OBJ, exist_ok=True)
    jobs = []
    objs = []
    for name in SOURCES:
        src = os.path.join(CSRC, name)
        obj = os.path.join(OBJ, name + ".o")
        objs.append(obj)
        if force or _object_stale(src, obj):
            cmd = [hipcc()] + FLAGS + EXTRA.get(name, []) + ["-x", "hip", "-c", src, "-o", obj + ".tmp.%d" % os.getpid()]
            if verbose:
                cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
            jobs.append((subprocess.Popen(cmd), obj))
    failed = False
    for proc, obj in jobs:  # (at most len(SOURCES) = 6 compilers at once)
        tmp_o = obj + ".tmp.%d" % os.getpid()
        if proc.wait() == 0:
            os.replace(tmp_o, obj)
        else:
            failed = True
            if os.path.exists(tmp_o):
                os.remove(tmp_o)
    if failed:
        raise RuntimeError("hipcc failed (see its messages above)")
    tmp = LIB + ".tmp.%d" % os.getpid()  # written aside and renamed: a concurrent loader never sees half a file
    try:
        subprocess.check_call([hipcc(), "--offload-arch=" + ARCH, "-fPIC", "-shared", "-o", tmp] + objs)
        os.replace(tmp, LIB)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
