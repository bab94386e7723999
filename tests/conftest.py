import faulthandler
import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# An abort must never again be "without a message" (round 2 lost two):
#  * pytest.ini captures at sys level only, so the HIP runtime's own last words on fd 2 reach the log;
#  * faulthandler (below, and pytest's own plugin) prints the Python traceback of the call that died on
#    SIGABRT / SIGSEGV / SIGBUS to the real stderr;
#  * the runtime logs its errors (AMD_LOG_LEVEL=1) to a file under gpurun_out/, whose tail is shown when a test
#    fails and which survives the process when it does not get that far.
AMD_LOG = os.path.join(ROOT, "gpurun_out", "amd_runtime_errors_%d.txt" % os.getpid())
if "AMD_LOG_LEVEL" not in os.environ:
    try:
        os.makedirs(os.path.dirname(AMD_LOG), exist_ok=True)
        os.environ["AMD_LOG_LEVEL"] = "1"
        os.environ["AMD_LOG_LEVEL_FILE"] = AMD_LOG
    except OSError:
        pass
faulthandler.enable(file=sys.__stderr__, all_threads=True)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_core_post_mortem():
    """A "Memory access fault by GPU" leaves `gpucore.<pid>` in the working directory and takes the process with it, so
    nothing of THAT run can look at it (round 3 lost gpucore.919 that way).  The next session on the same tree does:
    every dump found is opened with rocgdb in batch mode -- agents, queues, dispatches, the faulting wave's threads and
    backtraces -- and the text kept under gpurun_out/ (tools/gpu_session.sh does the same right after a run it wrapped)."""
    import shutil
    import subprocess
    rocgdb = shutil.which("rocgdb") or "/opt/rocm/bin/rocgdb"
    for core in sorted(glob.glob(os.path.join(ROOT, "gpucore.*")) + glob.glob(os.path.join(os.getcwd(), "gpucore.*"))):
        if core.endswith(".seen") or not os.path.exists(rocgdb):
            continue
        out = os.path.join(ROOT, "gpurun_out", os.path.basename(core) + ".rocgdb.txt")
        try:
            os.makedirs(os.path.dirname(out), exist_ok=True)
            r = subprocess.run([rocgdb, "-batch", "-ex", "core-file " + core, "-ex", "info agents", "-ex", "info queues",
                                "-ex", "info dispatches", "-ex", "info threads", "-ex", "thread apply all bt 4", sys.executable],
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=180)
            with open(out, "w") as f:
                f.write("# %s (%d bytes), found at the start of a test session\n%s" % (core, os.path.getsize(core), r.stdout[-40000:]))
            os.rename(core, core + ".seen")
            sys.__stderr__.write("conftest: GPU core dump %s -> %s\n" % (core, out))
        except (OSError, subprocess.SubprocessError):
            pass


def pytest_sessionstart(session):
    if glob.glob(os.path.join(ROOT, "gpucore.*")) or glob.glob(os.path.join(os.getcwd(), "gpucore.*")):
        _gpu_core_post_mortem()


def _amd_log_tail(lines=25):
    for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "amd_runtime_errors_%d.txt*" % os.getpid()))):
        try:
            with open(path, errors="replace") as f:
                tail = f.readlines()[-lines:]
        except OSError:
            continue
        if tail:
            return "".join(tail)
    return ""


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_makereport(item, call):
    outcome = yield
    rep = outcome.get_result()
    if rep.when == "call" and rep.failed and item.get_closest_marker("gpu"):
        tail = _amd_log_tail()
        if tail:
            rep.sections.append(("HIP runtime errors (AMD_LOG_LEVEL=1, last lines)", tail))


def golden_cases():
    out = []
    for p in sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))):
        z = np.load(p)
        if "expected" in z.files:
            out.append(os.path.basename(p)[:-4])
    return out


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: (z[k] if z[k].ndim else z[k].item()) for k in z.files}


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def wslib():
    """The built product library; building it is part of the CPU check (hipcc cross-compiles)."""
    from stereo_reconstruction_amd import build
    build.build()
    import stereo_reconstruction_amd as ws
    ws.load_library()
    return ws


@pytest.fixture(scope="session")
def gpu_ctx(wslib):
    """A device context.  On the GPU box a missing device/extension must fail, not skip."""
    ctx = wslib.WindowSearch(0)
    yield ctx
    ctx.close()
