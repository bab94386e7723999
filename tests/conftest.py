import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
