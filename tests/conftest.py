import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def orc():
    from oracle import orc as o
    o.lib()
    return o


@pytest.fixture(scope="session")
def cornell():
    from aten_amd.scene import scenedefs
    return scenedefs.cornell_box()


@pytest.fixture(scope="session")
def sponza():
    from aten_amd.scene import scenedefs
    return scenedefs.sponza_lod()


@pytest.fixture(scope="session")
def sponza_disney():
    from aten_amd import layout as L
    from aten_amd.scene import scenedefs
    return scenedefs.sponza_lod(mtype=L.MTRL_DISNEY)


@pytest.fixture(scope="session")
def gpu():
    """One PathTracing context on cuda:0; fails (does not skip) when the HIP library or GPU is missing."""
    # torch (device tensors for the tile / gather tests) bundles its own HIP runtime: when it initialises AFTER
    # libaten_amd.so has brought up the system one it reports "No HIP GPUs are available" -- so bring it up first, as
    # bench.py does
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    from aten_amd.renderer import PathTracing
    r = PathTracing(0)
    yield r
    r.close()


def make_camera(orc, cam, w, h):
    return orc.create_camera(cam["pos"], cam["at"], cam["vfov"], w, h)


def ulp_diff(a, b):
    """Distance in units in the last place between two float32 arrays (same sign assumed mostly)."""
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    ia = a.view(np.int32).astype(np.int64); ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, np.int64(-2**31) - ia, ia)
    ib = np.where(ib < 0, np.int64(-2**31) - ib, ib)
    return np.abs(ia - ib)
