import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gl():
    """The product package (ctypes binding of libginger_hip.so)."""
    from __graft_entry__ import _load_pkg
    mod = sys.modules.get("ginger_lib_amd") or _load_pkg()
    mod.load_library()
    return mod


@pytest.fixture(scope="session")
def gpu(gl):
    gl.init()   # raises GingerHipError without a gfx950 device: GPU tests must not silently pass
    return gl


@pytest.fixture
def no_dedup(gpu):
    """Tests whose keys are a small pool of points tiled to a large size (so that the oracle can referee, or to fill buckets with
    equal and opposite bases on purpose) switch the adding-up of equal bases off: they are about the bucket paths at that size."""
    gpu.msm_set_dedup(0)
    yield
    gpu.msm_set_dedup(1)
