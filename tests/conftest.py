import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "path-tracer_amd", "python"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def repo_root():
    return ROOT


@pytest.fixture(scope="session", autouse=True)
def _native_libs():
    """CPU-side libraries are built on demand; the HIP library must already be in the tree on a GPU box."""
    from path_tracer_amd import build, paths
    if not os.path.exists(paths.HOST_LIB):
        build.build_host()
    import pyoracle
    pyoracle.build()
