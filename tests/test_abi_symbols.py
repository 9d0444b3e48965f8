"""The C-ABI libraries load and export every symbol their headers declare (no GPU needed: nothing is called)."""
import ctypes
import os
import re

import pytest

from path_tracer_amd import paths


def declared(header, prefix):
    text = open(os.path.join(paths.INCLUDE, header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", text)))


def test_hip_library_exports_every_declared_entry_point():
    if not os.path.exists(paths.HIP_LIB):
        pytest.fail(f"{paths.HIP_LIB} missing: build it with __graft_entry__.build() (hipcc cross-compiles without a GPU)")
    names = declared("ptr_hip.h", "ptr_")
    assert len(names) >= 25
    lib = ctypes.CDLL(paths.HIP_LIB)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    from path_tracer_amd import hip
    assert sorted(hip.ABI_SYMBOLS) == names


def test_host_library_exports_every_declared_entry_point():
    names = declared("ptr_host.h", "ptrh_")
    lib = ctypes.CDLL(paths.HOST_LIB)
    missing = [n for n in names if not hasattr(lib, n)]
    assert names and not missing, missing
