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


def test_headers_are_plain_c(tmp_path):
    """The boundary is a C ABI: both headers must compile as C99 on their own (no C++-isms, no torch or HIP types)."""
    import subprocess
    from path_tracer_amd import paths
    src = tmp_path / "abi.c"
    src.write_text('#include "ptr_hip.h"\n#include "ptr_host.h"\n'
                   'int use(ptr_ctx* c) { uint8_t k[64] = {0}; return ptr_dispatch(c, "compute", k, 1, 1, 1) + (int)sizeof(ptr_handle); }\n')
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-I", paths.INCLUDE, str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
