"""Builds the native libraries in-tree (the .so files travel with the repository snapshot)."""
import os
import subprocess

from . import paths


def _make(directory, *targets):
    subprocess.check_call(["make", "-C", directory, *targets], stdout=subprocess.DEVNULL)


def build_host():
    _make(paths.HOST, "libptr_host.so")
    return paths.HOST_LIB


def build_hip(jobs=4):
    subprocess.check_call(["make", "-C", paths.CSRC, f"-j{jobs}", "libptr_hip.so"], stdout=subprocess.DEVNULL)
    return paths.HIP_LIB


def build_headless():
    """The C++ headless driver (links libptr_hip.so through its C ABI)."""
    _make(paths.HOST, "ptr_headless", "exchange_selftest")
    return os.path.join(paths.HOST, "ptr_headless")


def build_all():
    build_host()
    build_hip()
    build_headless()
    assert os.path.exists(paths.HOST_LIB) and os.path.exists(paths.HIP_LIB)
