"""MI355X-native render hot path of ladevieq/path-tracer: Python plumbing over the C ABI.

The product is the HIP library (csrc/, declared in include/ptr_hip.h) and the C++ host mirror of the
reference's scene / bvh / camera / compute-renderpass classes (host/).  This package only binds them
with ctypes for the test-suite and the benchmark driver, and generates synthetic scenes."""
from . import paths  # noqa: F401
