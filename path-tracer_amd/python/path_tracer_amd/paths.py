import os

PKG_ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
REPO_ROOT = os.path.normpath(os.path.join(PKG_ROOT, ".."))
CSRC = os.path.join(PKG_ROOT, "csrc")
HOST = os.path.join(PKG_ROOT, "host")
HOST_LIB = os.path.join(HOST, "libptr_host.so")
# PTR_HIP_LIB: another build of the same library, for A/B timing of kernel changes
HIP_LIB = os.environ.get("PTR_HIP_LIB") or os.path.join(CSRC, "libptr_hip.so")
INCLUDE = os.path.join(REPO_ROOT, "include")
