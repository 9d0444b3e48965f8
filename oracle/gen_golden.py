#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  Generates the golden fixtures under tests/golden/ by running the
UNMODIFIED reference host code (oracle/_ref/libptref.so, built by `make -C oracle ref` from
/root/reference/src/{bvh,vec3,camera}.cpp).  Needs /root/reference, so it runs only in the build
container; the fixtures it writes are committed and travel to the GPU box.

Fixtures are data only (inputs and the reference's outputs):
  tests/golden/bvh_cases.npz     triangle soups + the reference's packed_bvh_node arrays
  tests/golden/camera_cases.json camera parameters/ops + the reference's 128-byte camera structs
  tests/golden/layout.json       sizeof() of the structs the GPU ABI depends on
"""
import ctypes
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "..", "tests", "golden")


def load_ref():
    lib = ctypes.CDLL(os.path.join(HERE, "_ref", "libptref.so"))
    lib.ptref_bvh_build.restype = ctypes.c_int64
    lib.ptref_bvh_build.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    lib.ptref_camera.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                 ctypes.c_float, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
    return lib


def bvh_inputs():
    """Deterministic triangle sets: every branch of the builder (count 1, 2, 3-4, >4),
    centroid ties, coplanar and structured (mesh-ordered) input."""
    rng = np.random.default_rng(20240607)
    cases = {}
    for n in (1, 2, 3, 4, 5, 6, 7, 9, 13, 33, 100, 400):
        c = rng.random((n, 1, 3), dtype=np.float32) * 10
        cases[f"soup_{n}"] = (c + rng.random((n, 3, 3), dtype=np.float32) * 0.5).reshape(n, 9)
        cases[f"wide_{n}"] = (c + rng.random((n, 3, 3), dtype=np.float32) * 5).reshape(n, 9)
        q = (c + rng.random((n, 3, 3), dtype=np.float32) * 2).reshape(n, 9)
        cases[f"ties_{n}"] = np.round(q * 2) / 2  # coarse grid -> many equal centroid coordinates
    # structured grid in the y=0 plane (coplanar, mesh order), jittered so no centroid column is constant
    g = 9
    xs, zs = np.meshgrid(np.arange(g, dtype=np.float32), np.arange(g, dtype=np.float32), indexing="ij")
    jit = (rng.random((g, g), dtype=np.float32) - 0.5) * 0.2
    px, pz = xs + jit, zs + jit.T
    tris = []
    for i in range(g - 1):
        for j in range(g - 1):
            a = (px[i, j], 0.0, pz[i, j]); b = (px[i + 1, j], 0.0, pz[i + 1, j])
            c2 = (px[i + 1, j + 1], 0.0, pz[i + 1, j + 1]); d = (px[i, j + 1], 0.0, pz[i, j + 1])
            tris.append(a + b + c2)
            tris.append(a + c2 + d)
    cases["grid_coplanar"] = np.array(tris, dtype=np.float32)
    return {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in cases.items()}


CAMERA_CASES = [
    # the reference application's camera (src/main.cpp:208-223)
    dict(pos=[13, 2, -3], target=[0, 0, 0], v_fov=90, aspect=16.0 / 9.0, aperture=0.1, focus=10, ops=[]),
    dict(pos=[13, 2, -3], target=[0, 0, 0], v_fov=90, aspect=16.0 / 9.0, aperture=0.1, focus=10,
         ops=[[1, 0.5, 1.0, 1.5], [2, 0.01, 0, 0], [0, 4.0 / 3.0, 0, 0]]),
    dict(pos=[0, 1.7, 5], target=[0.3, 1.2, -2], v_fov=60, aspect=1.5, aperture=0.0, focus=7.3,
         ops=[[2, -0.3, 0, 0], [1, -1, 0.25, 3]]),
    dict(pos=[-4, 3, 2], target=[1, 0, 1], v_fov=37.5, aspect=2.39, aperture=0.25, focus=3,
         ops=[[2, 1.1, 0, 0], [2, 0.7, 0, 0], [0, 1.0, 0, 0]]),
    dict(pos=[0, 6, 14], target=[0, 3, 0], v_fov=70, aspect=1920.0 / 1080.0, aperture=0.05, focus=12, ops=[]),
]


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    ref = load_ref()

    sizes = (ctypes.c_uint32 * 4)()
    ref.ptref_sizes(sizes)
    with open(os.path.join(GOLDEN, "layout.json"), "w") as f:
        json.dump({"packed_bvh_node": sizes[0], "vec3": sizes[1], "camera": sizes[2], "triangle": sizes[3],
                   "scene_metadata": sizes[2] + 32, "gpu_material": 48}, f, indent=1)

    out = {}
    for name, pos in bvh_inputs().items():
        n = pos.shape[0]
        nodes = np.zeros((2 * n - 1, 8), dtype=np.uint32)
        got = ref.ptref_bvh_build(pos.ctypes.data, n, nodes.ctypes.data)
        assert got == 2 * n - 1
        out[f"{name}.positions"] = pos
        out[f"{name}.nodes"] = nodes
    np.savez_compressed(os.path.join(GOLDEN, "bvh_cases.npz"), **out)

    cams = []
    for c in CAMERA_CASES:
        pos = np.array(c["pos"], np.float32); tgt = np.array(c["target"], np.float32)
        ops = np.array(c["ops"], np.float32).reshape(-1, 4)
        buf = np.zeros(32, np.uint32)
        ref.ptref_camera(pos.ctypes.data, tgt.ctypes.data, c["v_fov"], c["aspect"], c["aperture"], c["focus"],
                         ops.ctypes.data, len(ops), buf.ctypes.data)
        cams.append(dict(c, struct_u32=[int(x) for x in buf]))
    with open(os.path.join(GOLDEN, "camera_cases.json"), "w") as f:
        json.dump(cams, f, indent=1)
    print("wrote", sorted(os.listdir(GOLDEN)))


if __name__ == "__main__":
    main()
