"""The file readers in front of the render path (JSON, PNG, JPEG, glTF/GLB) take files from outside: a corrupted
file must end in one of their own exceptions, never in a crash.  host/fuzz_decoders is those readers built with
AddressSanitizer + UBSan (host code only) and fed seeded random corruptions of valid files."""
import io
import os
import subprocess

import numpy as np
import pytest

from path_tracer_amd import gltf_io, paths, scenes


def test_corrupted_files_never_crash_the_readers(tmp_path):
    r = subprocess.run(["make", "-C", paths.HOST, "fuzz_decoders"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("the sanitizer build is not available here: " + r.stderr[-300:])
    rng = np.random.default_rng(7)
    d = str(tmp_path)
    files = []

    def put(name, data):
        with open(os.path.join(d, name), "wb") as f:
            f.write(data)
        files.append(os.path.join(d, name))
    put("rgba.png", gltf_io.encode_png(rng.integers(0, 256, (24, 20, 4), dtype=np.uint32), 6, 8, filter_type="cycle"))
    put("palette_interlaced.png", gltf_io.encode_png(rng.integers(0, 16, (13, 11, 1), dtype=np.uint32), 3, 4, filter_type="cycle", interlace=True,
                                                      palette=rng.integers(0, 256, (16, 3)), trns=bytes(range(8))))
    put("grey_alpha_16.png", gltf_io.encode_png(rng.integers(0, 65536, (9, 7, 2), dtype=np.uint32), 4, 16, filter_type=4))
    try:
        from PIL import Image
        img = rng.integers(0, 256, (40, 48, 3)).astype(np.uint8)
        for name, kw in (("a.jpg", dict(quality=90, subsampling=2)), ("b.jpg", dict(quality=75, subsampling=0, restart_marker_blocks=2)),
                         ("progressive.jpg", dict(quality=85, subsampling=2, progressive=True)),
                         ("progressive_restarts.jpg", dict(quality=60, subsampling=1, progressive=True, restart_marker_blocks=3))):
            buf = io.BytesIO()
            Image.fromarray(img).save(buf, format="JPEG", **kw)
            put(name, buf.getvalue())
    except ImportError:
        pass
    sc = scenes.tiny_scene()
    files.append(gltf_io.save_gltf(sc, d, "embedded", embed=True))
    files.append(gltf_io.save_gltf(sc, d, "binary", container="glb"))
    files.append(gltf_io.save_gltf(sc, d, "external"))
    put("doc.json", b'{"a":[1,2.5e3,{"b":"\\u00e9\\n"}],"c":null,"d":[true,false],"e":{"f":{"g":[[[]]]}}}')
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([os.path.join(paths.HOST, "fuzz_decoders"), *files, "--rounds", "400", "--seed", "11"], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "0 crashes" in r.stdout
    accepted = int(r.stdout.split("inputs accepted")[0].split()[-1])
    assert accepted >= len(files)          # at least every unmodified file is read
