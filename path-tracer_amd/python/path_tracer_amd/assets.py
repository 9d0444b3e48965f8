"""Real-asset probe.  The reference's scenes (models/sponza, models/BistroInterior, models/BistroExterior) are Git-LFS
objects: its checkout holds 130-byte pointer files naming each object's sha256 and size (models/sponza/Sponza.gltf:1-3), not
the data, and nothing here may download them.  If somebody has put the real files next to this repository, the benchmark uses
them -- through the glTF loader, like any other file -- but only when every file it is about to read is the object the
pointer names; otherwise it renders its procedural stand-in and says so."""
import hashlib
import json
import os

TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_assets.json")

# scene of the benchmark -> the reference's folder and main file
SCENES = {"street_exterior": ("BistroExterior", "BistroExterior.gltf"), "street_interior": ("BistroInterior", "BistroInterior.gltf"),
          "atrium": ("sponza", "Sponza.gltf")}


def sha256_of(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 22), b""):
            h.update(chunk)
    return h.hexdigest()


def probe(models_dir, scene, table=None):
    """Returns (path of the verified .gltf or None, report).  Verified = the .gltf, every buffer and every image it names exist
    under models_dir/<folder>/ with the size and sha256 of the reference's pointer for that path."""
    folder, main = SCENES.get(scene, (None, None))
    report = dict(models_dir=models_dir, scene=scene, used=False)
    if folder is None or not models_dir or not os.path.isdir(os.path.join(models_dir, folder)):
        report["reason"] = "no such folder"
        return None, report
    if table is None:
        with open(TABLE) as f:
            table = json.load(f)["files"]
    gltf_path = os.path.join(models_dir, folder, main)
    if not os.path.isfile(gltf_path):
        report["reason"] = f"{folder}/{main} is not there"
        return None, report
    try:
        with open(gltf_path, "rb") as f:
            doc = json.loads(f.read().decode("utf-8"))
    except (ValueError, UnicodeDecodeError):
        report["reason"] = f"{folder}/{main} is not a glTF document (a Git-LFS pointer?)"
        return None, report
    names = [main] + [b["uri"] for b in doc.get("buffers", []) if "uri" in b] + [i["uri"] for i in doc.get("images", []) if "uri" in i]
    checked = 0
    for name in names:
        rel = os.path.normpath(os.path.join(folder, name))
        path = os.path.join(models_dir, rel)
        want = table.get(rel)
        if want is None or not os.path.isfile(path) or os.path.getsize(path) != want["size"] or sha256_of(path) != want["sha256"]:
            report["reason"] = f"{rel}: missing, or not the object the reference's pointer names"
            return None, report
        checked += 1
    report.update(used=True, files_verified=checked, gltf=gltf_path)
    return gltf_path, report
