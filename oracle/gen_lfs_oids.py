#!/usr/bin/env python3
"""Reads the Git-LFS pointer files under /root/reference/models (the reference's scene assets are not in its checkout, only
their pointers: `oid sha256:<hex>` and `size <bytes>`) and writes the table the benchmark's real-asset probe checks candidate
files against: path_tracer_amd/reference_assets.json.  Test infrastructure / data generation only; run in the build container."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODELS = "/root/reference/models"


def main():
    table = {}
    for folder, _, files in os.walk(MODELS):
        for name in sorted(files):
            path = os.path.join(folder, name)
            with open(path, "rb") as f:
                head = f.read(200)
            if not head.startswith(b"version https://git-lfs.github.com/spec/v1"):
                continue
            fields = dict(line.split(" ", 1) for line in head.decode().strip().splitlines())
            table[os.path.relpath(path, MODELS)] = dict(sha256=fields["oid"].split(":", 1)[1], size=int(fields["size"]))
    out = os.path.join(ROOT, "path-tracer_amd", "python", "path_tracer_amd", "reference_assets.json")
    with open(out, "w") as f:
        json.dump(dict(source="Git-LFS pointer files of ladevieq/path-tracer, models/**", files=table), f, indent=0, sort_keys=True)
    print(f"{len(table)} pointers -> {out}", file=sys.stderr)


if __name__ == "__main__":
    main()
