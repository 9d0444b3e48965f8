#!/bin/bash
# (--cpu-seconds 0 leaves the CPU baseline leg out; every line still carries parity_check: its image against the oracle on four bands.)
# The secondary bench lines kept under profiles/: GPU-built tree, 25 M / 10^8 triangles written out, 10^8 triangles as placed copies.
set -e
cd "$GRAFT_REPO_ROOT"
run() { out=$1; shift; timeout -k 10 500 python3 bench.py --cpu-seconds 0 "$@" 2> gpurun_out/$out.err > gpurun_out/$out.json; echo "$out done"; }
run line_gpu_bvh --bvh gpu
run line_auto_flatten --bvh auto
run line_25M --instances 3x3
run line_98M --instances 6x6
run line_98M_gpu_bvh --instances 6x6 --bvh gpu
run line_two_level --instances 6x6 --two-level
run line_two_level_gpu_bvh --instances 6x6 --two-level --bvh gpu
run line_4k --width 3840 --height 2160 --steps 20
run line_720p --width 1280 --height 720
