#!/bin/bash
# The bench lines and probes kept under profiles/ for one round (everything but the profiler runs of scripts/round_profiles.sh).
#   usage (on the GPU box): scripts/round_lines.sh <outdir under gpurun_out/>
set -e
cd "$GRAFT_REPO_ROOT"
OUT=$1
mkdir -p "$OUT"
timeout -k 10 400 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "default line done"
# bench.py --gpus 2 starts its own two ranks; on this one-GPU box both share the device and exchange over gloo (a rehearsal of the
# launcher and the exchange, not a scaling figure)
PTR_BENCH_DEVICE=0 PTR_BENCH_BACKEND=gloo timeout -k 10 400 python3 bench.py --gpus 2 --cpu-seconds 0 > "$OUT/bench_gpus2_own_launcher_gloo.json" 2> "$OUT/bench_gpus2_own_launcher_gloo.err"; echo "--gpus 2 done"
timeout -k 10 600 bash scripts/rehearse_multi_rank.sh "$OUT/rehearse"
timeout -k 10 300 python3 scripts/turn_stats.py > "$OUT/turn_stats.txt" 2>&1; echo "turn stats done"
timeout -k 10 600 python3 scripts/partition_probe.py --worlds 1 2 4 8 --per-dispatch 10 --steps 20 > "$OUT/partition_probe.txt" 2>&1; echo "partition probe done"
timeout -k 10 120 scripts/microbench/valu_issue > "$OUT/microbench_valu_issue.txt" 2>&1; echo "valu_issue done"
timeout -k 10 120 scripts/microbench/node_fetch > "$OUT/microbench_node_fetch.txt" 2>&1; echo "node_fetch done"
