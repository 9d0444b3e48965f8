#!/bin/bash
# Everything kept under profiles/ for one round, from the default bench command: the bench line, the rocprofv3 kernel statistics
# of that command, the counter passes (-> profiles/traffic.json, which bench.py reads for roofline.traffic) and their summary.
#   usage (on the GPU box): scripts/round_profiles.sh <round tag, e.g. r2>
set -e
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${TAG}_profiles
mkdir -p "$OUT"
# PTR_BENCH_SKIP_SINGLE and --skip-parity: without the one-sample-per-dispatch leg and the parity check's one-sample frames every launch of a kernel in this run has the size of the timed
# region's launches, so the average duration rocprofv3 reports is comparable with roofline.avg_launch_ms / solo.avg_launch_ms
PTR_BENCH_SKIP_SINGLE=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --skip-parity > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.err"
python3 scripts/phase_averages.py "$OUT/stats/s_kernel_trace.csv" "$OUT/bench_under_rocprof.json" > "$OUT/extend_launches.txt"
echo "kernel stats done"
PMC_TARGET=bench scripts/pmc_passes.sh "$OUT/pmc"
python3 scripts/pmc_summary.py "$OUT/pmc" > "$OUT/pmc_summary.txt"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --skip-parity > "$OUT/bench_plain.json" 2> "$OUT/bench_plain.err"
python3 scripts/traffic_from_pmc.py "$OUT/pmc" "$OUT/bench_plain.json" "$OUT/traffic.json" > /dev/null
echo "traffic.json done"
[ -n "$SKIP_G3" ] && exit 0   # SKIP_G3=1: the bench scene only (the 100 M-triangle passes take another ten minutes)
# the one scene class HBM can bind: 100.7 M triangles written out (7.8 GB of nodes and triangle records)
PMC_TARGET=bench scripts/pmc_passes.sh "$OUT/pmc_g3" --instances 6x6
python3 scripts/pmc_summary.py "$OUT/pmc_g3" > "$OUT/pmc_summary_g3.txt"
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --instances 6x6 > "$OUT/bench_g3.json" 2> "$OUT/bench_g3.err"
python3 scripts/traffic_from_pmc.py "$OUT/pmc_g3" "$OUT/bench_g3.json" "$OUT/traffic_g3.json" > /dev/null
echo "g3 done"
