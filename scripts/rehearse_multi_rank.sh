#!/bin/bash
# Rehearsal of bench.py's N > 1 paths on a ONE-GPU box: two ranks share the device and exchange over gloo (host staging),
# then one rank runs the RCCL collectives alone.  The numbers say nothing about scaling; the lines show that every leg of the
# multi-GPU flow (partition, exchange inside the timed region, reductions of the timings, parity of the merged frame) runs.
#   usage: scripts/rehearse_multi_rank.sh <outdir>
set -e
OUT=$1
mkdir -p "$OUT"
for mode in tiles samples; do
  PTR_BENCH_DEVICE=0 PTR_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 3 --cpu-seconds 4 --partition $mode \
    > "$OUT/bench_2rank_gloo_$mode.json" 2> "$OUT/bench_2rank_gloo_$mode.err"
  echo "2 ranks, gloo, $mode: done"
  PTR_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 3 --cpu-seconds 0 --partition $mode \
    > "$OUT/bench_1rank_rccl_$mode.json" 2> "$OUT/bench_1rank_rccl_$mode.err"
  echo "1 rank, RCCL, $mode: done"
done
