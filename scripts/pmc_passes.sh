#!/bin/bash
# PMC passes for the wavefront kernels, one rocprofv3 run per counter group (the hardware cannot hold more).
# usage: scripts/pmc_passes.sh <outdir> [extra args for render_frames.py]
#        PMC_TARGET=bench scripts/pmc_passes.sh <outdir>      profiles a short bench.py run instead (same command as the bench line)
set -e
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
# PMC_PASSES="sq1 tcp": only those passes (default: all)
run_pass() {
  name=$1; shift
  if [ -n "$PMC_PASSES" ] && ! echo " $PMC_PASSES " | grep -q " $name "; then return 0; fi
  timeout -s KILL 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -o p -- python3 $TARGET $EXTRA > "$OUT/$name.log" 2>&1
  echo "pass $name done"
}
EXTRA="$*"
export PTR_BENCH_SKIP_SINGLE=1
# bench: the default command, so that the launches are of the same sizes as the bench line's
if [ "$PMC_TARGET" = bench ]; then TARGET="bench.py --steps 20 --warmup 5 --cpu-seconds 0 --skip-parity"; else TARGET="scripts/render_frames.py --frames 4"; fi
run_pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY
run_pass sq2 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_BRANCH
run_pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY
run_pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_READ_sum
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE
