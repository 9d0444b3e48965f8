#!/bin/bash
# A/B timing of two builds of libptr_hip.so on one device: alternates bench runs of ab_libs/libptr_hip_base.so
# and the in-tree library.  usage: scripts/ab_libs.sh <rounds> [bench args]
set -e
ROUNDS=${1:-3}; shift || true
cd "$GRAFT_REPO_ROOT"
for i in $(seq 1 "$ROUNDS"); do
  for which in base new; do
    if [ "$which" = base ]; then export PTR_HIP_LIB="$GRAFT_REPO_ROOT/ab_libs/libptr_hip_base.so"; else unset PTR_HIP_LIB; fi
    timeout -k 10 200 python3 bench.py --cpu-seconds 0 --steps 40 --warmup 5 "$@" 2> /dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read()); k = d['kernels_ms_per_step_solo']
print('$which', 'ms/frame %.4f' % d['ms_per_step'], ' '.join('%s %.4f' % (n.split('.')[-1], v) for n, v in k.items()))"
  done
done
