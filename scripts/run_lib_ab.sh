# A/B of library builds under ab_libs/: the default bench line once per name and round, alternating.  usage: run_lib_ab.sh name...
set -e
mkdir -p gpurun_out/r3
for round in 1 2 ${AB_ROUNDS}; do for n in "$@"; do
PTR_HIP_LIB=$PWD/ab_libs/libptr_hip_$n.so timeout -k 10 200 python3 bench.py --cpu-seconds 0 --skip-parity --steps ${AB_STEPS:-200} > gpurun_out/r3/bench_ab.json 2> gpurun_out/r3/bench_ab.err
python3 -c "
import json; d=json.load(open('gpurun_out/r3/bench_ab.json')); k=d['kernels_ms_per_step_solo']; print('$n', round(d['value'],1), round(d['ms_per_step'],4), 'extend', round(k['wavefront.extend'],4), 'shade', round(k['wavefront.shade'],4))"
done; done
