#!/bin/bash
# Camera rays over the 4-wide view (k_primary4, tunable primary_wide) against over the BVH2 (k_primary + k_primary_tail): a frame alone and
# the driver's launch; both checked against the oracle first (-k primary, the deep-stack test walks all three forms).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
L=gpurun_out/r04/ab_primary_wide.log; : > $L
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "deep or tunable" 2>&1 | tail -2 | tee -a $L
run() { env "$@" python3 bench.py --steps 24 --warmup 6 --batch 1 --frames-in-flight 1 --isolated-frames 24 --no-cpu-baseline --present-frames 0 --verify 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$*', '| one frame at a time', j['one_frame_at_a_time']['ms_per_step'], '| ms/step', j['ms_per_step'], '| kernels', j['kernels_ms'], '| verify', j.get('verify'))" | tee -a $L; }
runb() { env "$@" python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline --present-frames 0 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$* (driver launch)', '| ms/step', j['ms_per_step'], '| frames per launch', j.get('frames_per_launch'), '| kernels', j['kernels_ms'])" | tee -a $L; }
run RTR_PRIMARY_WIDE=0
run RTR_PRIMARY_WIDE=1
run RTR_PRIMARY_WIDE=0
run RTR_PRIMARY_WIDE=1
runb RTR_PRIMARY_WIDE=0
runb RTR_PRIMARY_WIDE=1
