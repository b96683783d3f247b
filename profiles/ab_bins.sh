#!/bin/bash
# the host builder's SAH bins per axis (RTR_BVH_BINS, read at every build): 32 against 48 / 64 on the bench scene and the mixed-size one
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
L=gpurun_out/r04/ab_bvh_bins.log; : > $L
run() { b=$1; shift; RTR_BVH_BINS=$b python3 bench.py --steps 48 --warmup 8 --no-cpu-baseline --present-frames 0 --isolated-frames 8 "$@" 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('bins $b $*', '| ms/frame', j['ms_per_step'], '| Mrays/s', j['value'], '| in launches', j.get('kernels_ms_in_flight_event_brackets'), '| alone', j['one_frame_at_a_time']['ms_per_step'], '| per ray', j['roofline']['per_ray'])" | tee -a $L; }
for r in 1 2; do for b in 32 48 64; do run $b; done; done
for b in 32 48; do run $b --workload sponza_mixed; done
for b in 32 48; do run $b --config 3; done
