#!/bin/bash
# Can the HBM-bound stages of one launch (queue build, resolve) run UNDER the latency-bound any-hit kernel of another?  The any-hit kernel is
# persistent and takes every wave slot (8 workgroups per CU, all the LDS), so nothing co-runs with it by default; with 7 or 6 workgroups
# per CU (tunable trace_wgs_per_cu) a slot per SIMD and 20-40 KB of LDS stay free for another stream's kernels.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
L=gpurun_out/r04/sweep_corun.log; : > $L
run() { w=$1; b=$2; f=$3; RTR_TRACE_WGS_PER_CU=$w python3 bench.py --steps 96 --warmup 16 --batch $b --frames-in-flight $f --isolated-frames 2 --no-cpu-baseline --present-frames 0 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('trace_wgs_per_cu $w frames/launch $b frame objects $f | ms/frame', j['ms_per_step'], '| Mrays/s', j['value'], '| latency', j.get('frame_latency_ms'), '| kernels in flight', j.get('kernels_ms_in_flight_event_brackets') or j.get('kernels_ms'))" | tee -a $L; }
for w in 0 7 6; do
  run $w 8 8
  run $w 8 16
  run $w 4 8
  run $w 4 16
  run $w 2 8
done
