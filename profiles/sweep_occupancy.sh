#!/bin/bash
# What is k_shadow_trace4 bound by?  The same launches (eight frames per launch, moving camera) with fewer persistent workgroups per CU
# (tunable trace_wgs_per_cu: 8 = 8 waves per SIMD, the hardware maximum ... 2 = 2 waves per SIMD).  Time ~ 1 / waves: latency-bound
# (every wave waits for its own chain of dependent fetches; more waves = more chains in flight).  Flat: a shared unit is saturated.
for k in 8 7 6 5 4 3 2; do
  RTR_TRACE_WGS_PER_CU=$k python3 bench.py --steps 32 --warmup 8 --batch 8 --no-cpu-baseline --present-frames 0 --isolated-frames 4 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=j['roofline']
print('waves per SIMD $k | ms/frame', j['ms_per_step'], '| k_shadow_trace4 per frame in the 8-frame launches', r['avg_ms_per_frame'], '| one frame alone', r['one_frame_launch_ms'], '| clock MHz', r['clock_mhz'])
"
done
