#!/bin/bash
# the any-hit kernel's two scheduling thresholds once more after round 4's instruction trimming (the visit got cheaper relative to a refill pass)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
L=gpurun_out/r04/sweep_knobs_r04.log; : > $L
run() { env "$@" python3 bench.py --steps 48 --warmup 8 --no-cpu-baseline --present-frames 0 --isolated-frames 8 2>/dev/null | python3 -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$*', '| ms/frame', j['ms_per_step'], '| trace in launches', (j.get('kernels_ms_in_flight_event_brackets') or {}).get('shadow_trace'), '| alone', j['one_frame_at_a_time']['ms_per_step'], '| trace alone', j['kernels_ms']['shadow_trace'])" | tee -a $L; }
run RTR_TRACE_REFILL=20
run RTR_TRACE_REFILL=16
run RTR_TRACE_REFILL=24
run RTR_TRACE_REFILL=28
run RTR_TRACE_INNER_MIN=24
run RTR_TRACE_INNER_MIN=32
run RTR_TRACE_INNER_MIN=36
run RTR_TRACE_REFILL=24 RTR_TRACE_INNER_MIN=32
run RTR_TRACE_REFILL=20
