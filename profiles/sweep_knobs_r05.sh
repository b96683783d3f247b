#!/bin/bash
# The any-hit kernel's refill / inner-loop thresholds once more, now that four rays in nine start at their own leaf and finish in the first
# leaf phase (round 5):  bash profiles/sweep_knobs_r05.sh > gpurun_out/r05/sweep_knobs_r05.log
cd $GRAFT_REPO_ROOT
run() { echo -n "$1 | "; env $1 timeout -k 5 90 python bench.py --steps 21 --warmup 7 --no-cpu-baseline --isolated-frames 10 --present-frames 0 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('ms/frame', j['ms_per_step'], '| trace in launches', j['kernels_ms_in_flight_event_brackets']['shadow_trace'], '| alone', j['one_frame_at_a_time']['ms_per_step'], '| trace alone', j['kernels_ms']['shadow_trace'])"; }
run "RTR_TRACE_REFILL=20"
for r in 12 16 24 28 32; do run "RTR_TRACE_REFILL=$r"; done
for m in 20 24 32 36; do run "RTR_TRACE_INNER_MIN=$m"; done
run "RTR_TRACE_REFILL=16 RTR_TRACE_INNER_MIN=24"
run "RTR_TRACE_REFILL=24 RTR_TRACE_INNER_MIN=24"
run "RTR_TRACE_REFILL=28 RTR_TRACE_INNER_MIN=32"
run "RTR_TRACE_BATCH=512"
run "RTR_TRACE_BATCH=1024"
run "RTR_TRACE_REFILL=20"
