#!/bin/bash
# Run-time knobs of the any-hit kernel with eight frames per launch (the default bench command): bash profiles/sweep_knobs_r03.sh
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 48 --warmup 8 --no-cpu-baseline --isolated-frames 0 --present-frames 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); k=j['kernels_ms_in_flight_event_brackets'] or j['kernels_ms']; print('trace %.4f ms/frame  frame %.4f ms  primary %.3f gen %.3f resolve %.3f' % (k['shadow_trace'], j['ms_per_frame'], k['primary'], k['shadow_gen'], k['resolve']))"; }
echo -n "defaults : "; run
for inner in 20 24 32 36; do echo -n "inner_min $inner : "; RTR_TRACE_INNER_MIN=$inner run; done
for refill in 12 16 24 28; do echo -n "refill $refill : "; RTR_TRACE_REFILL=$refill run; done
for batch in 128 512 1024; do echo -n "batch $batch : "; RTR_TRACE_BATCH=$batch run; done
for w in 6 7; do echo -n "wgs_per_cu $w : "; RTR_TRACE_WGS_PER_CU=$w run; done
echo -n "binned off : "; RTR_TRACE_BINNED=0 run
echo -n "primary persistent : "; RTR_PRIMARY_PERSIST=1 run
echo -n "resolve row waves : "; RTR_RESOLVE_ROW_WAVES=1 run
echo -n "defaults again : "; run
