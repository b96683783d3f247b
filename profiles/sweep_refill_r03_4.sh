#!/bin/bash
# Refill threshold / inner-loop threshold / batch length of the any-hit kernel, re-swept for revision r03.4 (a refill pass costs ~20 vector instructions less)
cd ${GRAFT_REPO_ROOT:-.}
show() { python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1', 'ms/frame', j['ms_per_step'], j['value'], (j.get('kernels_ms_in_flight_event_brackets') or {}).get('shadow_trace'))
"; }
run() { env "$@" python3 bench.py --steps 96 --warmup 16 --no-cpu-baseline --present-frames 0 --isolated-frames 0 2>/dev/null | show "[$*]"; }
for rep in 1 2; do
run RTR_TRACE_REFILL=20
run RTR_TRACE_REFILL=12
run RTR_TRACE_REFILL=16
run RTR_TRACE_REFILL=24
run RTR_TRACE_REFILL=28
run RTR_TRACE_INNER_MIN=24
run RTR_TRACE_INNER_MIN=32
run RTR_TRACE_BATCH=128
run RTR_TRACE_BATCH=512
run RTR_TRACE_REFILL=16 RTR_TRACE_INNER_MIN=24
done
