#!/bin/bash
# RTR_QUEUE_AFFINITY=1: a workgroup's batches go to the list of the consumer XCD of its image band (not round-robin over the eight)
cd ${GRAFT_REPO_ROOT:-.}
show() { python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1', 'ms/frame', j['ms_per_step'], j['value'], (j.get('kernels_ms_in_flight_event_brackets') or {}), (j.get('verify') or {}))
"; }
for rep in 1 2 3; do for a in 0 1; do
  RTR_QUEUE_AFFINITY=$a python3 bench.py --steps 96 --warmup 32 --present-frames 0 --isolated-frames 0 --verify 2>/dev/null | show "[affinity $a]"
done; done
