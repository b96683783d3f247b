#!/bin/bash
# Frames per launch of the pipeline: 8 against 16 (RTR_MAX_BATCH), whole frames and one rank of eight.
cd ${GRAFT_REPO_ROOT:-.}
show() { python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1', 'ms/frame', j['ms_per_step'], j['value'], j['unit'], j.get('kernels_ms_in_flight_event_brackets'))
"; }
for rep in 1 2; do
  for b in 8 16; do
    python3 bench.py --batch $b --frames-in-flight $b --steps 96 --warmup 32 --present-frames 0 --isolated-frames 0 2>/dev/null | show "[N=1, $b frames per launch, $b frame objects]"
    python3 bench.py --emulate-rank-of 8 --batch $b --frames-in-flight 32 --steps 192 --warmup 64 --present-frames 0 --isolated-frames 0 2>/dev/null | show "[rank 0 of 8, $b frames per launch, 32 frame objects]"
    python3 bench.py --emulate-rank-of 8 --batch $b --frames-in-flight $b --steps 192 --warmup 64 --present-frames 0 --isolated-frames 0 2>/dev/null | show "[rank 0 of 8, $b frames per launch, $b frame objects]"
    python3 bench.py --emulate-rank-of 4 --batch $b --frames-in-flight 32 --steps 192 --warmup 64 --present-frames 0 --isolated-frames 0 2>/dev/null | show "[rank 0 of 4, $b frames per launch, 32 frame objects]"
    python3 bench.py --config 1 --batch $b --frames-in-flight $b --steps 192 --warmup 64 --present-frames 0 --isolated-frames 0 2>/dev/null | show "[config 1 (Cornell 256x256), $b frames per launch]"
  done
done
