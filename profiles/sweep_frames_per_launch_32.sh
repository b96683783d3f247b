#!/bin/bash
# Frames per launch beyond 16 (RTR_MAX_BATCH 32: kernel arguments of 7 KB), whole frames, the driver's command, shards, config 3
cd ${GRAFT_REPO_ROOT:-.}
show() { python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1', 'ms/frame', j['ms_per_step'], j['value'], j['unit'], 'launches', j.get('timed_launches'), 'slots', j.get('frames_in_flight'))
"; }
for rep in 1 2; do
  for b in 16 32; do
    python3 bench.py --batch $b --frames-in-flight $b --steps 192 --warmup 32 --present-frames 0 --isolated-frames 0 --no-cpu-baseline 2>/dev/null | show "[N=1, at most $b per launch]"
    python3 bench.py --batch $b --frames-in-flight $b --steps 20 --warmup 5 --present-frames 0 --isolated-frames 0 --no-cpu-baseline 2>/dev/null | show "[N=1, the driver's 20 steps, at most $b per launch]"
    python3 bench.py --emulate-rank-of 8 --batch $b --frames-in-flight $((2*b)) --steps 192 --warmup 64 --present-frames 0 --isolated-frames 0 --no-cpu-baseline 2>/dev/null | show "[rank 0 of 8, $b per launch, two launches in flight]"
    python3 bench.py --emulate-rank-of 4 --batch $b --frames-in-flight $((2*b)) --steps 192 --warmup 64 --present-frames 0 --isolated-frames 0 --no-cpu-baseline 2>/dev/null | show "[rank 0 of 4, $b per launch, two launches in flight]"
    python3 bench.py --emulate-rank-of 2 --batch $b --frames-in-flight $b --steps 192 --warmup 64 --present-frames 0 --isolated-frames 0 --no-cpu-baseline 2>/dev/null | show "[rank 0 of 2, $b per launch]"
    python3 bench.py --config 3 --batch $b --frames-in-flight $b --steps 96 --warmup 32 --present-frames 0 --isolated-frames 0 --no-cpu-baseline 2>/dev/null | show "[config 3, at most $b per launch]"
    python3 bench.py --config 1 --batch $b --frames-in-flight $b --steps 192 --warmup 64 --present-frames 0 --isolated-frames 0 --no-cpu-baseline 2>/dev/null | show "[config 1, at most $b per launch]"
  done
done
