#!/bin/bash
# One GPU renders shard 0 of N (no gather): how many eight-frame launches should a rank keep in flight?
cd ${GRAFT_REPO_ROOT:-.}
show() { python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1', 'ms/frame', j['ms_per_step'], j['value'], j['unit'])
"; }
for rep in 1 2; do
for n in 8 4 2 1; do
  for fif in 8 16 32; do
    extra=""; [ $n -gt 1 ] && extra="--emulate-rank-of $n"
    GPU_MAX_HW_QUEUES=8 python3 bench.py $extra --batch 8 --frames-in-flight $fif --steps 96 --warmup 32 --present-frames 0 --isolated-frames 0 2>/dev/null | show "[rank 0 of $n, 8 frames per launch, $fif frame objects]"
  done
done; done
