#!/bin/bash
# Batch length (rays a wave reserves per atomic) of the any-hit kernel against the length of the launch's queue, revision r03.4
cd ${GRAFT_REPO_ROOT:-.}
show() { python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1', 'ms/frame', j['ms_per_step'], j['value'])
"; }
run() { tag=$1; shift; env "$@" python3 bench.py $EXTRA --no-cpu-baseline --present-frames 0 --isolated-frames 0 2>/dev/null | show "[$tag $*]"; }
for rep in 1 2; do
for b in 256 512 1024 2048; do
  EXTRA="--steps 96 --warmup 16" run "N=1, 16 per launch" RTR_TRACE_BATCH=$b
  EXTRA="--steps 20 --warmup 5" run "N=1, driver command (10 per launch)" RTR_TRACE_BATCH=$b
  EXTRA="--steps 96 --warmup 32 --emulate-rank-of 2" run "rank 0 of 2" RTR_TRACE_BATCH=$b
  EXTRA="--steps 192 --warmup 64 --emulate-rank-of 8" run "rank 0 of 8" RTR_TRACE_BATCH=$b
  EXTRA="--steps 96 --warmup 16 --config 3" run "config 3" RTR_TRACE_BATCH=$b
  EXTRA="--steps 96 --warmup 16 --batch 1 --frames-in-flight 4" run "one frame per launch, 4 in flight" RTR_TRACE_BATCH=$b
done; done
