#!/bin/bash
# Frame throughput (4 frames in flight) with the one-ray-per-lane and the persistent camera-ray kernel: bash profiles/ab_primary_r02.sh
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r02; mkdir -p "$OUT"
LOG=$OUT/ab_primary.log; : > "$LOG"
run() { echo "== $*" >> "$LOG"; env "$@" timeout -k 10 170 python3 $REPO/bench.py --no-cpu-baseline --present-frames 0 $ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], 'Mrays/s', d['ms_per_frame'], 'ms/frame ; isolated kernels', d['kernels_ms'], '; one at a time', d['one_frame_at_a_time']['ms_per_step'])" >> "$LOG" || echo failed >> "$LOG"; }
for ARGS in "" "--config 3 --steps 60"; do
  echo "#### bench.py $ARGS" >> "$LOG"
  run RTR_PRIMARY_PERSIST=0
  run RTR_PRIMARY_PERSIST=1
  run RTR_PRIMARY_PERSIST=0
  run RTR_PRIMARY_PERSIST=1
  run RTR_PRIMARY_PERSIST=1 RTR_PRIMARY_REFILL=64
  run RTR_PRIMARY_PERSIST=1 RTR_PRIMARY_REFILL=8
  run RTR_PRIMARY_PERSIST=1 RTR_PRIMARY_WGS_PER_CU=4
done
cat "$LOG"
