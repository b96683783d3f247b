#!/bin/bash
# A/B between alternative builds of the library: bash profiles/ab_lib4.sh librtr_hip.so librtr_hip_pairs.so ...  (in-flight rate, frame alone, kernels)
cd $GRAFT_REPO_ROOT/realtimeraytracer_amd
cp librtr_hip.so /tmp/librtr_hip_base.so
for round in ${ROUNDS:-1 2 3}; do for v in "$@"; do
  if [ "$v" = "librtr_hip.so" ]; then cp /tmp/librtr_hip_base.so librtr_hip.so; else cp $v librtr_hip.so; fi
  echo -n "[$round] $v : "
  (cd .. && timeout -k 5 90 python bench.py --steps ${STEPS:-96} --warmup ${WARMUP:-16} --no-cpu-baseline --present-frames 0 $BENCH_EXTRA 2>/dev/null | python3 -c "
import sys,json; j=json.loads(sys.stdin.readlines()[-1])
print(j['value'], 'ms/frame', j['ms_per_step'], 'per frame in the launches', (j.get('kernels_ms_in_flight_event_brackets') or {}), 'alone', j['one_frame_at_a_time']['ms_per_step'], j['kernels_ms'])")
done; done
cp /tmp/librtr_hip_base.so librtr_hip.so
