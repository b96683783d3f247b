#!/bin/bash
# A/B with stats: bash profiles/ab3.sh "ENV.." "ENV.." ...  — bench (4 frames in flight + kernels alone) and the counting form's schedule per variant
cd $GRAFT_REPO_ROOT
for round in 1 2; do for v in "$@"; do
  echo -n "[$round] $v : "
  env $v timeout -k 5 90 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --present-frames 0 $BENCH_EXTRA 2>/dev/null | python3 -c "
import sys,json; j=json.loads(sys.stdin.readlines()[-1]); r=j['roofline']; s=r['schedule']
print(j['value'], 'ms/frame', j['ms_per_step'], 'alone', j['one_frame_at_a_time']['ms_per_step'], j['kernels_ms'], 'util', r['lane_util']['node_loop'], r['lane_util']['triangle_loop'], 'trips', s['node_loop_trips'], s['triangle_loop_trips'], 'refills', s['refill_passes'], 'spec', s.get('speculative_visits'))"
done; done
