#!/bin/bash
# bash profiles/ab4.sh "ENV.." ... — bench, per variant: in-flight rate, one frame alone, per-kernel ms
cd $GRAFT_REPO_ROOT
for round in 1 2; do for v in "$@"; do
  echo -n "[$round] $v : "
  env $v timeout -k 5 90 python bench.py --steps 100 --warmup 8 --no-cpu-baseline --present-frames ${PRESENT:-0} $BENCH_EXTRA 2>/dev/null | python3 -c "
import sys,json; j=json.loads(sys.stdin.readlines()[-1])
print(j['value'], 'ms/frame', j['ms_per_step'], 'alone', j['one_frame_at_a_time']['ms_per_step'], j['kernels_ms'], 'presented', (j.get('presented_frame') or {}).get('ms_per_frame'))"
done; done
