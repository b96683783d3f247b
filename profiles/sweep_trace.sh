#!/bin/bash
# Sweep of k_shadow_trace's run-time tunables on the bench frame (one frame at a time so the kernel time is clean).
cd $GRAFT_REPO_ROOT
for im in ${IMS:-8 14 20 26 32}; do for r in ${RFS:-16 28 40}; do
  echo -n "inner_min=$im refill=$r : "
  RTR_TRACE_INNER_MIN=$im RTR_TRACE_REFILL=$r timeout -k 5 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --frames-in-flight 1 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(j['value'], j['kernels_ms']['shadow_trace'])"
done; done
