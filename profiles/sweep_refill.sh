#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 8 16 20 28 40 64; do for b in 256 1024 4096; do
  echo -n "refill=$r batch=$b : "
  RTR_TRACE_REFILL=$r RTR_TRACE_BATCH=$b timeout -k 5 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(j['value'], j['kernels_ms'])"
done; done
