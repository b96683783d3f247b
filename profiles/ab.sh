#!/bin/bash
# A/B helper: bash profiles/ab.sh "ENV1=.. ENV2=.." "ENV..." ...   (each argument = one variant's environment)
cd $GRAFT_REPO_ROOT
for round in 1 2; do for v in "$@"; do
  echo -n "[$round] $v : "
  env $v timeout -k 5 60 python bench.py --steps 15 --warmup 3 --no-cpu-baseline $BENCH_EXTRA 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(j['value'], j['kernels_ms'])"
done; done
