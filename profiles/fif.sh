#!/bin/bash
# frames-in-flight sweep of the default bench (N=1):  bash profiles/fif.sh
cd $GRAFT_REPO_ROOT
for f in 1 2 3 4 6; do
  echo -n "frames in flight $f : "
  timeout -k 5 100 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --frames-in-flight $f 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(j['value'], 'Mrays/s', j['ms_per_step'], 'ms/step', j['kernels_ms'])"
done
