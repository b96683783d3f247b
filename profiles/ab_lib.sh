#!/bin/bash
# A/B between alternative builds of the library: bash profiles/ab_lib.sh librtr_hip.so librtr_hip_u2.so ...
cd $GRAFT_REPO_ROOT/realtimeraytracer_amd
cp librtr_hip.so /tmp/librtr_hip_base.so
for round in 1 2; do for v in "$@"; do
  if [ "$v" = "librtr_hip.so" ]; then cp /tmp/librtr_hip_base.so librtr_hip.so; else cp $v librtr_hip.so; fi
  echo -n "[$round] $v : "
  (cd .. && timeout -k 5 60 python bench.py --steps 15 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(j['value'], j['kernels_ms'])")
done; done
cp /tmp/librtr_hip_base.so librtr_hip.so
