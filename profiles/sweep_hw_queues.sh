#!/bin/bash
# frames in flight x GPU_MAX_HW_QUEUES (HIP maps streams onto that many hardware queues; default 4) for one rank of N
cd $GRAFT_REPO_ROOT
show='import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], j["value"], "Mrays/s", j["ms_per_step"], "ms/step")'
for n in ${NS:-8}; do for round in 1 2; do for q in 4 8 16; do for f in 4 6 8 12; do
    GPU_MAX_HW_QUEUES=$q timeout -k 5 120 python bench.py --steps 96 --warmup 12 --emulate-rank-of $n --frames-in-flight $f --isolated-frames 0 --present-frames 0 2>/dev/null | python3 -c "$show" "[$round] one rank of $n hwq $q F=$f"
done; done; done; done
