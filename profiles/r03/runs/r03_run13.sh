#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
show='import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], j["value"], "Mrays/s", j["ms_per_step"], "ms/step")'
for round in 1 2; do
for v in "4 4" "8 8" "4 12" "8 16" "2 2"; do set -- $v
  timeout -k 5 120 python bench.py --steps 192 --warmup 16 --batch $1 --frames-in-flight $2 --isolated-frames 0 --present-frames 0 --no-cpu-baseline 2>/dev/null | python3 -c "$show" "[$round] N=1 batch $1 F=$2"
done
for n in 8 4 2; do for v in "4 4" "4 8" "8 8" "2 4"; do set -- $v
  GPU_MAX_HW_QUEUES=8 timeout -k 5 120 python bench.py --steps 192 --warmup 16 --emulate-rank-of $n --batch $1 --frames-in-flight $2 --isolated-frames 0 --present-frames 0 --no-cpu-baseline 2>/dev/null | python3 -c "$show" "[$round] one rank of $n, batch $1 F=$2"
done; done
done > gpurun_out/r03/ab_frame_batch2.log 2>&1
cat gpurun_out/r03/ab_frame_batch2.log
