#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batched or cornell_256 or sponza_class_parity or deep_stack" > gpurun_out/r03/pytest12.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03/pytest12.log
show='import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], j["value"], "Mrays/s", j["ms_per_step"], "ms/step", j["kernels_ms_in_flight_event_brackets"] or j["kernels_ms"])'
for round in 1 2; do
for v in "1 4" "2 4" "2 8" "4 4" "4 8"; do set -- $v
  timeout -k 5 120 python bench.py --steps 192 --warmup 16 --batch $1 --frames-in-flight $2 --isolated-frames 0 --present-frames 0 --no-cpu-baseline 2>/dev/null | python3 -c "$show" "[$round] N=1 batch $1 F=$2"
done
for v in "1 8" "2 8" "4 8"; do set -- $v
  GPU_MAX_HW_QUEUES=8 timeout -k 5 120 python bench.py --steps 192 --warmup 16 --emulate-rank-of 8 --batch $1 --frames-in-flight $2 --isolated-frames 0 --present-frames 0 --no-cpu-baseline 2>/dev/null | python3 -c "$show" "[$round] one rank of 8, batch $1 F=$2"
done
done > gpurun_out/r03/ab_frame_batch.log 2>&1
cat gpurun_out/r03/ab_frame_batch.log | cut -c1-230
