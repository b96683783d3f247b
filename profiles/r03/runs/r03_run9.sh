#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
show='import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], j["value"], "Mrays/s", j["ms_per_step"], "ms/step")'
for n in 8 4 2; do for round in 1 2; do for q in 4 8; do for f in 4 8; do
    GPU_MAX_HW_QUEUES=$q timeout -k 5 120 python bench.py --steps 96 --warmup 12 --emulate-rank-of $n --frames-in-flight $f --isolated-frames 0 --present-frames 0 2>/dev/null | python3 -c "$show" "[$round] one rank of $n hwq $q F=$f"
done; done; done; done > gpurun_out/r03/sweep_hw_queues_one_rank_of_n_r03_3.log 2>&1
cat gpurun_out/r03/sweep_hw_queues_one_rank_of_n_r03_3.log
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mgpu or create_like" 2>&1 | tail -2
