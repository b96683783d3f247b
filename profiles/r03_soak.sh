#!/bin/bash
# one-off soak of the randomised GPU parity tests on the final binaries of round 3 (other seeds than the committed regression set)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
L=gpurun_out/r03/fuzz_soak_r03_6.log; : > $L
run() { desc=$1; shift; r=$(env "$@" timeout -k 10 400 python -m pytest tests/test_gpu_fuzz.py -m gpu -q 2>&1 | tail -1); echo "$desc: $r" | tee -a $L; }
run "RTR_FUZZ_SEEDS=13000-13300 RTR_TRACE_BINNED=1" RTR_FUZZ_SEEDS=13000-13300 RTR_TRACE_BINNED=1
run "RTR_FUZZ_SEEDS=13400-13700 RTR_TRACE_BINNED=0" RTR_FUZZ_SEEDS=13400-13700 RTR_TRACE_BINNED=0
run "RTR_FUZZ_SEEDS=13800-14000 RTR_TRACE_VIS_FILL=0" RTR_FUZZ_SEEDS=13800-14000 RTR_TRACE_VIS_FILL=0
run "RTR_FUZZ_SEEDS=14100-14300 RTR_BVH_REINSERT_PASSES=2" RTR_FUZZ_SEEDS=14100-14300 RTR_BVH_REINSERT_PASSES=2
run "RTR_FUZZ_SEEDS=14400-14600 RTR_BVH_WIDE_GREEDY=1 RTR_QUEUE_NT=0" RTR_FUZZ_SEEDS=14400-14600 RTR_BVH_WIDE_GREEDY=1 RTR_QUEUE_NT=0
r=$(RTR_FUZZ_SEEDS=2100-2700 timeout -k 10 300 python -m pytest tests/test_gpu_bvh.py -m gpu -q -k refit_random 2>&1 | tail -1); echo "RTR_FUZZ_SEEDS=2100-2700 tests/test_gpu_bvh.py -k refit_random: $r" | tee -a $L
