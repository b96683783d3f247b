#!/bin/bash
# one-off soak of the randomised GPU parity tests on the final binaries of round 3 (other seeds than the committed regression set)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
L=gpurun_out/r03/fuzz_soak_r03_4.log; : > $L
run() { desc=$1; shift; r=$(env "$@" timeout -k 10 400 python -m pytest tests/test_gpu_fuzz.py -m gpu -q 2>&1 | tail -1); echo "$desc: $r" | tee -a $L; }
run "RTR_FUZZ_SEEDS=11000-11300 RTR_TRACE_BINNED=1" RTR_FUZZ_SEEDS=11000-11300 RTR_TRACE_BINNED=1
run "RTR_FUZZ_SEEDS=11400-11700 RTR_TRACE_BINNED=0" RTR_FUZZ_SEEDS=11400-11700 RTR_TRACE_BINNED=0
run "RTR_FUZZ_SEEDS=11800-12000 RTR_TRACE_VIS_FILL=0" RTR_FUZZ_SEEDS=11800-12000 RTR_TRACE_VIS_FILL=0
run "RTR_FUZZ_SEEDS=12100-12300 RTR_BVH_REINSERT_PASSES=2" RTR_FUZZ_SEEDS=12100-12300 RTR_BVH_REINSERT_PASSES=2
run "RTR_FUZZ_SEEDS=12400-12600 RTR_BVH_WIDE_GREEDY=1 RTR_QUEUE_NT=0" RTR_FUZZ_SEEDS=12400-12600 RTR_BVH_WIDE_GREEDY=1 RTR_QUEUE_NT=0
r=$(RTR_FUZZ_SEEDS=1400-2000 timeout -k 10 300 python -m pytest tests/test_gpu_bvh.py -m gpu -q -k refit_random 2>&1 | tail -1); echo "RTR_FUZZ_SEEDS=1400-2000 tests/test_gpu_bvh.py -k refit_random: $r" | tee -a $L
