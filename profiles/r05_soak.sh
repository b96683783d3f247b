#!/bin/bash
# one-off soak of the randomised GPU parity tests on the binaries of round 5 (other seeds than the committed regression set); SOAK_BASE shifts the seeds
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
B=${SOAK_BASE:-15000}
L=gpurun_out/r05/fuzz_soak_r05_$B.log; : > $L
run() { desc=$1; shift; r=$(env "$@" timeout -k 10 500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q 2>&1 | tail -1); echo "$desc: $r" | tee -a $L; }
s() { echo "$((B + $1))-$((B + $2))"; }
run "RTR_FUZZ_SEEDS=$(s 0 300) RTR_TRACE_BINNED=1" RTR_FUZZ_SEEDS=$(s 0 300) RTR_TRACE_BINNED=1
run "RTR_FUZZ_SEEDS=$(s 400 700) RTR_TRACE_BINNED=0" RTR_FUZZ_SEEDS=$(s 400 700) RTR_TRACE_BINNED=0
run "RTR_FUZZ_SEEDS=$(s 800 1100) RTR_PRIMARY_PACKET=1 (camera rays as 8x8 packets, oracle restating that walk)" RTR_FUZZ_SEEDS=$(s 800 1100) RTR_PRIMARY_PACKET=1
run "RTR_FUZZ_SEEDS=$(s 1200 1400) RTR_BVH_REINSERT_PASSES=2" RTR_FUZZ_SEEDS=$(s 1200 1400) RTR_BVH_REINSERT_PASSES=2
run "RTR_FUZZ_SEEDS=$(s 1500 1700) RTR_BVH_WIDE_GREEDY=1 RTR_QUEUE_NT=0" RTR_FUZZ_SEEDS=$(s 1500 1700) RTR_BVH_WIDE_GREEDY=1 RTR_QUEUE_NT=0
run "RTR_FUZZ_SEEDS=$(s 1800 2000) RTR_PRIMARY_WIDE=1 (camera rays over the 4-wide view, oracle restating that walk)" RTR_FUZZ_SEEDS=$(s 1800 2000) RTR_PRIMARY_WIDE=1
r=$(RTR_FUZZ_SEEDS=$((B / 5 + 100))-$((B / 5 + 700)) timeout -k 10 300 python -m pytest tests/test_gpu_bvh.py -m gpu -q -k refit_random 2>&1 | tail -1); echo "RTR_FUZZ_SEEDS=$((B / 5 + 100))-$((B / 5 + 700)) tests/test_gpu_bvh.py -k refit_random: $r" | tee -a $L
run "RTR_FUZZ_SEEDS=$(s 2100 2400) RTR_TRACE_OWN_LEAF=0 (every shadow ray from the root; the oracle told so)" RTR_FUZZ_SEEDS=$(s 2100 2400) RTR_TRACE_OWN_LEAF=0
run "RTR_FUZZ_SEEDS=$(s 2500 2800) RTR_RESOLVE_COMPACT=0 (per-pixel resolve)" RTR_FUZZ_SEEDS=$(s 2500 2800) RTR_RESOLVE_COMPACT=0
