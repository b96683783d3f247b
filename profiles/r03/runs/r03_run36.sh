#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_gpu_r03_4.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r03/pytest_gpu_r03_4.log
bash profiles/pmc_r03.sh r03_4_b10 > gpurun_out/r03/pmc_r03_4_b10.log 2>&1; echo "pmc b10 rc $?"
STEPS=32 WARMUP=16 bash profiles/pmc_r03.sh r03_4_b16 > gpurun_out/r03/pmc_r03_4_b16.log 2>&1; echo "pmc b16 rc $?"
bash profiles/stats_r03.sh r03_4_b10s > gpurun_out/r03/stats_r03_4_b10.log 2>&1; echo "stats rc $?"
grep -A28 "^rtrdev::k_shadow_trace4<16, true, false>" gpurun_out/prof_r03_4_b10/summary.txt | grep "SQ_INSTS_VALU\|SQ_ACTIVE_INST_VALU\|avg_ms\|TCP_TOTAL_CACHE"
