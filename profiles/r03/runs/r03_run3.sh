#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
bash profiles/ab3.sh "X=0" "RTR_TRACE_DEFER=1" "RTR_QUEUE_NT=1" "RTR_TRACE_DEFER=1 RTR_QUEUE_NT=1" "RTR_TRACE_DEFER=1 RTR_TRACE_INNER_MIN=20" "RTR_TRACE_DEFER=1 RTR_TRACE_INNER_MIN=36" > gpurun_out/r03/ab_defer_nt.log 2>&1
cat gpurun_out/r03/ab_defer_nt.log
RTR_TRACE_DEFER=1 timeout -k 5 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r03/pytest_defer.log 2>&1; echo "pytest(defer) rc=$?"; tail -3 gpurun_out/r03/pytest_defer.log
RTR_QUEUE_NT=1 bash profiles/pmc_one.sh w3nt WRITE_SIZE | grep -E "trace4<16, true, false|gen_oct"
