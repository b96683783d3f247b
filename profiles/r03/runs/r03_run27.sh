#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_run27.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r03/pytest_run27.log
bash profiles/pmc_r03.sh r03_3_b10 > gpurun_out/r03/pmc_b10.log 2>&1; echo "pmc b10 rc $?"
STEPS=32 WARMUP=16 bash profiles/pmc_r03.sh r03_3_b16 > gpurun_out/r03/pmc_b16.log 2>&1; echo "pmc b16 rc $?"
bash profiles/stats_r03.sh r03_3_b10s > gpurun_out/r03/stats_b10.log 2>&1; echo "stats rc $?"
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r03/bench_driver_cmd_r03_3.log 2>&1; echo "bench driver rc $?"
python3 bench.py > gpurun_out/r03/bench_default_r03_3_b16.log 2>&1; echo "bench default rc $?"
tail -c 400 gpurun_out/r03/pmc_b10.log
