#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_gpu_r03_6_final.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r03/pytest_gpu_r03_6_final.log
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r03/bench_driver_cmd_r03_6_final.log 2>&1; echo "bench rc $?"; tail -c 300 gpurun_out/r03/bench_driver_cmd_r03_6_final.log
