#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 600 python3 -m pytest tests/test_bench_contract.py tests/test_gpu_parity.py -m gpu -x -q -k "mgpu or bench or batch" > gpurun_out/r03/pytest_run21.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r03/pytest_run21.log
RTR_BENCH_FORCE_INPROC=1 RTR_MGPU_SELF_EXCHANGE=1 timeout -k 10 300 python3 bench.py --frames-in-flight 32 --steps 64 --warmup 32 --verify --present-frames 0 > gpurun_out/r03/bench_inproc_one_rank_32_slots.log 2>&1; echo "inproc rc $?"; tail -c 1500 gpurun_out/r03/bench_inproc_one_rank_32_slots.log
timeout -k 10 300 python3 bench.py --emulate-rank-of 8 --steps 96 --warmup 32 --present-frames 0 > gpurun_out/r03/bench_rank0_of_8_default.log 2>&1; echo "emu rc $?"; tail -c 600 gpurun_out/r03/bench_rank0_of_8_default.log
