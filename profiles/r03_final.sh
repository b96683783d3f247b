#!/bin/bash
# The round's closing run on one MI355X: GPU tests, counter passes and kernel statistics of the bench command, the bench lines.  TAG = kernel revision.
cd $GRAFT_REPO_ROOT
TAG=${1:-r03_5}
mkdir -p gpurun_out/r03
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_gpu_$TAG.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r03/pytest_gpu_$TAG.log
bash profiles/pmc_r03.sh ${TAG}_b20 > gpurun_out/r03/pmc_${TAG}_b20.log 2>&1; echo "pmc b20 rc $?"
STEPS=64 WARMUP=32 bash profiles/pmc_r03.sh ${TAG}_b32 > gpurun_out/r03/pmc_${TAG}_b32.log 2>&1; echo "pmc b32 rc $?"
bash profiles/stats_r03.sh ${TAG}_b20s > gpurun_out/r03/stats_${TAG}_b20.log 2>&1; echo "stats rc $?"
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 20 --warmup 5 --verify > gpurun_out/r03/bench_driver_cmd_$TAG.log 2>&1; echo "bench driver rc $?"
python3 bench.py > gpurun_out/r03/bench_default_$TAG.log 2>&1; echo "bench default rc $?"
for n in 2 4 8; do python3 bench.py --emulate-rank-of $n --present-frames 0 --no-cpu-baseline > gpurun_out/r03/bench_rank0_of_${n}_$TAG.log 2>&1; echo "emu $n rc $?"; done
for c in 1 2 3 5; do python3 bench.py --config $c --verify --present-frames 0 > gpurun_out/r03/bench_config${c}_$TAG.log 2>&1; echo "config $c rc $?"; done
RTR_BENCH_FORCE_INPROC=1 RTR_MGPU_SELF_EXCHANGE=1 timeout -k 10 300 python3 bench.py --verify --present-frames 0 > gpurun_out/r03/bench_inproc_one_rank_rccl_$TAG.log 2>&1; echo "inproc rc $?"
python3 bench.py --batch 1 --frames-in-flight 4 --present-frames 0 --no-cpu-baseline > gpurun_out/r03/bench_one_per_launch_4_in_flight_$TAG.log 2>&1; echo "b1 rc $?"
python3 - $TAG <<'PY'
import json, glob, sys
for f in sorted(glob.glob(f"gpurun_out/r03/bench_*_{sys.argv[1]}.log")):
    for l in open(f):
        if l.startswith('{'):
            j = json.loads(l); r = j.get('roofline') or {}
            print(f.split('/')[-1], j['ms_per_step'], j['value'], j.get('timed_launches', [0])[:2], 'frac', r.get('frac'), 'launch ms', r.get('avg_launch_ms'), 'clk', r.get('clock_mhz'), r.get('pmc_note'), (j.get('verify') or {}), j.get('kernels_ms_in_flight_event_brackets'), (j.get('one_frame_at_a_time') or {}).get('ms_per_step'), (j.get('presented_frame') or {}).get('ms_per_frame'))
PY
