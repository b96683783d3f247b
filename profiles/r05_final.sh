#!/bin/bash
# Round 5, closing runs on one box: the GPU test suite, then the bench lines kept under profiles/r05/.
mkdir -p gpurun_out/r05
TAG=${1:-r05_1}
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r05/pytest_gpu_$TAG.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r05/pytest_gpu_$TAG.log
show() { python3 -c "
import sys,json
j=json.loads([l for l in open('$1') if l.startswith('{')][-1]); r=j.get('roofline') or {}
print('$1', '| ms/step', j['ms_per_step'], '| Grays/s', round(j['value']/1e3,3), '| frames/launch', j['frames_per_launch'], '| latency ms', j['frame_latency_ms'], '| one at a time', (j.get('one_frame_at_a_time') or {}).get('ms_per_step'),
      '| frac', r.get('frac'), 'useful', r.get('frac_useful'), 'pmc', r.get('pmc_key'), r.get('pmc_scaled_by'), '| gen_oct', (j.get('roofline_secondary') or {}).get('frac'), '| verify', j.get('verify'), '| cpu', (j.get('cpu_baseline') or {}).get('value'))
"; }
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench_driver_cmd_$TAG.log 2>/dev/null; echo "driver cmd rc $?"; show gpurun_out/r05/bench_driver_cmd_$TAG.log
python3 bench.py --verify > gpurun_out/r05/bench_default_$TAG.log 2>/dev/null; echo "default rc $?"; show gpurun_out/r05/bench_default_$TAG.log
for c in 1 2 3 5; do
  python3 bench.py --config $c --verify $( [ $c = 5 ] && echo "--steps 3 --warmup 1" ) > gpurun_out/r05/bench_config${c}_$TAG.log 2>/dev/null; echo "config $c rc $?"; show gpurun_out/r05/bench_config${c}_$TAG.log
done
python3 bench.py --workload sponza_mixed --verify > gpurun_out/r05/bench_sponza_mixed_$TAG.log 2>/dev/null; echo "mixed rc $?"; show gpurun_out/r05/bench_sponza_mixed_$TAG.log
python3 bench.py --batch 32 --max-latency-ms 0 --no-cpu-baseline --present-frames 0 > gpurun_out/r05/bench_batch32_$TAG.log 2>/dev/null; show gpurun_out/r05/bench_batch32_$TAG.log
# N > 1 bring-up rehearsed on this one GPU (test build + RCCL double): the verified first launches, and the fallback to one group per slot
FAKE=tests/fake_rccl/libfake_rccl.so
for n in 2 4 8; do
  LD_PRELOAD=$FAKE RTR_MGPU_TEST_SHARED_DEVICE=1 python3 bench.py --gpus $n --steps 24 --warmup 8 > gpurun_out/r05/bench_rehearse_n${n}_$TAG.log 2> gpurun_out/r05/bench_rehearse_n${n}_$TAG.err; echo "rehearsal N=$n rc $?"
  grep "N>1 start" gpurun_out/r05/bench_rehearse_n${n}_$TAG.err | cut -c1-200
  python3 -c "
import json
j=json.loads([l for l in open('gpurun_out/r05/bench_rehearse_n${n}_$TAG.log') if l.startswith('{')][-1])
print('  N=$n', 'ms/step', j['ms_per_step'], 'frames/launch', j['frames_per_launch'], 'launches in flight', j['latency']['launches_in_flight'], 'latency ms', j['frame_latency_ms'], 'rccl', {k: j['rccl'][k] for k in ('nranks','first_exchange_verified','first_batch_frames','group_per_slot','host_enqueue_ms_per_frame')}, 'verify', j['verify'])
"
done
LD_PRELOAD=$FAKE RTR_MGPU_TEST_SHARED_DEVICE=1 RTR_MGPU_TEST_WRONG_PLACE=2 python3 bench.py --gpus 4 --steps 12 --warmup 4 > gpurun_out/r05/bench_rehearse_fallback_$TAG.log 2> gpurun_out/r05/bench_rehearse_fallback_$TAG.err; echo "fallback rehearsal rc $?"; grep "N>1 start" gpurun_out/r05/bench_rehearse_fallback_$TAG.err | cut -c1-220
LD_PRELOAD=$FAKE RTR_MGPU_TEST_SHARED_DEVICE=1 RTR_MGPU_TEST_WRONG_PLACE=1 python3 bench.py --gpus 4 --steps 12 --warmup 4 > gpurun_out/r05/bench_rehearse_wrong_$TAG.log 2> gpurun_out/r05/bench_rehearse_wrong_$TAG.err; echo "wrong-exchange rehearsal rc $? (4 expected)"; grep "N>1 start" gpurun_out/r05/bench_rehearse_wrong_$TAG.err | cut -c1-220; cat gpurun_out/r05/bench_rehearse_wrong_$TAG.log | cut -c1-400
