#!/bin/bash
# bench.py's in-process N > 1 path with N ranks sharing ONE GPU (test hook + tests/fake_rccl in front of RCCL): a functional rehearsal
# of `python bench.py --gpus N` and a measure of what the exchange machinery costs (the rate is one GPU's, time-sliced by N ranks).
cd ${GRAFT_REPO_ROOT:-.}
for n in 1 2 4 8; do
  if [ $n = 1 ]; then python3 bench.py --steps 96 --warmup 16 --present-frames 0 --no-cpu-baseline --isolated-frames 0 2>/dev/null
  else LD_PRELOAD=$PWD/tests/fake_rccl/libfake_rccl.so RTR_MGPU_TEST_SHARED_DEVICE=1 python3 bench.py --gpus $n --steps 96 --warmup 32 2>/dev/null; fi | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('ranks', j['n_gpus'], 'ms/frame', j['ms_per_step'], j['value'], j['unit'], 'slots', j.get('frames_in_flight'), 'per launch', j.get('frames_per_launch'), (j.get('rccl') or {}).get('version'), 'host enqueue ms/frame', (j.get('rccl') or {}).get('host_enqueue_ms_per_frame'), 'inside rccl calls', (j.get('rccl') or {}).get('of_which_inside_rccl_calls'), j.get('verify'), (j.get('rehearsal') or '')[:40])
"
done
# the same with sixteen shards per launch given explicitly (round 3's launches): what ONE exchange per launch does to the ranks' host time
for n in 4 8; do
  LD_PRELOAD=$PWD/tests/fake_rccl/libfake_rccl.so RTR_MGPU_TEST_SHARED_DEVICE=1 python3 bench.py --gpus $n --steps 96 --warmup 32 --batch 16 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('ranks', j['n_gpus'], '--batch 16: ms/frame', j['ms_per_step'], 'slots', j.get('frames_in_flight'), 'per launch', j.get('frames_per_launch'), 'host enqueue ms/frame', (j.get('rccl') or {}).get('host_enqueue_ms_per_frame'), 'inside rccl calls', (j.get('rccl') or {}).get('of_which_inside_rccl_calls'), j.get('verify'))
"
done
