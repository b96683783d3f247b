#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
show() { python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$1', 'ms/frame', j['ms_per_step'], j['value'], 'slots', j.get('frames_in_flight'), 'per launch', j.get('frames_per_launch'), j.get('verify'), (j.get('rccl') or {}).get('host_enqueue_ms_per_frame'))
"; }
{
for rep in 1 2; do
  python3 bench.py --emulate-rank-of 2 --frames-in-flight 32 --steps 192 --warmup 64 --present-frames 0 --no-cpu-baseline --isolated-frames 0 2>/dev/null | show "[rank 0 of 2, one launch of 32 in flight]"
  python3 bench.py --emulate-rank-of 2 --steps 192 --warmup 64 --present-frames 0 --no-cpu-baseline --isolated-frames 0 2>/dev/null | show "[rank 0 of 2, default: two launches of 32 in flight]"
done
LD_PRELOAD=$PWD/tests/fake_rccl/libfake_rccl.so RTR_MGPU_TEST_SHARED_DEVICE=1 python3 bench.py --gpus 2 --steps 128 --warmup 64 2>/dev/null | show "[--gpus 2, ranks sharing the GPU, default]"
LD_PRELOAD=$PWD/tests/fake_rccl/libfake_rccl.so RTR_MGPU_TEST_SHARED_DEVICE=1 python3 bench.py --gpus 2 --frames-in-flight 32 --steps 128 --warmup 64 2>/dev/null | show "[--gpus 2, ranks sharing the GPU, one launch in flight]"
} > gpurun_out/r03/sweep_two_launches_n2.log 2>&1; cat gpurun_out/r03/sweep_two_launches_n2.log
