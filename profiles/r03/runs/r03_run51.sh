#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
for i in 1 2 3 4; do python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --present-frames 0 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernels_ms_in_flight_event_brackets']; print('driver cmd: ms/frame', j['ms_per_step'], j['value'], 'sum of kernels', round(sum(k.values()),4), (j['roofline'] or {}).get('frac'))
"; done 2>&1 | tee gpurun_out/r03/bench_driver_cmd_marshalled_ahead.log
timeout -k 10 600 python3 -m pytest tests/test_bench_contract.py tests/test_gpu_parity.py -m gpu -x -q -k "bench or batched" 2>&1 | tail -2
