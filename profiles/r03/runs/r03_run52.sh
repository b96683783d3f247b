#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
for i in 1 2 3; do python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --present-frames 0 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print('ms/frame', j['ms_per_step'], j['value'], 'launch ms', r['avg_launch_ms'], 'clk', r['clock_mhz'], 'inflight clk', r['clock_mhz_in_flight'], 'achieved', r['achieved'], 'peak', r['peak'], 'frac', r['frac'])
"; done 2>&1 | tee gpurun_out/r03/bench_driver_cmd_clock_check.log
