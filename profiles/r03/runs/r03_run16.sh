#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest16.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r03/pytest16.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03/bench16_driver_cmd.log 2> gpurun_out/r03/bench16_driver_cmd.err; echo "driver-cmd rc=$?"
python3 bench.py > gpurun_out/r03/bench16_default.log 2> gpurun_out/r03/bench16_default.err; echo "default rc=$?"
RTR_BENCH_FORCE_INPROC=1 RTR_MGPU_SELF_EXCHANGE=1 python bench.py --steps 20 --warmup 5 > gpurun_out/r03/bench16_inproc.log 2> gpurun_out/r03/bench16_inproc.err; echo "inproc rc=$?"
RTR_BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --isolated-frames 2 > gpurun_out/r03/bench16_dist.log 2> gpurun_out/r03/bench16_dist.err; echo "dist rc=$?"
python - <<'PY'
import json
for f in ('bench16_driver_cmd','bench16_default','bench16_inproc','bench16_dist'):
    try:
        d=json.loads([l for l in open(f'gpurun_out/r03/{f}.log') if l.startswith('{')][-1]); r=d.get('roofline') or {}
        print(f, d['value'], d['ms_per_step'], d.get('frames_per_launch'), d.get('verify'), r.get('frac'), r.get('avg_launch_ms'), r.get('pmc_note'), (d.get('cpu_baseline') or {}).get('value'))
    except Exception as e: print(f, 'ERR', e); print(open(f'gpurun_out/r03/{f}.err').read()[-1500:])
PY
