#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest14.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03/pytest14.log
python bench.py --verify > gpurun_out/r03/bench14.log 2> gpurun_out/r03/bench14.err; echo "bench rc=$?"
RTR_BENCH_FORCE_INPROC=1 RTR_MGPU_SELF_EXCHANGE=1 python bench.py --steps 96 > gpurun_out/r03/bench14_inproc.log 2> gpurun_out/r03/bench14_inproc.err; echo "inproc rc=$?"
RTR_BENCH_FORCE_DIST=1 python bench.py --steps 96 --isolated-frames 2 > gpurun_out/r03/bench14_dist.log 2> gpurun_out/r03/bench14_dist.err; echo "dist rc=$?"
python - <<'PY'
import json
for f in ('bench14','bench14_inproc','bench14_dist'):
    try:
        d=json.loads([l for l in open(f'gpurun_out/r03/{f}.log') if l.startswith('{')][-1])
        print(f, d['value'], d['ms_per_step'], d.get('frames_in_flight'), d.get('frames_per_launch'), d.get('verify'), (d.get('roofline') or {}).get('frac'), d.get('kernels_ms'), d.get('kernels_ms_in_flight_event_brackets'))
    except Exception as e: print(f, 'ERR', e); print(open(f'gpurun_out/r03/{f}.err').read()[-1500:])
PY
