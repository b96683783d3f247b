#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest1.log 2>&1; echo "pytest rc=$?" ; tail -3 gpurun_out/r03/pytest1.log
python bench.py --steps 200 --warmup 8 > gpurun_out/r03/bench1.json 2> gpurun_out/r03/bench1.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03/bench1.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['kernels_ms'], d['one_frame_at_a_time'])
PY
bash profiles/pmc_one.sh w1 WRITE_SIZE FETCH_SIZE
