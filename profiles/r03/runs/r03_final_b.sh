#!/bin/bash
# round 3, final measurements (b): counter passes + kernel stats of the final kernels, then the bench lines of every BASELINE config
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
bash profiles/pmc_r03.sh r03g > gpurun_out/r03/pmc_r03g.log 2>&1; tail -3 gpurun_out/r03/pmc_r03g.log
cd $GRAFT_REPO_ROOT
bash profiles/stats_r03.sh r03t > gpurun_out/r03/stats_r03t.log 2>&1; tail -2 gpurun_out/r03/stats_r03t.log
cd $GRAFT_REPO_ROOT
python bench.py --verify > gpurun_out/r03/bench_default_r03_2.log 2> gpurun_out/r03/bench_default_r03_2.err; echo "default rc=$?"
for c in 1 2 3 5; do python bench.py --config $c --verify --steps 40 > gpurun_out/r03/bench_config${c}_r03_2.log 2> gpurun_out/r03/bench_config${c}_r03_2.err; echo "config $c rc=$?"; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03/bench_*_r03_2.log')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f, d['value'], d['ms_per_step'], d.get('verify'), d['kernels_ms'])
    except Exception as e: print(f, 'ERR', e)
PY
