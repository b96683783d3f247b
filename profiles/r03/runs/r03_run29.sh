#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
for c in 1 2 3 5; do python3 bench.py --config $c --verify --present-frames 0 > gpurun_out/r03/bench_config${c}_r03_3_b16.log 2>&1; echo "config $c rc $?"; done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03/bench_config*_b16.log")):
    for l in open(f):
        if l.startswith('{'):
            j = json.loads(l)
            print(f.split('/')[-1], j['ms_per_step'], j['value'], j.get('timed_launches', [0])[:2], j.get('verify'), (j.get('cpu_baseline') or {}).get('value'))
PY
