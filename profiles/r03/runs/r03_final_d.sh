#!/bin/bash
# round 3, final measurements of kernel revision r03.3: GPU tests, counter passes, kernel stats, bench lines of every BASELINE config
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -x -q > gpurun_out/r03/pytest_gpu_r03_3.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r03/pytest_gpu_r03_3.log
bash profiles/pmc_r03.sh r03h > gpurun_out/r03/pmc_r03h.log 2>&1; tail -2 gpurun_out/r03/pmc_r03h.log
cd $GRAFT_REPO_ROOT
bash profiles/stats_r03.sh r03u > gpurun_out/r03/stats_r03u.log 2>&1; tail -1 gpurun_out/r03/stats_r03u.log
cd $GRAFT_REPO_ROOT
python profiles/make_pmc_json.py gpurun_out/prof_r03h/pmc.json sponza_class_1920x1080_spp1_gpus1 r03.3 24808565 2073600 262148 > profiles/r03/pmc_roofline.json && cp profiles/r03/pmc_roofline.json gpurun_out/r03/pmc_roofline.json
python bench.py --verify > gpurun_out/r03/bench_default_r03_3.log 2> gpurun_out/r03/bench_default_r03_3.err; echo "default rc=$?"
for c in 1 2 3 5; do python bench.py --config $c --verify --steps 40 > gpurun_out/r03/bench_config${c}_r03_3.log 2> gpurun_out/r03/bench_config${c}_r03_3.err; echo "config $c rc=$?"; done
RTR_BENCH_FORCE_INPROC=1 RTR_MGPU_SELF_EXCHANGE=1 python bench.py --steps 100 > gpurun_out/r03/bench_inproc_one_rank_rccl_r03_3.log 2> gpurun_out/r03/bench_inproc_one_rank_rccl_r03_3.err; echo "inproc rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03/bench_*_r03_3.log')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); r=d.get('roofline') or {}
        print(f, d['value'], d['ms_per_step'], d.get('verify'), d.get('kernels_ms'), r.get('frac'), r.get('hbm_frac'), (d.get('roofline_secondary') or {}).get('frac'), (d.get('frame_hbm') or {}).get('frac_of_hbm_peak'))
    except Exception as e: print(f, 'ERR', e)
PY
