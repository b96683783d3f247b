#!/bin/bash
# round 3, final measurements of kernel revision r03.3 with 8 frames per launch: counter passes, kernel stats, bench lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
bash profiles/pmc_r03.sh r03i > gpurun_out/r03/pmc_r03i.log 2>&1; tail -2 gpurun_out/r03/pmc_r03i.log
cd $GRAFT_REPO_ROOT
bash profiles/stats_r03.sh r03v > gpurun_out/r03/stats_r03v.log 2>&1; head -12 gpurun_out/r03/stats_r03v.log | cut -c1-180
cd $GRAFT_REPO_ROOT
cp profiles/r03/pmc_roofline.json gpurun_out/r03/pmc_roofline_single.json
python - <<'PY'
import json, subprocess, sys
a = json.loads(subprocess.run([sys.executable, "profiles/make_pmc_json.py", "gpurun_out/prof_r03i/pmc.json", "sponza_class_1920x1080_spp1_gpus1", "r03.3", "24808565", "2073600", "262148", "8"], capture_output=True, text=True, check=True).stdout)
b = json.load(open("profiles/r03/pmc_roofline.json"))
b.update({k: v for k, v in a.items() if k != "_comment"})
json.dump(b, open("profiles/r03/pmc_roofline.json", "w"), indent=1)
json.dump(b, open("gpurun_out/r03/pmc_roofline.json", "w"), indent=1)
PY
python bench.py --verify > gpurun_out/r03/bench_default_r03_3b.log 2> gpurun_out/r03/bench_default_r03_3b.err; echo "default rc=$?"
for c in 1 2 3 5; do python bench.py --config $c --verify --steps 48 > gpurun_out/r03/bench_config${c}_r03_3b.log 2> gpurun_out/r03/bench_config${c}_r03_3b.err; echo "config $c rc=$?"; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03/bench_*_r03_3b.log')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); r=d.get('roofline') or {}
        print(f, d['value'], d['ms_per_step'], d.get('frames_per_launch'), d.get('verify'), r.get('frac'), r.get('avg_launch_ms'), r.get('hbm_frac'), r.get('pmc_note'), (d.get('roofline_secondary') or {}).get('frac'), (d.get('frame_hbm') or {}).get('frac_of_hbm_peak'))
    except Exception as e: print(f, 'ERR', e)
PY
