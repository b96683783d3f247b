#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
python3 bench.py --steps 20 --warmup 5 --verify > gpurun_out/r03/bench_driver_cmd_r03_4.log 2>&1; echo "bench driver rc $?"
python3 bench.py > gpurun_out/r03/bench_default_r03_4.log 2>&1; echo "bench default rc $?"
for n in 2 4 8; do python3 bench.py --emulate-rank-of $n --present-frames 0 --no-cpu-baseline > gpurun_out/r03/bench_rank0_of_${n}_r03_4.log 2>&1; echo "emu $n rc $?"; done
for c in 1 2 3 5; do python3 bench.py --config $c --verify --present-frames 0 > gpurun_out/r03/bench_config${c}_r03_4.log 2>&1; echo "config $c rc $?"; done
RTR_BENCH_FORCE_INPROC=1 RTR_MGPU_SELF_EXCHANGE=1 timeout -k 10 300 python3 bench.py --verify --present-frames 0 > gpurun_out/r03/bench_inproc_one_rank_rccl_r03_4.log 2>&1; echo "inproc rc $?"
python3 bench.py --batch 1 --frames-in-flight 4 --present-frames 0 --no-cpu-baseline > gpurun_out/r03/bench_one_per_launch_4_in_flight_r03_4.log 2>&1; echo "b1 rc $?"
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03/bench_*_r03_4.log")):
    for l in open(f):
        if l.startswith('{'):
            j = json.loads(l); r = j.get('roofline') or {}
            print(f.split('/')[-1], j['ms_per_step'], j['value'], j.get('timed_launches', [0])[:2], 'fif', j.get('frames_in_flight'), 'frac', r.get('frac'), 'launch ms', r.get('avg_launch_ms'), 'clk', r.get('clock_mhz'), 'l1', r.get('l1_tag_lookups_per_l1_clock'), 'hbm', r.get('hbm_frac'), r.get('pmc_note'), (j.get('verify') or {}), (j.get('roofline_secondary') or {}).get('frac'), (j.get('frame_hbm') or {}).get('frac_of_hbm_peak'), j.get('kernels_ms_in_flight_event_brackets'), (j.get('one_frame_at_a_time') or {}).get('ms_per_step'), (j.get('presented_frame') or {}).get('ms_per_frame'))
PY
