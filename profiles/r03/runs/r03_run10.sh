#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
for n in 8 4; do
GPU_MAX_HW_QUEUES=8 python bench.py --steps 200 --warmup 16 --emulate-rank-of $n --frames-in-flight 8 --isolated-frames 20 --present-frames 0 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('rank of $n:', j['value'], j['ms_per_step'], 'alone', j['one_frame_at_a_time']['ms_per_step'], j['kernels_ms'], 'sum', round(sum(j['kernels_ms'].values()),4), 'brackets', j['kernels_ms_in_flight_event_brackets'])"
done
cd /tmp && export TMPDIR=/tmp
GPU_MAX_HW_QUEUES=8 timeout -k 10 170 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r8/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 16 --emulate-rank-of 8 --frames-in-flight 8 --isolated-frames 0 --present-frames 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r8/bench.log 2> $GRAFT_REPO_ROOT/gpurun_out/prof_r8/rocprof.log
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_r8/stats -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]: print(r['Name'][:60], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'])
PY
