#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_r8
cd /tmp && export TMPDIR=/tmp
GPU_MAX_HW_QUEUES=8 timeout -k 10 170 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r8/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 16 --emulate-rank-of 8 --frames-in-flight 8 --isolated-frames 0 --present-frames 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r8/bench.log 2> $GRAFT_REPO_ROOT/gpurun_out/prof_r8/rocprof.log
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_r8/bench.log | python3 -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(j['value'], j['ms_per_step'])"
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_r8/stats -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]: print(r['Name'][:60], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'])
PY
t=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_r8/stats -name "*kernel_trace.csv" | head -1); python3 - "$t" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
ev=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][:40]) for r in rows]
ev.sort()
# take the last 60% of the run as steady state
t0=ev[len(ev)*4//10][0]; t1=ev[-1][1]
# busy time = union of intervals; concurrency = sum durations / wall
ints=[(s,e) for s,e,_ in ev if s>=t0]
ints.sort(); busy=0; cs,ce=ints[0]
for s,e in ints[1:]:
    if s>ce: busy+=ce-cs; cs,ce=s,e
    else: ce=max(ce,e)
busy+=ce-cs
tot=sum(e-s for s,e in ints)
print("steady window %.2f ms: some kernel running %.1f %% of it; mean kernels resident %.2f" % ((t1-t0)/1e6, 100*busy/(t1-t0), tot/(t1-t0)))
PY
