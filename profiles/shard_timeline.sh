#!/bin/bash
# Kernel timeline of ONE rank of 8 (shard 0, 4 frames in flight): how busy is the GPU, what overlaps what.
#   bash profiles/shard_timeline.sh   -> gpurun_out/shard_timeline.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_shard8
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 170 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 $REPO/bench.py --steps 40 --warmup 6 --emulate-rank-of ${1:-8} --isolated-frames 0 --present-frames 0 > "$OUT/bench.log" 2>&1
python3 - "$OUT" <<'PY' > $REPO/gpurun_out/shard_timeline.txt
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "rtrdev" in r["Kernel_Name"] and "true>(rtrdev::DeviceScene, rtrdev::RenderA" not in r["Kernel_Name"] and "k_resolve<true" not in r["Kernel_Name"] and "count" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the steady part: drop the first and last 20 %
n = len(rows); rows = rows[n // 5: n - n // 5]
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
ev = []
tot = collections.defaultdict(int); cnt = collections.defaultdict(int)
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((s, 1)); ev.append((e, -1))
    k = r["Kernel_Name"].split("(")[0].replace("void rtrdev::", "").replace("rtrdev::", "")
    tot[k] += e - s; cnt[k] += 1
ev.sort()
busy = 0; depth = 0; last = None; weighted = 0
for t, d in ev:
    if depth > 0: busy += t - last; weighted += (t - last) * depth
    depth += d; last = t
span = t1 - t0
print(f"steady window {span/1e6:.3f} ms, {len(rows)} kernel dispatches; some kernel running {100.0*busy/span:.1f} % of the time; mean kernels in flight while busy {weighted/max(busy,1):.2f}")
for k in sorted(tot, key=lambda k: -tot[k]):
    print(f"  {k:40s} n={cnt[k]:4d} mean begin->end {tot[k]/cnt[k]/1e3:8.1f} us   sum/window {100.0*tot[k]/span:6.1f} %")
PY
cat $REPO/gpurun_out/shard_timeline.txt
