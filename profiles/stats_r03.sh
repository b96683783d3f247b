#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command (round 3): bash profiles/stats_r03.sh <tag>
#   stats        : python bench.py --steps 20 --warmup 5 — the driver's command: one launch of 20 frames at a time (STEPS / WARMUP override)
#   stats_serial : --batch 1 --frames-in-flight 1 — one frame per launch, one frame at a time
# The untimed passes after the timed region (--isolated-frames, --present-frames) are switched off.  A run also launches every kernel
# over ONE frame (each frame object's first render, the counting pass), so rocprofv3's own per-kernel average mixes two kinds of
# launch; kernel_stats.txt therefore lists, from the kernel trace, the launches of the timed kind (duration within 20 % of
# the longest) next to the rest, and keeps rocprofv3's table below.
set -o pipefail
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps ${STEPS:-20} --warmup ${WARMUP:-5} --no-cpu-baseline --isolated-frames 0 --present-frames 0"
timeout -k 10 170 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH > "$OUT/bench_stats.log" 2> "$OUT/rocprof_stats.log" || echo "stats failed"
timeout -k 10 170 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_serial" -- $BENCH --batch 1 --frames-in-flight 1 > "$OUT/bench_stats_serial.log" 2> "$OUT/rocprof_stats_serial.log" || echo "stats_serial failed"
python3 - "$OUT" > "$OUT/kernel_stats.txt" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for d, what in (("stats", "bench command: equal launches of several frames, one launch at a time"), ("stats_serial", "--batch 1 --frames-in-flight 1: one frame per launch, one at a time")):
    tr = glob.glob(os.path.join(out, d, "**", "*kernel_trace.csv"), recursive=True)
    print(f"== {d}: {what}")
    if tr:
        dur = defaultdict(list)
        for row in csv.DictReader(open(tr[0])):
            dur[row["Kernel_Name"]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6)
        print(f"{'kernel':58s} {'launches of the timed kind':>28s} {'avg ms':>9s} {'min':>8s} {'max':>8s}   | other launches (one frame: set-up, counting pass)")
        for k in sorted(dur, key=lambda k: -sum(dur[k])):
            if "rtrdev" not in k: continue
            v = dur[k]; m = max(v); big = [x for x in v if x >= 0.8 * m]; small = [x for x in v if x < 0.8 * m]
            short = k.split("(")[0].replace("void ", "")[:58]
            print(f"{short:58s} {len(big):28d} {sum(big)/len(big):9.4f} {min(big):8.4f} {max(big):8.4f}   | {len(small)} launches" + (f", avg {sum(small)/len(small):.4f} ms" if small else ""))
    f = glob.glob(os.path.join(out, d, "**", "*kernel_stats.csv"), recursive=True)
    if f:
        print(f"-- rocprofv3's own table ({os.path.basename(f[0])}): averages over ALL launches of a kernel")
        sys.stdout.write(open(f[0]).read())
PY
cat "$OUT/kernel_stats.txt" | cut -c1-200 | head -40
