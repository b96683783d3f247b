#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command (round 3): bash profiles/stats_r03.sh <tag>
#   stats        : python bench.py --steps 20 --warmup 3 (4 frames in flight, as the default bench line is measured)
#   stats_serial : the same frames one at a time (--frames-in-flight 1): kernel durations free of cross-frame overlap
# The untimed passes after the timed region (--isolated-frames, --present-frames) are switched off so the averages cover the timed
# launches (+ set-up and warm-up) only.
set -o pipefail
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --isolated-frames 0 --present-frames 0"
timeout -k 10 170 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH > "$OUT/bench_stats.log" 2> "$OUT/rocprof_stats.log" || echo "stats failed"
timeout -k 10 170 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_serial" -- $BENCH --frames-in-flight 1 > "$OUT/bench_stats_serial.log" 2> "$OUT/rocprof_stats_serial.log" || echo "stats_serial failed"
for d in stats stats_serial; do
  f=$(find "$OUT/$d" -name "*kernel_stats.csv" | head -1)
  echo "== $d ($f)"; [ -n "$f" ] && cat "$f"
done > "$OUT/kernel_stats.txt"
cat "$OUT/kernel_stats.txt"
