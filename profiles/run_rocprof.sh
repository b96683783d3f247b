#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   bash profiles/run_rocprof.sh <tag>        e.g. r01
# 1) --kernel-trace --stats of the default bench command  -> gpurun_out/prof_<tag>/stats
# 2) separate --pmc passes (FETCH_SIZE / WRITE_SIZE cannot share a pass on gfx950: TCC has 4 slots)
# Copy the *_kernel_stats.csv / pmc summaries you want judged into profiles/.
set -o pipefail
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# the default bench command (3 frames in flight); --isolated-frames 0 --present-frames 0 drops the untimed one-at-a-time pass so the
# averages below cover the timed launches only.  stats_serial = the same frames one at a time (kernel cost in isolation).
BENCH="python3 $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --isolated-frames 0 --present-frames 0"
timeout -k 10 170 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH > "$OUT/bench_stats.log" 2>&1
timeout -k 10 170 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_serial" -- $BENCH --frames-in-flight 1 > "$OUT/bench_stats_serial.log" 2>&1
timeout -k 10 170 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/bench_pmc_fetch.log" 2>&1
timeout -k 10 170 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/bench_pmc_write.log" 2>&1
timeout -k 10 170 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_l2" -- $BENCH > "$OUT/bench_pmc_l2.log" 2>&1
timeout -k 10 170 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/pmc_sq" -- $BENCH > "$OUT/bench_pmc_sq.log" 2>&1
echo passes done; python3 $REPO/profiles/summarize_rocprof.py "$OUT" > "$OUT/summary.txt" 2>&1 || true
tail -40 "$OUT/summary.txt"
