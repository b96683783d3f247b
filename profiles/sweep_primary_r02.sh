#!/bin/bash
# Knobs of k_primary_persist (camera rays, persistent waves over the BVH2): bash profiles/sweep_primary_r02.sh  -> gpurun_out/r02/sweep_primary.log
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r02; mkdir -p "$OUT"
LOG=$OUT/sweep_primary.log; : > "$LOG"
run() { echo "== $*" >> "$LOG"; env "$@" timeout -k 10 120 python3 $REPO/profiles/print_stats.py sponza_class 1920 1080 1 2>/dev/null | grep -E "timed form|per shadow" >> "$LOG" || echo failed >> "$LOG"; }
run RTR_PRIMARY_PERSIST=0
run RTR_PRIMARY_PERSIST=1
for r in 8 16 24 32 48 64; do run RTR_PRIMARY_REFILL=$r; done
for m in 0 12 20 28 40; do run RTR_PRIMARY_INNER_MIN=$m; done
for b in 64 128 512 1024; do run RTR_PRIMARY_BATCH=$b; done
for w in 4 6; do run RTR_PRIMARY_WGS_PER_CU=$w; done
cat "$LOG"
