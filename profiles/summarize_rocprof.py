#!/usr/bin/env python3
"""Reduce rocprofv3 CSV output (kernel stats + PMC passes) to a short per-kernel summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


for sub, what in (("stats", "the default bench command, frames in flight as bench.py defaults"),
                  ("stats_serial", "same command with --frames-in-flight 1: every kernel has the GPU to itself")):
    files = find(f"{sub}/**/*kernel_stats.csv")
    if not files:
        continue
    print(f"== kernel stats (rocprofv3 --kernel-trace --stats), {what} ==")
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Name", "")[:70]
                print(f"{name:70s} calls={row.get('Calls'):>5s} total_ns={row.get('TotalDurationNs'):>12s} avg_ns={row.get('AverageNs'):>12s} pct={row.get('Percentage')}")

for tag in ("pmc_fetch", "pmc_write", "pmc_l2", "pmc_sq"):
    files = find(f"{tag}/**/*counter_collection.csv")
    if not files:
        continue
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "")[:60]
                c = row.get("Counter_Name", "")
                agg[k][c] += float(row.get("Counter_Value", 0) or 0)
                cnt[k][c] += 1
    print(f"== {tag}: per-kernel counter mean per dispatch ==")
    for k in agg:
        for c in agg[k]:
            print(f"{k:60s} {c:22s} mean={agg[k][c] / max(cnt[k][c], 1):16.1f} dispatches={cnt[k][c]}")
