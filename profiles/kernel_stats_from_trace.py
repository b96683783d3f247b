#!/usr/bin/env python3
"""profiles/kernel_stats_from_trace.py <gpurun_out/prof_TAG> "<the command that was profiled>" > profiles/rNN/rocprof_kernel_stats_TAG.txt
Per-kernel durations of the `stats` pass of profiles/pmc_rNN.sh (rocprofv3 --kernel-trace --stats): a run launches every kernel over ONE
frame too (each frame object's first render, the counting passes), so rocprofv3's own average mixes two kinds of launch; this lists the
launches of the timed kind (duration within 20 % of the longest) next to the rest, and keeps rocprofv3's table below."""
import csv
import glob
import os
import sys
from collections import defaultdict

out, title = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
print(f"# {title}")
tr = glob.glob(os.path.join(out, "stats", "**", "*kernel_trace.csv"), recursive=True)
dur = defaultdict(list)
for row in csv.DictReader(open(tr[0])):
    dur[row["Kernel_Name"]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6)
print(f"{'kernel':76s} {'launches of the timed kind':>28s} {'avg ms':>9s} {'min':>8s} {'max':>8s}   | other launches (one frame: set-up, counting passes)")
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    if "rtrdev" not in k:
        continue
    v = dur[k]; m = max(v); big = [x for x in v if x >= 0.8 * m]; small = [x for x in v if x < 0.8 * m]
    short = k.split("(")[0].replace("void ", "")[:76]
    print(f"{short:76s} {len(big):28d} {sum(big) / len(big):9.4f} {min(big):8.4f} {max(big):8.4f}   | {len(small)} launches" + (f", avg {sum(small) / len(small):.4f} ms" if small else ""))
f = glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True)
if f:
    print(f"-- rocprofv3's own table ({os.path.basename(f[0])}): averages over ALL launches of a kernel")
    sys.stdout.write(open(f[0]).read())
