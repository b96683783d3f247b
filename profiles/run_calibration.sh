#!/bin/bash
set -e -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_calib
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 $REPO/profiles/calibrate_fetch.py > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 $REPO/profiles/calibrate_fetch.py > "$OUT/write.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
for tag in ("fetch", "write"):
    for f in glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            v = float(row["Counter_Value"] or 0)
            if v > 100000:
                print(f"{tag:5s} {row['Kernel_Name'][:90]:90s} {row['Counter_Name']:10s} {v:14.0f} KB = {v*1024/2**20:9.1f} MiB")
PY
