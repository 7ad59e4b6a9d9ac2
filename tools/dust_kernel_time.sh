#!/bin/bash
# k_dust's own time (rocprofv3 kernel trace) on 10 M x 150 bp reads: random, and with 2 % low-complexity reads mixed in.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for lc in 0 0.02; do
  rm -rf /tmp/dustp_$lc
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dustp_$lc -o d -- python $ROOT/tools/dust_device_timing.py --lowc $lc "$@" 2>&1 | grep -E "rep 2|equals"
  python - <<PY
import csv, glob
for f in glob.glob("/tmp/dustp_$lc/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dust" in r["Name"]:
            print("lowc $lc:", r["Name"][:48], "calls", r["Calls"], "avg ms %.2f" % (float(r["AverageNs"]) / 1e6))
for f in glob.glob("/tmp/dustp_$lc/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "dust" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    calls = [rows[i:i + 3] for i in range(0, len(rows), 3)] if len(rows) % 3 == 0 else []
    for c in calls[-1:]:
        print("lowc $lc: span of the last call's dust kernels %.2f ms" % ((max(int(r["End_Timestamp"]) for r in c) - min(int(r["Start_Timestamp"]) for r in c)) / 1e6))
PY
done
