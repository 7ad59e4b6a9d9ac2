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
PY
done
