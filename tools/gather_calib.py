#!/usr/bin/env python3
"""Summary of tools/gather_calib.sh: fabric read requests and FETCH_SIZE per touched record, per mode.
usage: gather_calib.py <dir> <lanes> <steps>"""
import csv
import glob
import json
import os
import sys

MODES = {1: "one 16-B load in one random 64-B record", 4: "two 16-B loads, both halves of one random 128-B line",
         5: "two 16-B loads in two independent random 64-B records", 6: "one 4-B load (random dword)",
         2: "8-B + 16-B load in one random 64-B record"}


def main(d, lanes, steps):
    n = int(lanes) * int(steps)
    res = {"steps_per_launch": n}
    for mode, what in MODES.items():
        c = {}
        for f in glob.glob(os.path.join(d, f"pmc*_mode{mode}", "*", "*counter_collection.csv")):
            for row in csv.DictReader(open(f)):
                if "chase" in row["Kernel_Name"]:
                    c[row["Counter_Name"]] = float(row["Counter_Value"])        # last dispatch (3 reps) wins
        if not c:
            continue
        rd = c.get("TCC_EA0_RDREQ_sum")
        entry = {"pattern": what, "counters": c}
        if rd:
            entry["requests_per_step"] = rd / n
            entry["FETCH_SIZE_bytes_per_step"] = c.get("FETCH_SIZE", 0) * 1024 / n
            entry["FETCH_SIZE_bytes_per_request"] = c.get("FETCH_SIZE", 0) * 1024 / rd
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
            entry["l2_hit"] = c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        t = open(os.path.join(d, f"mode{mode}.txt")).read().strip() if os.path.exists(os.path.join(d, f"mode{mode}.txt")) else ""
        entry["unprofiled_run"] = t
        res[f"mode{mode}"] = entry
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
