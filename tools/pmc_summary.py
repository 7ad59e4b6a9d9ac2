#!/usr/bin/env python3
"""Fold rocprofv3 --pmc counter_collection.csv files (one per pass) into a per-kernel table.
usage: pmc_summary.py <dir with pmc*/...counter_collection.csv> [out.txt]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"cfr::(k_[a-z_]+)", name)
    return m.group(1) if m else name[:40]


def main(d, out=None):
    agg = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> [values per dispatch]
    dur = defaultdict(list)
    for f in sorted(glob.glob(os.path.join(d, "pmc*", "*", "*counter_collection.csv"))):
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            dur[(k, row["Counter_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    lines = ["# per-dispatch PMC values (LAST dispatch of each kernel in the profiled run = the timed step)", ""]
    for k in sorted(agg):
        lines.append(f"[{k}]")
        for cname in sorted(agg[k]):
            v = agg[k][cname]
            t = dur[(k, cname)]
            lines.append(f"  {cname:<32} last={v[-1]:>18.1f}  dispatches={len(v)}  kernel_ns(last)={t[-1]}")
        c = {n: agg[k][n][-1] for n in agg[k]}
        if "FETCH_SIZE" in c:
            lines.append(f"  -> FETCH_SIZE is in KiB: {c['FETCH_SIZE']*1024/1e6:.1f} MB read from the fabric (x2 if the gfx950 half-count applies)")
        if "WRITE_SIZE" in c:
            lines.append(f"  -> WRITE_SIZE: {c['WRITE_SIZE']*1024/1e6:.1f} MB")
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
            lines.append(f"  -> L2 hit rate: {c['TCC_HIT_sum']/(c['TCC_HIT_sum']+c['TCC_MISS_sum']):.3f}")
        lines.append("")
    text = "\n".join(lines)
    if out:
        open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
