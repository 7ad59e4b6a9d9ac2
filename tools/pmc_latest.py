#!/usr/bin/env python3
"""Turn the PMC passes of tools/pmc_passes.sh into profiles/pmc_latest.json (what bench.py reports as roofline.traffic).
usage: pmc_latest.py <pmc dir> <round tag> <out json>"""
import csv
import glob
import json
import os
import sys


def main(d, tag, out):
    vals = {}
    for f in sorted(glob.glob(os.path.join(d, "pmc*", "*", "*counter_collection.csv"))):
        for row in csv.DictReader(open(f)):
            if "k_search_chains_v2" in row["Kernel_Name"]:
                vals[row["Counter_Name"]] = float(row["Counter_Value"])      # last dispatch wins = the timed step
    reads = json.load(open(os.path.join(d, "bench_plain.json")))["config"]["reads_per_step_per_gpu"]
    res = {"round": tag, "source": f"profiles/{tag}_pmc_summary.txt (rocprofv3 --pmc, separate passes, {reads}-read launch, 1 Gbp index)",
           "reads_in_profiled_launch": reads,
           "k_search_chains_v2": {
               "FETCH_SIZE_KiB": vals.get("FETCH_SIZE"), "TCC_EA0_RDREQ": vals.get("TCC_EA0_RDREQ_sum"),
               "TCC_EA0_RDREQ_32B": vals.get("TCC_EA0_RDREQ_32B_sum"), "WRITE_SIZE_KiB": vals.get("WRITE_SIZE"),
               "fabric_read_bytes_per_read": vals["FETCH_SIZE"] * 1024 / reads, "write_bytes_per_read": vals["WRITE_SIZE"] * 1024 / reads,
               "note": "FETCH_SIZE == TCC_EA0_RDREQ x 64 B (all requests are 64 B; random gathers, so the streaming half-count "
                       "correction of the guide does not apply)"}}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["k_search_chains_v2"]))


if __name__ == "__main__":
    main(*sys.argv[1:4])
