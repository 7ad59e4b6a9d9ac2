#!/usr/bin/env python3
"""Turn the PMC passes of tools/pmc_passes.sh into profiles/pmc_latest.json (what bench.py reports as roofline.traffic).
usage: pmc_latest.py <pmc dir> <round tag> <out json>"""
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_sha():      # the same digest bench.py computes: tells a reader whether the profile is of the benched kernels
    h = hashlib.sha1()
    for f in ("cfr_kernels.hip.inc", "cfr_device.hip", "cfr_device.hpp"):
        h.update(open(os.path.join(ROOT, "centrifuger_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:12]


def main(d, tag, out):
    vals = {}
    for f in sorted(glob.glob(os.path.join(d, "pmc*", "*", "*counter_collection.csv"))):
        for row in csv.DictReader(open(f)):
            if "k_search_chains_v2" in row["Kernel_Name"]:
                vals[row["Counter_Name"]] = float(row["Counter_Value"])      # last dispatch wins = the timed step
    reads = json.load(open(os.path.join(d, "bench_plain.json")))["config"]["reads_per_step_per_gpu"]
    res = {"round": tag, "source": f"profiles/{tag}_pmc_summary.txt (rocprofv3 --pmc, separate passes, {reads}-read launch, 1 Gbp index)",
           "reads_in_profiled_launch": reads, "kernel_source_sha": kernel_source_sha(),
           "k_search_chains_v2": {
               "FETCH_SIZE_KiB": vals.get("FETCH_SIZE"), "TCC_EA0_RDREQ": vals.get("TCC_EA0_RDREQ_sum"),
               "TCC_EA0_RDREQ_32B": vals.get("TCC_EA0_RDREQ_32B_sum"), "WRITE_SIZE_KiB": vals.get("WRITE_SIZE"),
               "fabric_read_bytes_per_read": vals["FETCH_SIZE"] * 1024 / reads, "write_bytes_per_read": vals["WRITE_SIZE"] * 1024 / reads,
               "note": "FETCH_SIZE == TCC_EA0_RDREQ x 64 B is the counter's tally; calibrated on this chip a random gather's request carries a "
                       "128-byte line (profiles/r2a_gather_calib.json), so fabric bytes = TCC_EA0_RDREQ x 128 (what bench.py uses)"}}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["k_search_chains_v2"]))


if __name__ == "__main__":
    main(*sys.argv[1:4])
