#!/usr/bin/env python3
"""Golden vectors of `--expand-taxid` (Classifier.hpp:792-838, Taxonomy.hpp:733-973, ResultWriter.hpp:194-195, 226-227), made by the
REAL reference (oracle/_ref).  Run in the dev container only:

    make -C oracle ref && python tests/golden/make_golden_expand.py

Writes tests/golden/expand/: a small index `x8` over a taxonomy built to walk every branch of the children bookkeeping
(no-rank nodes between ranked ones, missing levels, a sequence filed at an inner node, two sequences with one tax id, a
sequence filed at the root, a FASTA sequence the conversion table does not name), its read files, and the reference's TSVs
for the committed `f6` index and for `x8` with `--expand-taxid` at -k 1 / 2 / 3 / 5, single-end and paired.  Data only.
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from centrifuger_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
OUT = os.path.join(HERE, "expand")
SEED = 20260929

# (tax id, parent, rank)
NODES = [
    (1, 1, "no rank"), (2, 1, "superkingdom"),
    (10, 2, "phylum"), (20, 10, "class"), (30, 20, "order"), (40, 30, "family"),
    (100, 40, "genus"),
    (110, 100, "no rank"),            # a clade between genus and species
    (120, 110, "species"), (121, 120, "strain"), (122, 120, "strain"),
    (123, 120, "subspecies"), (124, 123, "strain"),
    (130, 100, "species"), (131, 130, "strain"),
    (140, 40, "species"),             # no genus above it
    (141, 140, "no rank"),            # an unranked leaf
    (200, 2, "genus"),                # no family .. phylum above it
    (210, 200, "species"), (211, 210, "strain"), (212, 210, "strain"),
    (300, 1, "species"),              # filed under the root
    (301, 300, "strain"),
]
# (sequence name, tax id or None = not in the conversion table, family of near-identical sequences, substitutions per base against the family's base)
SEQS = [
    ("XA_121a", 121, 0, 0.000), ("XA_121b", 121, 0, 0.002),     # two sequences with one tax id
    ("XA_122", 122, 0, 0.003), ("XA_124", 124, 0, 0.004),
    ("XA_130", 130, 0, 0.003),                                  # filed at the species itself: an input id that is an ancestor of others
    ("XA_131", 131, 0, 0.005), ("XA_141", 141, 0, 0.006),
    ("XA_211", 211, 0, 0.008), ("XA_extra", None, 0, 0.004),    # no tax id: SeqIdToTaxId gives the node count
    ("XB_212", 212, 1, 0.000), ("XB_301", 301, 1, 0.003), ("XB_root", 1, 1, 0.004), ("XB_122", 122, 1, 0.005),
    ("XC_211", 211, 2, 0.000), ("XC_212", 212, 2, 0.002), ("XC_210", 210, 2, 0.004),
    ("XD_124", 124, 3, 0.000), ("XD_123", 123, 3, 0.002), ("XD_121", 121, 3, 0.003), ("XD_140", 140, 3, 0.004),
]
GENOME_LEN = 9000


def run(cmd, stdout=None):
    subprocess.run(cmd, check=True, stdout=stdout, stderr=subprocess.DEVNULL)


def md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


def main():
    rng = np.random.default_rng(SEED)
    tmp = tempfile.mkdtemp(prefix="cfr_golden_expand_")
    os.makedirs(os.path.join(OUT, "tsv"), exist_ok=True)
    bases = [rng.integers(0, 4, size=GENOME_LEN, dtype=np.uint8) for _ in range(4)]
    names, taxids, seqs = [], [], []
    for name, tid, fam, div in SEQS:
        g = bases[fam].copy()
        nmut = int(GENOME_LEN * div)
        pos = rng.integers(0, GENOME_LEN, size=nmut)
        g[pos] = (g[pos] + rng.integers(1, 4, size=nmut, dtype=np.uint8)) & 3
        names.append(name); taxids.append(tid); seqs.append(synth.ACGT[g])
    g = synth.Genomes(names, taxids, seqs, NODES, [(t, f"taxon{t}") for t, _, _ in NODES])
    synth.write_reference_inputs(g, tmp)
    with open(os.path.join(tmp, "seqid.map"), "w") as f:          # the table leaves XA_extra out
        for name, tid in zip(names, taxids):
            if tid is not None:
                f.write(f"{name}\t{tid}\n")
    prefix = os.path.join(tmp, "x8")
    run([os.path.join(REF, "centrifuger-build"), "-t", "2", "-r", os.path.join(tmp, "ref.fa"), "--taxonomy-tree", os.path.join(tmp, "nodes.dmp"),
         "--name-table", os.path.join(tmp, "names.dmp"), "--conversion-table", os.path.join(tmp, "seqid.map"), "--ftabchars", "6", "-o", prefix])
    for k in (1, 2, 4):
        shutil.copy(f"{prefix}.{k}.cfr", os.path.join(OUT, f"x8.{k}.cfr"))
    for k in (1, 2, 4):                                            # the committed f6 index, side by side for the reference binary
        shutil.copy(os.path.join(HERE, f"f6.{k}.cfr"), os.path.join(tmp, f"f6.{k}.cfr"))

    se = synth.make_reads(g, 500, 150, seed=SEED + 1, sub_rate=0.004, n_rate=0.002)
    synth.write_fastq(se, os.path.join(OUT, "x8_se.fq"))
    p1, p2 = synth.make_pairs(g, 250, 150, seed=SEED + 2, ins_lo=200, ins_hi=400, sub_rate=0.004, n_rate=0.002)
    synth.write_fastq(p1, os.path.join(OUT, "x8_pe_1.fq"), suffix="/1")
    synth.write_fastq(p2, os.path.join(OUT, "x8_pe_2.fq"), suffix="/2")

    cf = os.path.join(REF, "centrifuger")
    cases = {}
    for k in (1, 2, 3, 5):
        cases[f"x8.se_k{k}_expand"] = ("x8", ["-u", "expand/x8_se.fq", "-k", str(k), "--expand-taxid"])
        cases[f"x8.pe_k{k}_expand"] = ("x8", ["-1", "expand/x8_pe_1.fq", "-2", "expand/x8_pe_2.fq", "-k", str(k), "--expand-taxid"])
    cases["x8.se_k3"] = ("x8", ["-u", "expand/x8_se.fq", "-k", "3"])
    cases["x8.se_k1_expand_nodust_hitk2"] = ("x8", ["-u", "expand/x8_se.fq", "--expand-taxid", "--no-dust", "--hitk-factor", "2"])
    for k in (1, 2):
        cases[f"f6.se_k{k}_expand"] = ("f6", ["-u", "se.fq", "-k", str(k), "--expand-taxid"])
        cases[f"f6.pe_k{k}_expand"] = ("f6", ["-1", "pe_1.fq", "-2", "pe_2.fq", "-k", str(k), "--expand-taxid"])
    cases["f6.edge_k1_expand"] = ("f6", ["-u", "edge.fa", "--expand-taxid"])
    cases["f6.edge_pe_k2_expand"] = ("f6", ["-1", "edge.fa", "-2", "edge_2.fa", "-k", "2", "--expand-taxid"])
    manifest = {"seed": SEED, "cases": {}}
    for cname, (iname, args) in cases.items():
        out = os.path.join(OUT, "tsv", cname + ".tsv")
        a = [os.path.join(HERE, x) if x.endswith((".fq", ".fa")) else x for x in args]
        with open(out, "wb") as fo:
            run([cf, "-x", os.path.join(tmp, iname), "-t", "1"] + a, stdout=fo)
        body = open(out, "rb").read().split(b"\n")[1:]
        nonempty = sum(1 for ln in body if ln.count(b"\t") == 8 and ln.split(b"\t")[8])
        manifest["cases"][cname] = {"index": iname, "args": args, "md5": md5(out), "rows_with_expanded_ids": nonempty}
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    shutil.rmtree(tmp)
    for c, v in sorted(manifest["cases"].items()):
        print(c, v["rows_with_expanded_ids"])


if __name__ == "__main__":
    main()
