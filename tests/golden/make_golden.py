#!/usr/bin/env python3
"""Regenerate tests/golden/* from the REAL reference (oracle/_ref, built by oracle/Makefile
from /root/reference).  Run in the dev container only:

    make -C oracle ref && python tests/golden/make_golden.py

What is committed is data only: synthetic inputs (genomes are NOT kept, only the .cfr
indexes the reference's centrifuger-build wrote), read files, and the reference's outputs
(TSV + intermediate-vector dumps from oracle/ref_dump.cpp).  No reference source.
"""
import gzip
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
SEED = 20260928


def run(cmd, stdout=None):
    subprocess.run(cmd, check=True, stdout=stdout, stderr=subprocess.DEVNULL)


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        h.update(f.read())
    return h.hexdigest()


def edge_reads(g, rng):
    """Reads exercising Appendix-B corner cases of SURVEY.md."""
    cat = np.concatenate(g.seqs)
    recs = []

    def take(p, L):
        return cat[p:p + L].tobytes()

    p0 = 5000
    recs.append(("exact150", take(p0, 150)))
    recs.append(("short22", take(p0, 22)))
    recs.append(("short23", take(p0 + 100, 23)))
    recs.append(("short9", take(p0 + 200, 9)))
    recs.append(("one", b"A"))
    recs.append(("allN", b"N" * 80))
    recs.append(("lower", take(p0 + 300, 150).lower()))
    mixed = bytearray(take(p0 + 500, 150)); mixed[70:75] = b"acgtn"
    recs.append(("mixedcase", bytes(mixed)))
    withn = bytearray(take(p0 + 700, 150)); withn[30] = ord("N"); withn[31] = ord("N"); withn[100] = ord("R")
    recs.append(("withN", bytes(withn)))
    nend = bytearray(take(p0 + 900, 150)); nend[149] = ord("N"); nend[0] = ord("N")
    recs.append(("Nends", bytes(nend)))
    recs.append(("polyA", b"A" * 150))
    recs.append(("dinuc", b"AC" * 75))
    lowc = bytearray(take(p0 + 1100, 150)); lowc[40:100] = b"ATT" * 20
    recs.append(("lowcomplex_mid", bytes(lowc)))
    # chimera: two strains / two strands -> AdjustHitBoundary traffic
    a = take(20000, 90); b = synth.revcomp(np.frombuffer(take(len(g.seqs[0]) + 40000, 80), dtype=np.uint8)).tobytes()
    recs.append(("chimera_fr", a + b"N" + b[:59]))
    recs.append(("chimera_ff", take(30000, 70) + take(len(g.seqs[0]) * 3 + 1000, 80)))
    # random (unclassifiable) read
    recs.append(("random", synth.ACGT[rng.integers(0, 4, size=150)].tobytes()))
    # text start / text end (cyclic wrap of the $-less BWT, Appendix B item 4)
    recs.append(("textstart", take(0, 120)))
    recs.append(("textend", cat[-120:].tobytes()))
    recs.append(("wrap", cat[-60:].tobytes() + cat[:60].tobytes()))
    # genome boundary
    L0 = len(g.seqs[0])
    recs.append(("boundary", take(L0 - 75, 150)))
    return recs


def main():
    rng = np.random.default_rng(SEED)
    tmp = tempfile.mkdtemp(prefix="cfr_golden_")
    g = synth.make_genomes(n_species=5, n_strains=3, genome_len=20000, seed=SEED)
    synth.write_reference_inputs(g, tmp)
    build = [os.path.join(REF, "centrifuger-build"), "-t", "4", "-r", os.path.join(tmp, "ref.fa"),
             "--taxonomy-tree", os.path.join(tmp, "nodes.dmp"), "--name-table", os.path.join(tmp, "names.dmp"),
             "--conversion-table", os.path.join(tmp, "seqid.map")]
    variants = {
        "f6": ["--ftabchars", "6"],
        "f6_b1": ["--ftabchars", "6", "--rbbwt-b", "1"],
        "f6_b8": ["--ftabchars", "6", "--rbbwt-b", "8"],
        "f6_off3": ["--ftabchars", "6", "--offrate", "3"],
        "f10": [],
    }
    manifest = {"seed": SEED, "indexes": {}, "cases": {}}
    for name, extra in variants.items():
        prefix = os.path.join(tmp, name)
        run(build + extra + ["-o", prefix])
        for k in (1, 2, 4):
            src = f"{prefix}.{k}.cfr"
            if name == "f10" and k == 1:   # 16 MiB ftab: keep gz only
                with open(src, "rb") as fi, gzip.GzipFile(os.path.join(HERE, f"{name}.1.cfr.gz"), "wb", mtime=0) as fo:
                    fo.write(fi.read())
            else:
                shutil.copy(src, os.path.join(HERE, f"{name}.{k}.cfr"))
        manifest["indexes"][name] = {"md5_1cfr": md5(f"{prefix}.1.cfr"), "build_args": extra}

    # ---- read sets
    se = synth.make_reads(g, 400, 150, seed=SEED + 1, sub_rate=0.02, n_rate=0.004)
    synth.write_fastq(se, os.path.join(HERE, "se.fq"))
    p1, p2 = synth.make_pairs(g, 200, 150, seed=SEED + 2, ins_lo=200, ins_hi=400, sub_rate=0.02, n_rate=0.004)
    synth.write_fastq(p1, os.path.join(HERE, "pe_1.fq"), suffix="/1")
    synth.write_fastq(p2, os.path.join(HERE, "pe_2.fq"), suffix="/2")
    lr = synth.make_long_reads(g, 12, 2000, 6000, seed=SEED + 3)
    synth.write_fastq(lr, os.path.join(HERE, "long.fq"))
    with open(os.path.join(HERE, "edge.fa"), "wb") as f:
        for rid, s in edge_reads(g, rng):
            f.write(b">" + rid.encode() + b"\n" + s + b"\n")
    # mates for the edge set: reverse order of the same reads (ragged pair lengths)
    er = edge_reads(g, np.random.default_rng(SEED))
    with open(os.path.join(HERE, "edge_2.fa"), "wb") as f:
        for (rid, _), (_, s2) in zip(er, er[::-1]):
            f.write(b">" + rid.encode() + b"\n" + s2 + b"\n")

    cf = os.path.join(REF, "centrifuger")
    cases = {
        "se_default": ["-u", "se.fq"],
        "se_nodust": ["-u", "se.fq", "--no-dust"],
        "se_k5": ["-u", "se.fq", "-k", "5"],
        "se_hitk2": ["-u", "se.fq", "--hitk-factor", "2", "-k", "2"],
        "se_hitk0": ["-u", "se.fq", "--hitk-factor", "0"],
        "se_minhit16": ["-u", "se.fq", "--min-hitlen", "16"],
        "pe_k5": ["-1", "pe_1.fq", "-2", "pe_2.fq", "-k", "5"],
        "pe_default": ["-1", "pe_1.fq", "-2", "pe_2.fq"],
        "long_default": ["-u", "long.fq"],
        "edge_default": ["-u", "edge.fa"],
        "edge_nodust": ["-u", "edge.fa", "--no-dust"],
        "edge_pe_k3": ["-1", "edge.fa", "-2", "edge_2.fa", "-k", "3"],
    }
    os.makedirs(os.path.join(HERE, "tsv"), exist_ok=True)
    for iname in variants:
        for cname, args in cases.items():
            if iname not in ("f6", "f10") and cname not in ("se_default", "pe_k5", "edge_default"):
                continue
            out = os.path.join(HERE, "tsv", f"{iname}.{cname}.tsv")
            a = [x if not x.endswith((".fq", ".fa")) else os.path.join(HERE, x) for x in args]
            with open(out, "wb") as fo:
                run([cf, "-x", os.path.join(tmp, iname), "-t", "1"] + a, stdout=fo)
            manifest["cases"][f"{iname}.{cname}"] = {"index": iname, "args": args, "md5": md5(out)}

    # ---- intermediate vectors (reference headers driven by oracle/ref_dump.cpp)
    rd = os.path.join(REF, "ref_dump")
    os.makedirs(os.path.join(HERE, "vec"), exist_ok=True)
    manifest["vectors"] = {}
    for iname in variants:
        idx1 = os.path.join(tmp, iname + ".1.cfr")
        for kind, arg in (("rank", "1" if iname != "f10" else "37"), ("locate", "1" if iname != "f10" else "5"),
                          ("bs", os.path.join(HERE, "edge.fa")), ("bs_se", os.path.join(HERE, "se.fq"))):
            k = "bs" if kind.startswith("bs") else kind
            raw = subprocess.run([rd, k, idx1, arg], check=True, stdout=subprocess.PIPE).stdout
            entry = {"md5": hashlib.md5(raw).hexdigest(), "arg": os.path.basename(arg), "lines": raw.count(b"\n")}
            if iname == "f6" or kind == "bs":     # keep full streams only where they are small
                if len(raw) < (1 << 21):
                    with gzip.GzipFile(os.path.join(HERE, "vec", f"{iname}.{kind}.txt.gz"), "wb", mtime=0) as fo:
                        fo.write(raw)
                    entry["file"] = f"vec/{iname}.{kind}.txt.gz"
            manifest["vectors"][f"{iname}.{kind}"] = entry

    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    shutil.rmtree(tmp)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
