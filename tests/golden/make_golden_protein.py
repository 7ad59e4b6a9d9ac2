#!/usr/bin/env python3
"""Regenerate tests/golden/prot/* (protein indexes + translated-search fixtures) from the REAL reference
(oracle/_ref, built by oracle/Makefile from /root/reference).  Run in the dev container only:

    make -C oracle ref && python tests/golden/make_golden_protein.py

Committed: data only - the .cfr files centrifuger-build --protein wrote for a synthetic proteome, DNA read files, the
reference binary's TSVs and the rank/access/locate streams of oracle/ref_dump.cpp's protein mode.  No reference source.
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
OUT = os.path.join(HERE, "prot")
REF = os.path.join(ROOT, "oracle", "_ref")
SEED = 20260929
AA = "ARNDCEQGHILKMFPSTWYV"

def dna_to_aa(a, b, c):
    """the standard code as Classifier::DnaToAa spells it (only used to build the reverse table)"""
    t = {"AA": "KNKN", "AC": "TTTT", "AG": "RSRS", "AT": "IIMI", "CA": "QHQH", "CC": "PPPP", "CG": "RRRR", "CT": "LLLL",
         "GA": "EDED", "GC": "AAAA", "GG": "GGGG", "GT": "VVVV", "TA": "_Y_Y", "TC": "SSSS", "TG": "_CWC", "TT": "LFLF"}
    return t[a + b]["ACGT".index(c)]


REV = {}
for a in "ACGT":
    for b in "ACGT":
        for c in "ACGT":
            REV.setdefault(dna_to_aa(a, b, c), []).append(a + b + c)


def run(cmd, stdout=None):
    subprocess.run(cmd, check=True, stdout=stdout, stderr=subprocess.DEVNULL)


def revcomp(s):
    return s[::-1].translate(str.maketrans("ACGTacgt", "TGCAtgca"))


def main():
    rng = np.random.default_rng(SEED)
    tmp = tempfile.mkdtemp(prefix="cfr_golden_prot_")
    os.makedirs(OUT, exist_ok=True)
    # ---- proteome: 4 species x 3 strains, 12 proteins each; strain k = base with 3k % substitutions; one protein shared by
    # all species (LCA at the root), one shared inside a genus, one duplicated inside a proteome
    prots, nodes, names = [], [(1, 1, "no rank"), (10, 1, "genus"), (11, 1, "genus")], [(1, "root"), (10, "GenusA"), (11, "GenusB")]
    shared_all = "".join(rng.choice(list(AA), size=220))
    shared_genus = ["".join(rng.choice(list(AA), size=180)) for _ in range(2)]
    tid = 100
    for sp in range(4):
        sp_tid = tid; tid += 1
        nodes.append((sp_tid, 10 + sp // 2, "species")); names.append((sp_tid, f"species {sp}"))
        base = ["".join(rng.choice(list(AA), size=int(rng.integers(150, 400)))) for _ in range(9)]
        base += [shared_all, shared_genus[sp // 2], base[0]]
        for k in range(3):
            st_tid = tid; tid += 1
            nodes.append((st_tid, sp_tid, "strain")); names.append((st_tid, f"species {sp} strain {k}"))
            for pi, p in enumerate(base):
                q = list(p)
                for pos in rng.integers(0, len(q), size=int(len(q) * 0.03 * k)):
                    q[pos] = AA[int(rng.integers(20))]
                if pi == 4:
                    q[50:80] = list("Q" * 30)          # a low-complexity stretch: runs in the BWT
                prots.append((f"P{sp}_{k}_{pi}", st_tid, "".join(q)))
    with open(os.path.join(tmp, "prot.fa"), "w") as f, open(os.path.join(tmp, "seqid.map"), "w") as m:
        for name, t, p in prots:
            f.write(f">{name}\n")
            for i in range(0, len(p), 60):
                f.write(p[i:i + 60] + "\n")
            m.write(f"{name}\t{t}\n")
    with open(os.path.join(tmp, "nodes.dmp"), "w") as f:
        for t, par, rank in nodes:
            f.write(f"{t}\t|\t{par}\t|\t{rank}\t|\n")
    with open(os.path.join(tmp, "names.dmp"), "w") as f:
        for t, nm in names:
            f.write(f"{t}\t|\t{nm}\t|\t\t|\tscientific name\t|\n")
    # the builder's inputs are fixtures too: the native protein writer (cfr_build_index, protein = 1) is checked against the
    # .cfr files below on exactly these files (tests/test_gpu_build_protein.py)
    os.makedirs(os.path.join(OUT, "input"), exist_ok=True)
    for fn in ("prot.fa", "seqid.map", "nodes.dmp", "names.dmp"):
        shutil.copy(os.path.join(tmp, fn), os.path.join(OUT, "input", fn))
    if "--inputs-only" in sys.argv:
        return
    build = [os.path.join(REF, "centrifuger-build"), "--protein", "-t", "4", "-r", os.path.join(tmp, "prot.fa"), "--taxonomy-tree", os.path.join(tmp, "nodes.dmp"),
             "--name-table", os.path.join(tmp, "names.dmp"), "--conversion-table", os.path.join(tmp, "seqid.map")]
    variants = {"p2": ["--ftabchars", "2"], "p3_b4": ["--ftabchars", "3", "--rbbwt-b", "4"], "p2_b1_off2": ["--ftabchars", "2", "--rbbwt-b", "1", "--offrate", "2"],
                "p4": []}
    manifest = {"seed": SEED, "indexes": {}, "cases": {}, "vectors": {}}
    for name, extra in variants.items():
        prefix = os.path.join(tmp, name)
        run(build + extra + ["-o", prefix])
        for k in (1, 2, 4):
            src = f"{prefix}.{k}.cfr"
            if k == 1 and os.path.getsize(src) > (1 << 20):
                with open(src, "rb") as fi, gzip.GzipFile(os.path.join(OUT, f"{name}.1.cfr.gz"), "wb", mtime=0) as fo:
                    fo.write(fi.read())
            else:
                shutil.copy(src, os.path.join(OUT, f"{name}.{k}.cfr"))
        manifest["indexes"][name] = {"md5_1cfr": hashlib.md5(open(f"{prefix}.1.cfr", "rb").read()).hexdigest(), "build_args": extra}

    # ---- DNA reads
    def reverse_translate(p):
        return "".join(REV[a][int(rng.integers(len(REV[a])))] for a in p)

    def read_from(pi, start, naa, frame=0, rc=False, sub=0.01):
        seg = prots[pi][2][start:start + naa]
        d = list("ACG"[:frame] + reverse_translate(seg) + "TC"[:int(rng.integers(3))])
        for pos in np.nonzero(rng.random(len(d)) < sub)[0]:
            d[pos] = "ACGT"[int(rng.integers(4))]
        d = "".join(d)
        return revcomp(d) if rc else d
    se = []
    for i in range(300):
        pi = int(rng.integers(len(prots)))
        L = len(prots[pi][2])
        naa = int(rng.integers(30, 84))
        st = int(rng.integers(0, max(1, L - naa)))
        se.append((f"r{i}", read_from(pi, st, naa, int(rng.integers(3)), bool(rng.random() < 0.5))))
    edge = []
    p0 = prots[0][2]
    edge.append(("exact", read_from(0, 10, 60, 0, False, 0)))
    edge.append(("exact_rc_frame2", read_from(0, 10, 60, 2, True, 0)))
    edge.append(("short32", read_from(0, 20, 10, 0, False, 0)[:32]))
    edge.append(("short33", read_from(0, 20, 11, 0, False, 0)[:33]))
    edge.append(("tiny", "AC"))
    edge.append(("allN", "N" * 90))
    e = list(read_from(1, 30, 70, 1, False, 0)); e[40] = "N"; e[41] = "N"; e[120] = "R"
    edge.append(("withN", "".join(e)))
    edge.append(("lower", read_from(2, 30, 50, 0, False, 0).lower()))
    e = list(read_from(3, 30, 60, 0, False, 0)); e[60:66] = list("acgtnn")
    edge.append(("mixedcase", "".join(e)))
    e = read_from(5, 20, 30, 0, False, 0); e2 = read_from(40, 50, 30, 0, False, 0)
    edge.append(("stop_between", e[:90] + "TAA" + e2[:90]))
    edge.append(("chimera_two_strands", read_from(6, 10, 35, 0, False, 0) + read_from(50, 10, 35, 0, True, 0)))
    edge.append(("shared_all", read_from(9, 40, 70, 0, False, 0)))         # the protein every species has
    edge.append(("shared_genus", read_from(10, 40, 60, 1, True, 0)))
    edge.append(("duplicate_in_proteome", read_from(11, 30, 60, 0, False, 0)))
    edge.append(("polyQ", read_from(4, 45, 40, 0, False, 0)))
    edge.append(("protein_end", read_from(7, len(prots[7][2]) - 40, 40, 0, False, 0)))
    edge.append(("across_two_proteins", read_from(7, len(prots[7][2]) - 25, 25, 0, False, 0)[:75] + read_from(8, 0, 25, 0, False, 0)[:75]))
    edge.append(("random", "".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=150))))
    edge.append(("long", read_from(20, 0, 140, 0, False, 0.005) + read_from(21, 5, 150, 0, False, 0.005)))
    pairs1, pairs2 = [], []
    for i in range(150):
        pi = int(rng.integers(len(prots)))
        L = len(prots[pi][2])
        st = int(rng.integers(0, max(1, L - 140)))
        pairs1.append((f"p{i}/1", read_from(pi, st, 45, int(rng.integers(3)), False)))
        pairs2.append((f"p{i}/2", read_from(pi, min(L - 45, st + 80), 45, int(rng.integers(3)), True)))

    def write_fa(recs, path):
        with open(path, "w") as f:
            for rid, s in recs:
                f.write(f">{rid}\n{s}\n")
    write_fa(se, os.path.join(OUT, "se.fa")); write_fa(edge, os.path.join(OUT, "edge.fa"))
    write_fa(pairs1, os.path.join(OUT, "pe_1.fa")); write_fa(pairs2, os.path.join(OUT, "pe_2.fa"))
    write_fa(edge[::-1], os.path.join(OUT, "edge_2.fa"))

    cf = os.path.join(REF, "centrifuger")
    cases = {"se_default": ["-u", "se.fa"], "se_k5": ["-u", "se.fa", "-k", "5"], "se_minhit8": ["-u", "se.fa", "--min-hitlen", "8", "-k", "3"],
             "edge_default": ["-u", "edge.fa"], "edge_k4": ["-u", "edge.fa", "-k", "4"], "pe_k3": ["-1", "pe_1.fa", "-2", "pe_2.fa", "-k", "3"],
             "edge_pe": ["-1", "edge.fa", "-2", "edge_2.fa"], "se_hitk1": ["-u", "se.fa", "--hitk-factor", "1", "-k", "2"]}
    os.makedirs(os.path.join(OUT, "tsv"), exist_ok=True)
    for iname in variants:
        for cname, args in cases.items():
            if iname not in ("p2", "p3_b4") and cname not in ("se_default", "edge_k4", "pe_k3"):
                continue
            out = os.path.join(OUT, "tsv", f"{iname}.{cname}.tsv")
            a = [x if not x.endswith(".fa") else os.path.join(OUT, x) for x in args]
            with open(out, "wb") as fo:
                run([cf, "-x", os.path.join(tmp, iname), "-t", "1"] + a, stdout=fo)
            manifest["cases"][f"{iname}.{cname}"] = {"index": iname, "args": args, "md5": hashlib.md5(open(out, "rb").read()).hexdigest()}
    rd = os.path.join(REF, "ref_dump")
    os.makedirs(os.path.join(OUT, "vec"), exist_ok=True)
    for iname in variants:
        idx1 = os.path.join(tmp, iname + ".1.cfr")
        for kind, arg in (("prank", "1" if iname != "p4" else "7"), ("plocate", "1")):
            raw = subprocess.run([rd, kind, idx1, arg], check=True, stdout=subprocess.PIPE).stdout
            entry = {"md5": hashlib.md5(raw).hexdigest(), "arg": arg, "lines": raw.count(b"\n")}
            if iname == "p3_b4" and kind == "prank":
                with gzip.GzipFile(os.path.join(OUT, "vec", f"{iname}.{kind}.txt.gz"), "wb", mtime=0) as fo:
                    fo.write(raw)
                entry["file"] = f"vec/{iname}.{kind}.txt.gz"
            manifest["vectors"][f"{iname}.{kind}"] = entry
    json.dump(manifest, open(os.path.join(OUT, "manifest.json"), "w"), indent=1, sort_keys=True)
    shutil.rmtree(tmp)
    print("protein golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
