"""The native PROTEIN index writer (cfr_build_index with protein = 1: suffix array of the byte text on the MI355X, csrc/cfr_build_sa.hip
build_sa_bytes; Sequence_RunBlockOneTree compression + endMarkerSA on the host, csrc/cfr_build.cpp) against the indexes the REAL
reference's `centrifuger-build --protein` wrote for the same proteome (tests/golden/prot/*.cfr, inputs under tests/golden/prot/input),
field by field, and the command line `bin/centrifuger-build --protein` against the same files.  -m gpu."""
import gzip
import os
import shutil
import subprocess

import numpy as np
import pytest

from centrifuger_amd import capi
from cfr_fields import parse_1cfr
from conftest import GOLDEN, ROOT, have_ref, REF_DIR

pytestmark = pytest.mark.gpu
PROT = os.path.join(GOLDEN, "prot")
INPUT = os.path.join(PROT, "input")
VARIANTS = {"p2": dict(ftab_chars=2), "p3_b4": dict(ftab_chars=3, rbbwt_b=4), "p2_b1_off2": dict(ftab_chars=2, rbbwt_b=1, offrate=2), "p4": dict(ftab_chars=4)}
CLI_ARGS = {"p2": ["--ftabchars", "2"], "p3_b4": ["--ftabchars", "3", "--rbbwt-b", "4"], "p2_b1_off2": ["--ftabchars", "2", "--rbbwt-b", "1", "--offrate", "2"], "p4": []}


def read_inputs(directory=INPUT):
    names, taxids, seqs = [], [], []
    tid_of = dict(ln.split() for ln in open(os.path.join(directory, "seqid.map")))
    cur = None
    for ln in open(os.path.join(directory, "prot.fa")):
        ln = ln.strip()
        if ln.startswith(">"):
            cur = ln[1:].split()[0]
            names.append(cur); taxids.append(int(tid_of[cur])); seqs.append([])
        else:
            seqs[-1].append(ln)
    seqs = [np.frombuffer("".join(x).encode(), dtype=np.uint8) for x in seqs]
    nodes, tax_names = [], []
    for ln in open(os.path.join(directory, "nodes.dmp")):
        f = [x.strip() for x in ln.split("|")]
        nodes.append((int(f[0]), int(f[1]), f[2]))
    for ln in open(os.path.join(directory, "names.dmp")):
        f = [x.strip() for x in ln.split("|")]
        tax_names.append((int(f[0]), f[1]))
    return names, taxids, seqs, nodes, tax_names


def golden_1cfr(name, tmp_path):
    src = os.path.join(PROT, name + ".1.cfr")
    if os.path.exists(src):
        return src
    dst = str(tmp_path / (name + ".golden.1.cfr"))
    with gzip.open(src + ".gz", "rb") as fi, open(dst, "wb") as fo:
        shutil.copyfileobj(fi, fo)
    return dst


def assert_same_fields(mine, ref):
    a, b = parse_1cfr(mine, protein=True), parse_1cfr(ref, protein=True)
    assert len(a) == len(b)
    for (na, va), (nb, vb) in zip(a, b):
        assert na == nb
        assert va == vb, f"field {na} differs"


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_native_protein_writer_equals_reference_index_field_by_field(name, tmp_path):
    names, taxids, seqs, nodes, tax_names = read_inputs()
    prefix = str(tmp_path / name)
    rep = capi.build_index(names, taxids, seqs, nodes, tax_names, prefix, protein=True, **VARIANTS[name])
    assert rep["n"] == sum(len(s) + 1 for s in seqs)
    assert_same_fields(prefix + ".1.cfr", golden_1cfr(name, tmp_path))
    assert open(prefix + ".2.cfr", "rb").read() == open(os.path.join(PROT, name + ".2.cfr"), "rb").read()
    meta = open(prefix + ".4.cfr").read().split("\n")
    gold = open(os.path.join(PROT, name + ".4.cfr")).read().split("\n")
    assert meta[:3] == gold[:3] and meta[2] == "sequence_type\tamino_acid"


@pytest.mark.parametrize("name", ["p2", "p4"])
def test_command_line_protein_build(name, tmp_path):
    prefix = str(tmp_path / name)
    cmd = [os.path.join(ROOT, "centrifuger_amd", "bin", "centrifuger-build"), "--protein", "-r", os.path.join(INPUT, "prot.fa"), "--taxonomy-tree",
           os.path.join(INPUT, "nodes.dmp"), "--name-table", os.path.join(INPUT, "names.dmp"), "--conversion-table", os.path.join(INPUT, "seqid.map"),
           "-o", prefix] + CLI_ARGS[name]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    assert_same_fields(prefix + ".1.cfr", golden_1cfr(name, tmp_path))
    assert open(prefix + ".2.cfr", "rb").read() == open(os.path.join(PROT, name + ".2.cfr"), "rb").read()


def test_written_protein_index_classifies_like_the_reference_built_one(tmp_path):
    """The index this writer made is opened and used: the golden translated-search TSV of the p2 index must come out of it."""
    names, taxids, seqs, nodes, tax_names = read_inputs()
    prefix = str(tmp_path / "p2")
    capi.build_index(names, taxids, seqs, nodes, tax_names, prefix, protein=True, ftab_chars=2)
    cli = os.path.join(ROOT, "centrifuger_amd", "bin", "centrifuger")
    mine = subprocess.run([cli, "-x", prefix, "-u", os.path.join(PROT, "se.fa")], check=True, capture_output=True).stdout
    gold = subprocess.run([cli, "-x", os.path.join(PROT, "p2"), "-u", os.path.join(PROT, "se.fa")], check=True, capture_output=True).stdout
    assert mine == gold and mine.count(b"\n") > 10


def test_redundant_proteome_with_characters_outside_the_alphabet(tmp_path):
    """12 near-identical proteomes (long repeats: many doubling rounds, run blocks of 8) written by both builders when the compiled
    reference is present; always: the suffix order of the written index is the naive one (BWT decoded through the image)."""
    rng = np.random.default_rng(9)
    aa = np.frombuffer(b"ARNDCEQGHILKMFPSTWYV", dtype=np.uint8)
    base = [aa[rng.integers(0, 20, size=int(rng.integers(60, 200)))] for _ in range(12)]
    names, taxids, seqs = [], [], []
    nodes, tax_names = [(1, 1, "no rank")], [(1, "root")]
    for s in range(12):
        nodes.append((100 + s, 1, "species")); tax_names.append((100 + s, f"sp {s}"))
        for pi, p in enumerate(base):
            q = p.copy()
            for pos in rng.integers(0, len(q), size=int(rng.integers(0, 2))):
                q[pos] = aa[int(rng.integers(20))]
            names.append(f"S{s}_{pi}"); taxids.append(100 + s); seqs.append(q)
    prefix = str(tmp_path / "red")
    rep = capi.build_index(names, taxids, seqs, nodes, tax_names, prefix, protein=True, ftab_chars=4)
    n = rep["n"]
    # naive suffix order on the REAL codes ($ARNDCEQGHILKMFPSTWYV)
    lut = np.zeros(256, dtype=np.uint8)
    for k, ch in enumerate(b"$ARNDCEQGHILKMFPSTWYV"):
        lut[ch] = k
    t = np.concatenate([np.concatenate([lut[s], np.zeros(1, dtype=np.uint8)]) for s in seqs]).astype(np.uint8)
    assert len(t) == n
    raw = bytes((t + 1).tolist())
    sa = sorted(range(n), key=lambda i: raw[i:])
    want_bwt = np.array([t[p - 1] if p else t[n - 1] for p in sa], dtype=np.uint8)
    fields = dict(parse_1cfr(prefix + ".1.cfr", protein=True))
    assert fields["firstISA"] == sa.index(0)
    cnt = np.bincount(want_bwt, minlength=21)
    C = np.frombuffer(fields["C"], dtype=np.uint64)
    assert (np.diff(C.astype(np.int64)) == cnt).all()
    if have_ref():
        d = tmp_path / "in"
        d.mkdir()
        with open(d / "prot.fa", "w") as f, open(d / "seqid.map", "w") as m:
            for nm, tid, s in zip(names, taxids, seqs):
                f.write(f">{nm}\n{bytes(s).decode()}\n"); m.write(f"{nm}\t{tid}\n")
        with open(d / "nodes.dmp", "w") as f:
            for a, b, r in nodes:
                f.write(f"{a}\t|\t{b}\t|\t{r}\t|\n")
        with open(d / "names.dmp", "w") as f:
            for a, nm in tax_names:
                f.write(f"{a}\t|\t{nm}\t|\t\t|\tscientific name\t|\n")
        ref_prefix = str(tmp_path / "ref")
        subprocess.run([os.path.join(REF_DIR, "centrifuger-build"), "--protein", "-t", "4", "-r", str(d / "prot.fa"), "--taxonomy-tree", str(d / "nodes.dmp"),
                        "--name-table", str(d / "names.dmp"), "--conversion-table", str(d / "seqid.map"), "-o", ref_prefix], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        assert_same_fields(prefix + ".1.cfr", ref_prefix + ".1.cfr")
        assert open(prefix + ".2.cfr", "rb").read() == open(ref_prefix + ".2.cfr", "rb").read()
        assert open(prefix + ".3.cfr", "rb").read() == open(ref_prefix + ".3.cfr", "rb").read()
