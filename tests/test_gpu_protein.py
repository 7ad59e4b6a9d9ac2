"""Protein indexes (FMIndex<Sequence_RunBlockOneTree>, translated search) through the C-ABI against what the REAL reference
produced (tests/golden/prot, made by tests/golden/make_golden_protein.py): Rank / Access of every row and symbol,
BackwardToSampledSA of every row (end-marker rows included), 22 TSVs byte for byte, the command line.  -m gpu."""
import gzip
import hashlib
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import oracle_lib as ora
from centrifuger_amd import capi
from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
PROT = os.path.join(GOLDEN, "prot")
MAN = json.load(open(os.path.join(PROT, "manifest.json")))
ALPHA = b"$ARNDCEQGHILKMFPSTWYV"


@pytest.fixture(scope="module")
def prot_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("prot")
    for f in os.listdir(PROT):
        src = os.path.join(PROT, f)
        if f.endswith(".cfr.gz"):
            with gzip.open(src, "rb") as fi, open(d / f[:-3], "wb") as fo:
                shutil.copyfileobj(fi, fo)
        elif os.path.isfile(src):
            os.symlink(src, d / f)
    return str(d)


def _parse_args(args):
    kw, files = {}, {}
    it = iter(args)
    for a in it:
        if a == "-k": kw["max_result"] = int(next(it))
        elif a == "--min-hitlen": kw["min_hit_len"] = int(next(it))
        elif a == "--hitk-factor": kw["max_result_per_hit_factor"] = int(next(it))
        elif a in ("-u", "-1", "-2"): files[a] = next(it)
    return kw, files


@pytest.mark.parametrize("iname", sorted(MAN["indexes"]))
def test_protein_rank_access_locate_every_row(iname, prot_dir):
    idx = capi.Index(os.path.join(prot_dir, iname))
    assert idx.info().is_protein == 1
    dev = capi.DeviceIndex(idx)
    n = dev.info().n
    v = MAN["vectors"][iname + ".prank"]
    step = int(v["arg"])
    pos = np.arange(0, n, step, dtype=np.uint64)
    cols, acc = [], None
    for incl in (1, 0):
        for ch in ALPHA:
            r, a = dev.rank(np.full(len(pos), ch, dtype=np.uint8), pos, np.full(len(pos), incl, dtype=np.uint8))
            cols.append(r)
            acc = a
    out = b"".join(b"%d %c " % (int(pos[i]), int(acc[i])) + b" ".join(b"%d" % int(c[i]) for c in cols) + b"\n" for i in range(len(pos)))
    assert out.count(b"\n") == v["lines"]
    assert hashlib.md5(out).hexdigest() == v["md5"]
    v = MAN["vectors"][iname + ".plocate"]
    rows = np.arange(0, n, int(v["arg"]), dtype=np.uint64)
    val, steps = dev.locate(rows)
    out = b"".join(b"%d %d %d\n" % (int(rows[i]), int(val[i]), int(steps[i])) for i in range(len(rows)))
    assert hashlib.md5(out).hexdigest() == v["md5"]
    assert dev.selfcheck()["bad_memo"] == 0
    dev.close()


@pytest.mark.parametrize("case", sorted(MAN["cases"]))
def test_protein_tsv_equals_reference(case, prot_dir):
    c = MAN["cases"][case]
    kw, files = _parse_args(c["args"])
    idx = capi.Index(os.path.join(prot_dir, c["index"]), capi.default_params(**kw))
    dev = capi.DeviceIndex(idx)
    dev.set_dust(True)                       # must be ignored for a protein index (CentrifugerClass.cpp:276)
    if "-u" in files:
        ids, b1, o1 = ora.read_fastx(os.path.join(prot_dir, files["-u"]))
        res, mat = dev.classify(b1, o1)
    else:
        ids, b1, o1 = ora.read_fastx(os.path.join(prot_dir, files["-1"]))
        _, b2, o2 = ora.read_fastx(os.path.join(prot_dir, files["-2"]))
        res, mat = dev.classify(b1, o1, b2, o2)
    got = capi.tsv_header() + b"".join(idx.format_tsv(ids[i], res[i], mat) for i in range(len(ids)))
    want = open(os.path.join(PROT, "tsv", case + ".tsv"), "rb").read()
    assert got == want
    dev.close()


@pytest.mark.parametrize("case", ["p2.se_k5", "p3_b4.pe_k3", "p4.edge_k4"])
def test_protein_command_line_equals_reference(case, prot_dir):
    c = MAN["cases"][case]
    args = [os.path.join(prot_dir, a) if a.endswith(".fa") else a for a in c["args"]]
    out = subprocess.run([os.path.join(ROOT, "centrifuger_amd", "bin", "centrifuger"), "-x", os.path.join(prot_dir, c["index"]), "-t", "4"] + args,
                         check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
    assert out == open(os.path.join(PROT, "tsv", case + ".tsv"), "rb").read()


def test_protein_hits_equal_oracle(prot_dir):
    """the hit lists (sp, ep, l, strand, offset in amino acids) of the translated search, read by read, against the C oracle"""
    prefix = os.path.join(prot_dir, "p3_b4")
    idx = capi.Index(prefix, capi.default_params(max_result=3))
    dev = capi.DeviceIndex(idx)
    o = ora.OracleIndex(prefix, max_result=3)
    ids, b1, o1 = ora.read_fastx(os.path.join(prot_dir, "edge.fa"))
    _, b2, o2 = ora.read_fastx(os.path.join(prot_dir, "edge_2.fa"))
    hits, hb = dev.search(b1, o1, b2, o2)
    for i in range(len(ids)):
        r1 = bytes(b1[int(o1[i]):int(o1[i + 1])])
        r2 = bytes(b2[int(o2[i]):int(o2[i + 1])])
        want = o.query_hits(r1, r2)
        got = hits[int(hb[i]):int(hb[i + 1])]
        assert len(got) == len(want), ids[i]
        for f in ("sp", "ep", "l", "strand", "offset"):
            assert np.array_equal(got[f], want[f]), (ids[i], f)
    dev.close()


def test_translation_of_odd_reads_equals_oracle(prot_dir):
    """k_translate_prot's two forms (reads staged in LDS up to 1024 characters, longer ones straight from memory) on reads of awkward
    lengths and characters - empty, 1-5 bases, around the staging limit, several thousand bases, lower case, N, IUPAC letters -
    through the hit lists of the translated search against the C oracle (DnaToAa's ladder decides what an odd character becomes)."""
    prefix = os.path.join(prot_dir, "p3_b4")
    idx = capi.Index(prefix, capi.default_params(max_result=3))
    dev = capi.DeviceIndex(idx)
    o = ora.OracleIndex(prefix, max_result=3)
    _, b, offs = ora.read_fastx(os.path.join(prot_dir, "se.fa"))
    base = [bytes(b[int(offs[i]):int(offs[i + 1])]) for i in range(min(40, len(offs) - 1))]
    rng = np.random.default_rng(11)
    reads = [b"", b"A", b"AC", b"ACG", b"ACGT", b"ACGTA", b"N" * 70, b"acgt" * 30]
    joined = b"".join(base)
    for L in (149, 150, 151, 1022, 1023, 1024, 1025, 1026, 3000, len(joined)):
        reads.append(joined[:L])
        reads.append(joined[7:7 + L])
    for r in base[:12]:
        a = bytearray(r)
        for ch in (b"N", b"n", b"R", b"a", b"t", b"-"):
            a[int(rng.integers(0, len(a)))] = ch[0]
        reads.append(bytes(a))
    long_odd = bytearray(joined[:2500])
    for p in rng.integers(0, len(long_odd), size=40):
        long_odd[int(p)] = b"NnRYacgt"[int(rng.integers(0, 8))]
    reads.append(bytes(long_odd))
    o1 = np.zeros(len(reads) + 1, dtype=np.uint64)
    o1[1:] = np.cumsum([len(r) for r in reads])
    b1 = np.frombuffer(b"".join(reads), dtype=np.uint8).copy()
    hits, hb = dev.search(b1, o1)
    for i, r in enumerate(reads):
        want = o.query_hits(r)
        got = hits[int(hb[i]):int(hb[i + 1])]
        assert len(got) == len(want), (i, len(r))
        for f in ("sp", "ep", "l", "strand", "offset"):
            assert np.array_equal(got[f], want[f]), (i, len(r), f)
    # and the classification itself (text mode, hits in text-position space): the TSV lines against the oracle's
    res, mat = dev.classify(b1, o1)
    ids = [f"r{i}" for i in range(len(reads))]
    want_tsv = o.tsv(ids, o.classify(b1, o1))
    got_tsv = capi.tsv_header() + b"".join(idx.format_tsv(ids[i], res[i], mat) for i in range(len(reads)))
    assert got_tsv == want_tsv
    dev.close()
