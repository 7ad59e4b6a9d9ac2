"""Parity of the HIP path (through the C-ABI) against the golden vectors of the real reference and
against the C oracle on the same inputs.  Integer work: everything must be bit-exact.  -m gpu only."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ora
from centrifuger_amd import capi, synth
from conftest import GOLDEN, REF_DIR, have_ref

pytestmark = pytest.mark.gpu
MAN = json.load(open(os.path.join(GOLDEN, "manifest.json")))
VARIANTS = ["f6", "f6_b1", "f6_b8", "f6_off3", "f10"]


@pytest.fixture(scope="module")
def dev(golden_dir):
    cache = {}

    def get(iname, **kw):
        key = (iname, tuple(sorted(kw.items())))
        if key not in cache:
            idx = capi.Index(os.path.join(golden_dir, iname), capi.default_params(**kw))
            cache[key] = capi.DeviceIndex(idx)
        return cache[key]
    yield get
    for d in cache.values():
        d.close()


def test_native_library_is_the_one_running():
    assert capi.device_count() >= 1
    assert os.path.exists(capi.LIB_PATH)
    maps = open("/proc/self/maps").read()
    capi.lib()
    maps = open("/proc/self/maps").read()
    assert "libcfr_hip.so" in maps


@pytest.mark.parametrize("iname", VARIANTS)
def test_rank_and_access_exhaustive(iname, dev):
    """FMIndex::Rank(c,i,incl/excl) and Sequence::Access(i): the stream oracle/ref_dump.cpp printed."""
    d = dev(iname)
    v = MAN["vectors"][iname + ".rank"]
    step = int(v["arg"])
    n = d.info().n
    pos = np.arange(0, n, step, dtype=np.uint64)
    cols = []
    acc = None
    for incl in (1, 0):
        for ch in b"ACGT":
            r, a = d.rank(np.full(len(pos), ch, dtype=np.uint8), pos, np.full(len(pos), incl, dtype=np.uint8))
            cols.append(r)
            acc = a
    lines = [b"%d %c %d %d %d %d %d %d %d %d\n" % ((int(pos[i]), int(acc[i])) + tuple(int(c[i]) for c in cols)) for i in range(len(pos))]
    out = b"".join(lines)
    assert len(lines) == v["lines"]
    assert hashlib.md5(out).hexdigest() == v["md5"]


@pytest.mark.parametrize("iname", VARIANTS)
def test_derived_tables_are_consistent_with_the_bwt(iname, dev):
    """SA / 2-bit text (list ranking by rulers) and the direct locate (memo, or suffix array + step function) against the BWT
    itself, for every row: the head and the tail of the LF list included (the rows before the first ruler once had no owner).
    Runs in whatever form the process is in (u32 entries, or 36-bit packed entries under CFR_FORCE_WIDE in tests/test_gpu_variants.py)."""
    c = dev(iname).selfcheck()
    assert (c["bad_sa_isa"], c["bad_text"], c["bad_lf"], c["bad_memo"]) == (0, 0, 0, 0), c
    if not os.environ.get("CFR_TEXT_MODE") and not os.environ.get("CFR_PROFILE") and not os.environ.get("CFR_LAYOUT") and not os.environ.get("CFR_STEPS_OFF"):
        assert c["text_tables"]


@pytest.mark.parametrize("iname", VARIANTS)
def test_locate_every_row(iname, dev):
    d = dev(iname)
    v = MAN["vectors"][iname + ".locate"]
    rows = np.arange(0, d.info().n, int(v["arg"]), dtype=np.uint64)
    val, steps = d.locate(rows)
    out = b"".join(b"%d %d %d\n" % (int(rows[i]), int(val[i]), int(steps[i])) for i in range(len(rows)))
    assert hashlib.md5(out).hexdigest() == v["md5"]


@pytest.mark.parametrize("iname", VARIANTS)
@pytest.mark.parametrize("kind", ["bs", "bs_se"])
def test_backward_search_tuples(iname, kind, dev, golden_dir):
    d = dev(iname)
    v = MAN["vectors"][f"{iname}.{kind}"]
    ids, bases, offs = ora.read_fastx(os.path.join(golden_dir, v["arg"]))
    width = d.info().precompute_width
    q_read, q_m = [], []
    for i in range(len(ids)):
        m = int(offs[i + 1] - offs[i])
        while m > 0:
            q_read.append(i); q_m.append(m)
            m = m - 13 if m > 13 else 0
    # one query per (read, m): replicate the read per query through an offsets gather
    q_read = np.array(q_read); q_m = np.array(q_m, dtype=np.uint32)
    lens = (offs[1:] - offs[:-1])[q_read]
    qoffs = np.zeros(len(q_read) + 1, dtype=np.uint64); qoffs[1:] = np.cumsum(lens)
    qbases = np.concatenate([bases[int(offs[r]):int(offs[r + 1])] for r in q_read]) if len(q_read) else np.zeros(0, np.uint8)
    l, sp, ep = d.backward_search(qbases, qoffs, q_m)
    out = []
    for k in range(len(q_read)):
        s, e = (7, 3) if q_m[k] < width else (int(sp[k]), int(ep[k]))   # dump tools start from sp=7, ep=3 (untouched when m < width)
        out.append(b"%d %d %d %d %d\n" % (int(q_read[k]), int(q_m[k]), int(l[k]), s, e))
    assert hashlib.md5(b"".join(out)).hexdigest() == v["md5"]


def _load_case_reads(args, golden_dir):
    if "-u" in args:
        ids, b1, o1 = ora.read_fastx(os.path.join(golden_dir, args[args.index("-u") + 1]))
        return ids, b1, o1, None, None
    ids, b1, o1 = ora.read_fastx(os.path.join(golden_dir, args[args.index("-1") + 1]))
    _, b2, o2 = ora.read_fastx(os.path.join(golden_dir, args[args.index("-2") + 1]))
    return ids, b1, o1, b2, o2


def _case_kw(args):
    kw = {}
    if "-k" in args: kw["max_result"] = int(args[args.index("-k") + 1])
    if "--hitk-factor" in args: kw["max_result_per_hit_factor"] = int(args[args.index("--hitk-factor") + 1])
    if "--min-hitlen" in args: kw["min_hit_len"] = int(args[args.index("--min-hitlen") + 1])
    return kw


@pytest.mark.parametrize("case", sorted(MAN["cases"]))
def test_classification_tsv_equals_reference(case, dev, golden_dir):
    """Full Query on the device (+host tail), formatted like ResultWriter: byte-identical to the reference's TSV."""
    c = MAN["cases"][case]
    d = dev(c["index"], **_case_kw(c["args"]))
    ids, b1, o1, b2, o2 = _load_case_reads(c["args"], golden_dir)
    if "--no-dust" not in c["args"]:
        capi.dust_mask(b1, o1)
        if b2 is not None: capi.dust_mask(b2, o2)
    results, matches = d.classify(b1, o1, b2, o2)
    out = capi.tsv_header() + b"".join(d.index.format_tsv(ids[i], results[i], matches) for i in range(len(ids)))
    want = open(os.path.join(GOLDEN, "tsv", case + ".tsv"), "rb").read()
    assert out == want


@pytest.mark.parametrize("case", ["f6.se_nodust", "f6.pe_default", "f6.edge_nodust", "f6.long_default", "f10.edge_pe_k3", "f6_b8.pe_k5"])
def test_hit_lists_equal_oracle(case, dev, golden_dir):
    """SearchForwardAndReverse output (after AdjustHitBoundary + strand choice): every _BWTHit field."""
    c = MAN["cases"][case]
    kw = _case_kw(c["args"])
    d = dev(c["index"], **kw)
    o = ora.OracleIndex(os.path.join(golden_dir, c["index"]), max_result=kw.get("max_result", 1),
                        min_hit_len=kw.get("min_hit_len", 0), hitk_factor=kw.get("max_result_per_hit_factor", 40))
    ids, b1, o1, b2, o2 = _load_case_reads(c["args"], golden_dir)
    if "--no-dust" not in c["args"]:
        capi.dust_mask(b1, o1)
        if b2 is not None: capi.dust_mask(b2, o2)
    hits, hb = d.search(b1, o1, b2, o2)
    for i in range(len(ids)):
        r1 = b1[int(o1[i]):int(o1[i + 1])].tobytes()
        r2 = None if b2 is None else b2[int(o2[i]):int(o2[i + 1])].tobytes()
        want = o.query_hits(r1, r2)
        got = hits[int(hb[i]):int(hb[i + 1])]
        assert len(got) == len(want), ids[i]
        for f in ("sp", "ep", "l", "strand", "offset"):
            assert np.array_equal(got[f], want[f]), (ids[i], f)


def test_empty_and_degenerate_batches(dev):
    d = dev("f6")
    r, m = d.classify(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    assert len(r) == 0 and len(m) == 0
    # zero-length reads, single base, exactly min_hit_len - 1
    seqs = [b"", b"A", b"ACGT" * 5 + b"AC", b""]
    offs = np.zeros(len(seqs) + 1, np.uint64); offs[1:] = np.cumsum([len(s) for s in seqs])
    bases = np.frombuffer(b"".join(seqs), np.uint8).copy()
    r, m = d.classify(bases, offs)
    assert list(r["n_match"]) == [0, 0, 0, 0] and list(r["query_length"]) == [0, 1, 22, 0]
    r, m = d.classify(bases, offs, bases, offs)
    assert list(r["query_length"]) == [0, 2, 44, 0]


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref (compiled reference) not present")
@pytest.mark.parametrize("paired,k", [(False, 1), (True, 5)])
def test_fresh_index_against_reference_binary_and_oracle(paired, k, tmp_path):
    """A new 3 Mbp index built by the reference's centrifuger-build on this box; 20k reads.
    GPU == C oracle (POD fields) and GPU TSV == reference binary's TSV (dust on, default options)."""
    g = synth.make_genomes(n_species=6, n_strains=4, genome_len=125000, seed=77)
    synth.write_reference_inputs(g, str(tmp_path))
    prefix = str(tmp_path / "idx")
    subprocess.run([os.path.join(REF_DIR, "centrifuger-build"), "-t", "8", "-r", str(tmp_path / "ref.fa"),
                    "--taxonomy-tree", str(tmp_path / "nodes.dmp"), "--name-table", str(tmp_path / "names.dmp"),
                    "--conversion-table", str(tmp_path / "seqid.map"), "-o", prefix], check=True, stderr=subprocess.DEVNULL)
    n = 20000
    if paired:
        r1, r2 = synth.make_pairs(g, n, 150, seed=78)
        synth.write_fastq(r1, str(tmp_path / "r_1.fq"), suffix="/1"); synth.write_fastq(r2, str(tmp_path / "r_2.fq"), suffix="/2")
        cmd = ["-1", str(tmp_path / "r_1.fq"), "-2", str(tmp_path / "r_2.fq")]
    else:
        r1 = synth.make_reads(g, n, 150, seed=78); r2 = None
        synth.write_fastq(r1, str(tmp_path / "r.fq"))
        cmd = ["-u", str(tmp_path / "r.fq")]
    want_tsv = subprocess.run([os.path.join(REF_DIR, "centrifuger"), "-x", prefix, "-t", "4", "-k", str(k)] + cmd,
                              check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
    idx = capi.Index(prefix, capi.default_params(max_result=k))
    d = capi.DeviceIndex(idx)
    b1, o1 = r1.bases.copy(), r1.offsets
    b2, o2 = (r2.bases.copy(), r2.offsets) if paired else (None, None)
    capi.dust_mask(b1, o1, threads=4)
    if paired: capi.dust_mask(b2, o2, threads=4)
    results, matches = d.classify(b1, o1, b2, o2)
    out = capi.tsv_header() + b"".join(idx.format_tsv(f"r{i}", results[i], matches) for i in range(n))
    assert out == want_tsv
    o = ora.OracleIndex(prefix, max_result=k)
    ores = o.classify(b1, o1, b2, o2, dust=False, threads=4)
    for i in range(n):
        assert (results[i]["score"], results[i]["secondary_score"], results[i]["hit_length"], results[i]["n_match"]) == \
               (ores[i].score, ores[i].secondaryScore, ores[i].hitLength, ores[i].nmatch)
    st = d.last_stats()
    # (n_hits / n_rows are only counted by the pipelines that need those totals on the host: the one-launch post stage does not)
    assert st.n_chains == n * (4 if paired else 2)


def test_old_format_index_without_end_marker_field(golden_dir, tmp_path):
    """An index file that ends before the `hasEndMarker` byte (written by older reference versions, FMIndex.hpp:178-181)
    classifies exactly like the full file."""
    import shutil
    full = open(os.path.join(golden_dir, "f6.1.cfr"), "rb").read()
    for ext in (".2.cfr", ".3.cfr", ".4.cfr"):
        if os.path.exists(os.path.join(golden_dir, "f6" + ext)):
            shutil.copy(os.path.join(golden_dir, "f6" + ext), tmp_path / ("old" + ext))
    (tmp_path / "old.1.cfr").write_bytes(full[:-1])
    idx = capi.Index(str(tmp_path / "old"), capi.default_params(max_result=1))
    d = capi.DeviceIndex(idx)
    rs_ids, b, o = [], [], [0]
    for rec in open(os.path.join(golden_dir, "se.fq"), "rb").read().split(b"\n@")[:400]:
        lines = rec.split(b"\n")
        rs_ids.append(lines[0].lstrip(b"@").split()[0].decode())
        b.append(lines[1])
        o.append(o[-1] + len(lines[1]))
    bases = np.frombuffer(b"".join(b), dtype=np.uint8).copy()
    offs = np.array(o, dtype=np.uint64)
    res, mat = d.classify(bases, offs)
    idx_full = capi.Index(os.path.join(golden_dir, "f6"), capi.default_params(max_result=1))
    d_full = capi.DeviceIndex(idx_full)
    res2, mat2 = d_full.classify(bases, offs)
    assert res.tobytes() == res2.tobytes() and mat.tobytes() == mat2.tobytes() and int((res["n_match"] > 0).sum()) > 300
    d.close()
    d_full.close()
