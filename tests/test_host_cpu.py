"""CPU-side checks of the product's host code: the C-ABI surface, the .cfr parser, SDUST, the host
tail (fed with hits / located ids produced by the oracle), TSV formatting.  No GPU needed."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

import oracle_lib as ora
from centrifuger_amd import capi
from conftest import GOLDEN, ROOT

MAN = json.load(open(os.path.join(GOLDEN, "manifest.json")))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "cfr_hip.h")).read()
    declared = set(re.findall(r"\b(cfr_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    L = capi.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/cfr_hip.h but not exported"
    assert declared == set(capi.EXPORTS)


def test_struct_layouts_match_header():
    assert capi.HIT_DTYPE.itemsize == 32 and capi.RESULT_DTYPE.itemsize == 40 and capi.MATCH_DTYPE.itemsize == 24
    assert C.sizeof(capi.Params) == 32
    assert C.sizeof(capi.DeviceOptions) == 32
    o = capi.default_device_options()
    assert (o.profile, o.ftabx_width, o.text_mode, o.run_block_layout, o.sub_batch) == (capi.PROFILE_THROUGHPUT, -1, -1, 0, 0) and o.loc_memo_gb < 0


@pytest.mark.parametrize("iname", ["f6", "f6_b1", "f6_b8", "f6_off3", "f10"])
def test_parser_reads_every_index_variant(iname, golden_dir):
    idx = capi.Index(os.path.join(golden_dir, iname))
    info = idx.info()
    assert info.n == 300000 and info.min_hit_len == 23
    assert info.precompute_width == (10 if iname == "f10" else 6)
    assert info.sample_rate == (8 if iname == "f6_off3" else 16)
    if iname == "f6_b1":
        assert info.block_size == info.n
    if iname == "f6_b8":
        assert info.block_size == 8
    assert info.seq_cnt == 15 and info.selected_cnt == 14


def test_open_errors_are_statuses_not_exits(tmp_path):
    with pytest.raises(capi.CfrError) as e:
        capi.Index(str(tmp_path / "nope"))
    assert e.value.status == capi.CFR_ERR_IO
    bad = tmp_path / "bad"
    (tmp_path / "bad.1.cfr").write_bytes(b"\x01" * 100)
    (tmp_path / "bad.2.cfr").write_bytes(b"\x01" * 100)
    with pytest.raises(capi.CfrError) as e:
        capi.Index(str(bad))
    assert e.value.status == capi.CFR_ERR_FORMAT
    # a protein prefix (.4.cfr says amino_acid) whose .1.cfr holds a nucleotide index is a format error, not a crash
    import shutil
    gold = os.path.join(ROOT, "tests", "golden")
    for ext in (".1.cfr", ".2.cfr"):
        shutil.copy(os.path.join(gold, "f6" + ext), tmp_path / ("prot" + ext))
    (tmp_path / "prot.4.cfr").write_text("version\t1\nsequence_type\tamino_acid\n")
    with pytest.raises(capi.CfrError) as e:
        capi.Index(str(tmp_path / "prot"))
    assert e.value.status in (capi.CFR_ERR_FORMAT, capi.CFR_ERR_IO)


def test_protein_index_is_parsed_and_decoded_on_the_host():
    """FMIndex<Sequence_RunBlockOneTree> files (tests/golden/prot, written by the reference's centrifuger-build --protein):
    the parser accepts them, says so, and infers min-hitlen like Classifier::InferMinHitLen does for 21 symbols from 11 up."""
    prot = os.path.join(ROOT, "tests", "golden", "prot")
    for name in ("p2", "p3_b4", "p2_b1_off2"):
        info = capi.Index(os.path.join(prot, name)).info()
        assert info.is_protein == 1 and info.n > 30000 and info.min_hit_len == 11 and info.selected_cnt == 0
    assert capi.Index(os.path.join(prot, "p2"), capi.default_params(min_hit_len=8)).info().min_hit_len == 8
    assert capi.Index(os.path.join(ROOT, "tests", "golden", "f6")).info().is_protein == 0


def test_device_options_are_validated_before_any_device_work(golden_dir):
    idx = capi.Index(os.path.join(golden_dir, "f6"))
    for bad in (dict(profile=7), dict(ftabx_width=17), dict(ftabx_width=-2)):
        with pytest.raises(capi.CfrError) as e:
            capi.DeviceIndex(idx, 0, capi.default_device_options(**bad))
        assert e.value.status == capi.CFR_ERR_ARG


def test_no_device_is_a_loud_error_not_a_fallback(golden_dir):
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    idx = capi.Index(os.path.join(golden_dir, "f6"))
    with pytest.raises(capi.CfrError) as e:
        capi.DeviceIndex(idx)
    assert e.value.status == capi.CFR_ERR_NO_DEVICE


@pytest.mark.parametrize("fname", ["se.fq", "edge.fa", "long.fq"])
def test_dust_matches_oracle(fname, golden_dir):
    ids, bases, offs = ora.read_fastx(os.path.join(golden_dir, fname))
    masked = capi.dust_mask(bases.copy(), offs, threads=3)
    changed = 0
    for i in range(len(ids)):
        s = bases[int(offs[i]):int(offs[i + 1])].tobytes()
        want = ora.dust_mask(s)
        got = masked[int(offs[i]):int(offs[i + 1])].tobytes()
        assert got == want, ids[i]
        changed += got != s
    if fname == "edge.fa":
        assert changed >= 3     # polyA, dinuc, lowcomplex_mid


def test_dust_random_low_complexity():
    rng = np.random.default_rng(5)
    seqs = []
    for _ in range(300):
        L = int(rng.integers(1, 400))
        unit = bytes(rng.choice(list(b"ACGTN"), size=int(rng.integers(1, 6))).tolist())
        s = bytearray(rng.choice(list(b"ACGT"), size=L).tolist())
        a = int(rng.integers(0, L)); b = int(rng.integers(a, L))
        s[a:b] = (unit * (L // len(unit) + 1))[: b - a]
        seqs.append(bytes(s))
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64); offs[1:] = np.cumsum([len(s) for s in seqs])
    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy()
    masked = capi.dust_mask(bases, offs)
    for i, s in enumerate(seqs):
        assert masked[int(offs[i]):int(offs[i + 1])].tobytes() == ora.dust_mask(s)


def _rows_for_hit(h, max_entries, locate_all, min_hit_len):
    """the row sequence of Classifier.hpp:620-666"""
    if h["l"] < min_hit_len:
        return []
    sp, ep = int(h["sp"]), int(h["ep"])
    rng = ep - sp + 1
    if rng <= max_entries or locate_all:
        return list(range(sp, ep + 1))
    step = -(-rng // max_entries)
    rows = list(range(sp, ep + 1, step))
    resolved = len(rows)
    j = ep
    while sp <= j <= ep:
        rows.append(j); resolved += 1
        if resolved >= max_entries:
            break
        j -= step
    return rows


@pytest.mark.parametrize("case", ["f6.se_default", "f6.pe_k5", "f6.se_hitk2", "f6.edge_pe_k3", "f6.long_default", "f10.se_k5"])
def test_host_tail_from_oracle_hits_reproduces_reference_tsv(case, golden_dir):
    c = MAN["cases"][case]
    args = c["args"]
    kw = {}
    if "-k" in args: kw["max_result"] = int(args[args.index("-k") + 1])
    if "--hitk-factor" in args: kw["hitk_factor"] = int(args[args.index("--hitk-factor") + 1])
    if "--min-hitlen" in args: kw["min_hit_len"] = int(args[args.index("--min-hitlen") + 1])
    prefix = os.path.join(golden_dir, c["index"])
    o = ora.OracleIndex(prefix, **kw)
    params = capi.default_params(max_result=kw.get("max_result", 1), max_result_per_hit_factor=kw.get("hitk_factor", 40),
                                 min_hit_len=kw.get("min_hit_len", 0))
    idx = capi.Index(prefix, params)
    mhl = idx.info().min_hit_len
    if "-u" in args:
        ids, b1, o1 = ora.read_fastx(os.path.join(golden_dir, args[args.index("-u") + 1])); b2 = o2 = None
    else:
        ids, b1, o1 = ora.read_fastx(os.path.join(golden_dir, args[args.index("-1") + 1]))
        _, b2, o2 = ora.read_fastx(os.path.join(golden_dir, args[args.index("-2") + 1]))
    if "--no-dust" not in args:
        capi.dust_mask(b1, o1)
        if b2 is not None: capi.dust_mask(b2, o2)
    max_entries = params.max_result * params.max_result_per_hit_factor
    locate_all = params.max_result_per_hit_factor <= 0 or params.max_result <= 0
    hits, hit_begin, row_begin, row_vals, qlen = [], [0], [0], [], []
    for i in range(len(ids)):
        r1 = b1[int(o1[i]):int(o1[i + 1])].tobytes()
        r2 = None if b2 is None else b2[int(o2[i]):int(o2[i + 1])].tobytes()
        for h in o.query_hits(r1, r2):
            hits.append((h["sp"], h["ep"], h["l"], h["strand"], h["offset"], 0))
            rows = _rows_for_hit(h, max_entries % (1 << 64), locate_all, mhl)
            row_vals.extend(o.locate(r)[0] for r in rows)
            row_begin.append(len(row_vals))
        hit_begin.append(len(hits))
        qlen.append(len(r1) + (len(r2) if r2 is not None else 0))
    results, matches = idx.classify_from_hits(np.array(hits, dtype=capi.HIT_DTYPE), hit_begin, row_begin,
                                              np.array(row_vals, dtype=np.uint64), qlen, threads=3)
    out = capi.tsv_header() + b"".join(idx.format_tsv(ids[i], results[i], matches) for i in range(len(ids)))
    assert out == open(os.path.join(GOLDEN, "tsv", case + ".tsv"), "rb").read()


def test_cli_rejects_out_of_scope_options_and_missing_index(golden_dir):
    """The drop-in command line refuses options outside the path and reports a missing index as an error, both before
    any device work (so this runs without a GPU)."""
    import subprocess
    cli = os.path.join(ROOT, "centrifuger_amd", "bin", "centrifuger")
    if not os.path.exists(cli):
        pytest.skip("CLI not built")
    r = subprocess.run([cli, "-x", os.path.join(golden_dir, "f6"), "-u", os.path.join(golden_dir, "se.fq"), "--merge-readpair"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"not available in this build" in r.stderr
    r = subprocess.run([cli, "-x", "/nonexistent/idx", "-u", os.path.join(golden_dir, "se.fq")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"loading the index" in r.stderr
    r = subprocess.run([cli], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and b"-x FILE: index prefix" in r.stderr       # no arguments: usage, exit 0 (like the reference's CI smoke)
    if capi.device_count() == 0:
        r = subprocess.run([cli, "-x", os.path.join(golden_dir, "f6"), "-u", os.path.join(golden_dir, "se.fq")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode != 0 and b"no CPU fallback" in r.stderr


def test_tail_sorting_network_sorts_every_input():
    """The 8-key comparator network the device tail uses in registers (CFR_NET8 in cfr_kernels.hip.inc), read from the
    source and checked exhaustively on 0/1 inputs (zero-one principle)."""
    import itertools
    import re
    src = open(os.path.join(ROOT, "centrifuger_amd", "csrc", "cfr_kernels.hip.inc")).read()
    body = src[src.index("#define CFR_NET8(CE)"):]
    body = body[:body.index("\n\n")]
    net = [(int(a), int(b)) for a, b in re.findall(r"CE\((\d),(\d)\)", body)]
    assert len(net) == 19
    for bits in itertools.product([0, 1], repeat=8):
        a = list(bits)
        for i, j in net:
            if a[i] > a[j]:
                a[i], a[j] = a[j], a[i]
        assert a == sorted(a)


def _py_parse(data: bytes):
    """Plain restatement of the record grammar the command line accepts (multi-line FASTA/FASTQ, CRLF, blank lines)."""
    lines = [ln.rstrip(b"\r") for ln in data.split(b"\n")]
    if lines and lines[-1] == b"":
        lines.pop()
    recs, i = [], 0
    while i < len(lines):
        ln = lines[i]
        if not ln or ln[:1] not in (b">", b"@"):
            i += 1
            continue
        rid = ln[1:].split(b" ")[0].split(b"\t")[0]
        if len(rid) >= 2 and rid[-2:] in (b"/1", b"/2"):
            rid = rid[:-2]
        i += 1
        seq, qual, hq = b"", b"", False
        while i < len(lines):
            ln = lines[i]
            if not ln:
                i += 1
                continue
            if ln[:1] in (b">", b"@"):          # kseq: a header character at a line start ends the sequence, always
                break
            if ln[:1] == b"+":                  # ... and '+' starts the quality block, whatever the header character was
                hq = True
                i += 1
                while i < len(lines):           # at least one line, then until as long as the sequence
                    qual += lines[i]
                    i += 1
                    if len(qual) >= len(seq):
                        break
                break
            seq += ln
            i += 1
        recs.append((rid, seq, qual, hq))
    return recs


@pytest.mark.parametrize("gz", [False, True])
def test_cli_read_parser(tmp_path, gz):
    """The block parser of the command line (CFR_CLI_PARSE_ONLY hook: no index, no device) on awkward inputs."""
    import gzip
    import subprocess
    cli = os.path.join(ROOT, "centrifuger_amd", "bin", "centrifuger")
    if not os.path.exists(cli):
        pytest.skip("CLI not built")
    rng = np.random.default_rng(5)

    def dna(n):
        return bytes(rng.choice(list(b"ACGTN"), size=n, p=[.24, .24, .24, .24, .04]).astype(np.uint8))
    cases = {
        "fastq4": b"".join(b"@r%d/1 desc x\n%s\n+\n%s\n" % (i, dna(50 + i), b"I" * (50 + i)) for i in range(200)),
        "fastq_at_quality": b"@a\nACGT\n+\n@@@@\n@b extra\nGGCC\n+b\n@III\n",
        "fasta_multiline_crlf": b">s1 first\r\nACGT\r\nGGTT\r\n\r\n>s2\r\nTTTT\r\n",
        "fasta_no_trailing_newline": b">x\nACGTACGT\n>y/2\nGGG",
        "fastq_multiline": b"@m1\nACGT\nACGT\n+\nIIII\nIIII\n@m2\nAC\n+\nII\n",
        "mixed_blank": b"\n\n>f1\nAC\n\nGT\n@q1\nAAAA\n+\nIIII\n>f2\nCC\n",
        "long_record": b">big\n" + b"\n".join(dna(70) for _ in range(3000)) + b"\n>tail\nACGT\n",
        "fastq_empty_sequence": b"@e1\n@e2\nACGT\n+\nIIII\n@e3\n+\n\n@e4\nGG\n+\nII\n",
        "fasta_with_plus_line": b">f\nACGT\n+\nIIII\n>g\nCC\n",
        "huge_line": b">one\n" + dna(40_000_000 if not gz else 400_000) + b"\n>two\nAC\n",
    }
    for name, data in cases.items():
        path = tmp_path / (name + (".gz" if gz else ".txt"))
        if gz:
            with gzip.open(path, "wb", compresslevel=1) as f:
                f.write(data)
        else:
            path.write_bytes(data)
        out = subprocess.run([cli, "-x", "unused", "-u", str(path)], env=dict(os.environ, CFR_CLI_PARSE_ONLY="1"),
                             check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
        want = b"".join(rid + b"\t" + seq + b"\t" + (b"q:" + q if hq else b"-") + b"\n" for rid, seq, q, hq in _py_parse(data))
        assert out == want, name
        if not gz:
            # the parallel cutter (plain files only): pieces of ~64 bytes cut at verified record starts give the same records, or
            # the file is declared not cuttable (multi-line FASTQ, leading blank lines) and the sequential reader takes it
            par = subprocess.run([cli, "-x", "unused", "-u", str(path)], env=dict(os.environ, CFR_CLI_PARSE_ONLY="3", CFR_CLI_PIECE_BYTES="64"),
                                 check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
            assert b"CUT_MISMATCH" not in par, name
            if name in ("fastq4", "fasta_multiline_crlf", "fasta_no_trailing_newline", "long_record", "huge_line", "fastq_at_quality"):
                assert par == want, name
            else:
                assert par == want or par.startswith(b"NOT_CUTTABLE"), name
    # pairs: two files and one interleaved file give the same records
    m1 = b"".join(b"@p%d/1\n%s\n+\n%s\n" % (i, dna(40), b"J" * 40) for i in range(50))
    m2 = b"".join(b"@p%d/2\n%s\n+\n%s\n" % (i, dna(45), b"J" * 45) for i in range(50))
    (tmp_path / "m1.fq").write_bytes(m1)
    (tmp_path / "m2.fq").write_bytes(m2)
    r1, r2 = _py_parse(m1), _py_parse(m2)
    inter = b"".join(b"@%s/1\n%s\n+\n%s\n@%s/2\n%s\n+\n%s\n" % (a[0], a[1], a[2], b[0], b[1], b[2]) for a, b in zip(r1, r2))
    (tmp_path / "inter.fq").write_bytes(inter)
    env = dict(os.environ, CFR_CLI_PARSE_ONLY="1")
    o2 = subprocess.run([cli, "-x", "unused", "-1", str(tmp_path / "m1.fq"), "-2", str(tmp_path / "m2.fq")], env=env, check=True,
                        stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
    oi = subprocess.run([cli, "-x", "unused", "-i", str(tmp_path / "inter.fq")], env=env, check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
    want = b"".join(a[0] + b"\t" + a[1] + b"\t" + b[1] + b"\tq:" + a[2] + b"\n" for a, b in zip(r1, r2))
    assert o2 == want and oi == want
    # the pair cutter (both mate files cut at the same record numbers, the interleaved file at even ones; plain files of regular
    # shape only): pieces of 7 pairs give the same records; FASTA pairs too; irregular files are declared not cuttable
    for every in ("7", "1", "50", "64"):
        env4 = dict(os.environ, CFR_CLI_PARSE_ONLY="4", CFR_CLI_PIECE_RECORDS=every)
        p2 = subprocess.run([cli, "-x", "unused", "-1", str(tmp_path / "m1.fq"), "-2", str(tmp_path / "m2.fq")], env=env4, check=True,
                            stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
        pi = subprocess.run([cli, "-x", "unused", "-i", str(tmp_path / "inter.fq")], env=env4, check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
        assert p2 == want and pi == want, every
    fa1 = b"".join(b">p%d/1\n%s\n%s\n" % (i, r1[i][1][:30], r1[i][1][30:]) for i in range(50))        # multi-line FASTA records
    fa2 = b"".join(b">p%d/2\n%s\n" % (i, r2[i][1]) for i in range(50))[:-1]                            # no trailing newline
    (tmp_path / "m1.fa").write_bytes(fa1)
    (tmp_path / "m2.fa").write_bytes(fa2)
    wantfa = b"".join(a[0] + b"\t" + a[1] + b"\t" + b[1] + b"\t-\n" for a, b in zip(r1, r2))
    pfa = subprocess.run([cli, "-x", "unused", "-1", str(tmp_path / "m1.fa"), "-2", str(tmp_path / "m2.fa")], env=dict(os.environ, CFR_CLI_PARSE_ONLY="4", CFR_CLI_PIECE_RECORDS="9"),
                         check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
    assert pfa == wantfa
    (tmp_path / "m1_multi.fq").write_bytes(b"@m1\nACGT\nACGT\n+\nIIII\nIIII\n@m2\nAC\n+\nII\n" * 10)   # multi-line FASTQ: 10 lines per two records
    (tmp_path / "m2_short.fq").write_bytes(m2[:len(m2) // 2 + 3])
    for a_, b_ in (("m1_multi.fq", "m2.fq"), ("m1.fq", "m2_short.fq")):
        bad = subprocess.run([cli, "-x", "unused", "-1", str(tmp_path / a_), "-2", str(tmp_path / b_)], env=dict(os.environ, CFR_CLI_PARSE_ONLY="4"),
                             check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
        assert bad.startswith(b"NOT_CUTTABLE"), (a_, b_)


def test_index_written_before_the_end_marker_field(golden_dir, tmp_path, oracle_bin):
    """Indexes written before `hasEndMarker` existed end one byte earlier (FMIndex.hpp:178-181 loads them with the flag
    false).  The parser accepts such a file, and the oracle classifies with it exactly as with the full one."""
    import shutil
    import subprocess
    full = open(os.path.join(golden_dir, "f6.1.cfr"), "rb").read()
    assert full[-1] == 0                                  # the flag of a nucleotide index
    for ext in (".2.cfr", ".3.cfr", ".4.cfr"):
        if os.path.exists(os.path.join(golden_dir, "f6" + ext)):
            shutil.copy(os.path.join(golden_dir, "f6" + ext), tmp_path / ("old" + ext))
    (tmp_path / "old.1.cfr").write_bytes(full[:-1])
    a, b = capi.Index(os.path.join(golden_dir, "f6")).info(), capi.Index(str(tmp_path / "old")).info()
    for f in ("n", "block_size", "precompute_width", "sample_rate", "seq_cnt", "selected_cnt", "min_hit_len"):
        assert getattr(a, f) == getattr(b, f)
    reads = os.path.join(golden_dir, "se.fq")
    out_full = subprocess.run([oracle_bin, "classify", "-x", os.path.join(golden_dir, "f6"), "-u", reads], check=True, stdout=subprocess.PIPE).stdout
    out_old = subprocess.run([oracle_bin, "classify", "-x", str(tmp_path / "old"), "-u", reads], check=True, stdout=subprocess.PIPE).stdout
    assert out_old == out_full and out_full.count(b"\n") > 100
    # a file cut anywhere else is a format error, not a crash
    for cut in (len(full) // 3, len(full) - 9):
        (tmp_path / "old.1.cfr").write_bytes(full[:cut])
        with pytest.raises(capi.CfrError) as e:
            capi.Index(str(tmp_path / "old"))
        assert e.value.status in (capi.CFR_ERR_FORMAT, capi.CFR_ERR_IO)


def test_bounded_dust_equals_the_literal_scan():
    """cfr_dust_mask_batch runs the bounded form of the scan (one slot per interval start); the literal form keeps the
    reference's list.  Same masks on random, low-complexity and adversarial reads, and the bounded form does not blow up on
    homopolymers (the literal one takes milliseconds per poly-A read)."""
    import time
    rng = np.random.default_rng(11)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    reads = []
    for i in range(6000):
        L = int(rng.integers(0, 400))
        kind = i % 8
        if kind == 0:
            r = acgt[rng.integers(0, 4, size=L)]
        elif kind == 1:
            r = np.full(L, acgt[rng.integers(0, 4)], dtype=np.uint8)
        elif kind == 2:
            r = np.resize(acgt[rng.integers(0, 4, size=int(rng.integers(2, 7)))], L)
        elif kind == 3:
            r = acgt[rng.choice(4, size=L, p=[0.85, 0.05, 0.05, 0.05])]
        elif kind == 4:
            r = acgt[rng.integers(0, 4, size=L)].copy()
            if L > 100:
                a = int(rng.integers(0, L - 90))
                r[a:a + int(rng.integers(1, 90))] = ord("N")
        elif kind == 5:
            r = acgt[rng.choice(2, size=L)]
        elif kind == 6:
            r = np.concatenate([np.full(L // 2, ord("T"), dtype=np.uint8), acgt[rng.integers(0, 4, size=L - L // 2)]])
        else:
            r = np.frombuffer(bytes(acgt[rng.choice(4, size=L, p=[0.7, 0.1, 0.1, 0.1])]).lower(), dtype=np.uint8)
        reads.append(np.ascontiguousarray(r, dtype=np.uint8))
    reads.append(np.full(5000, ord("A"), dtype=np.uint8))
    reads.append(np.resize(np.frombuffer(b"ACG", dtype=np.uint8), 4000))
    b = np.concatenate(reads)
    o = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.uint64)
    t0 = time.perf_counter()
    fast = capi.dust_mask(b.copy(), o, threads=4)
    t_fast = time.perf_counter() - t0
    t0 = time.perf_counter()
    lit = capi.dust_mask(b.copy(), o, threads=4, literal=True)
    t_lit = time.perf_counter() - t0
    assert np.array_equal(fast, lit)
    assert int((lit != b).sum()) > 100_000
    assert t_fast < t_lit


def test_bench_gpus_flag_is_honoured():
    """`bench.py --gpus N` must mean N ranks: a launcher that started a different WORLD_SIZE is an error, and without a
    launcher the script starts the ranks itself (here the ranks then stop at "needs an MI355X": there is no CPU path)."""
    import subprocess
    import sys
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, bench, "--gpus", "2"], env=dict(env, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode != 0 and b"--gpus 2 but the launcher started WORLD_SIZE=3" in r.stderr
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600)
    assert b"spawning 2 ranks" in r.stderr
    import torch
    if not torch.cuda.is_available():
        # (the launcher stops the other rank as soon as the first one has failed: one message is certain, two are usual)
        assert r.returncode != 0 and r.stderr.count(b"needs an MI355X") >= 1


def test_host_dust_twins_and_oracle_equal_the_reference_dumps():
    """tests/golden/dust: the masked reads the REFERENCE wrote into its --un/--cl dumps for 1580 adversarial reads; the oracle's
    SDUST, the bounded host twin and the literal host twin must all reproduce them byte for byte."""
    import gzip
    ids, b, o = ora.read_fastx(os.path.join(ROOT, "tests", "golden", "dust", "reads.fa"))
    lines = gzip.open(os.path.join(ROOT, "tests", "golden", "dust", "masked_by_reference.fa.gz"), "rb").read().split(b"\n")
    want = {lines[i][1:].decode(): lines[i + 1] for i in range(0, len(lines) - 1, 2)}
    assert len(want) == len(ids) == 1580
    for literal in (False, True):
        got = b.copy()
        capi.dust_mask(got, o, threads=4, literal=literal)
        for i, rid in enumerate(ids):
            assert bytes(got[int(o[i]):int(o[i + 1])]) == want[rid], (rid, literal)
    for i, rid in enumerate(ids):
        s = bytes(b[int(o[i]):int(o[i + 1])])
        if len(s) < 3000:                      # (the oracle's literal list is slow on the 6 kbp homopolymer; the twins cover it)
            assert ora.dust_mask(s) == want[rid], rid


def _dust_set2():
    import gzip
    import tempfile
    raw = gzip.open(os.path.join(ROOT, "tests", "golden", "dust", "reads2.fa.gz"), "rb").read()
    with tempfile.NamedTemporaryFile(suffix=".fa", delete=False) as f:
        f.write(raw)
        path = f.name
    ids, b, o = ora.read_fastx(path)
    os.unlink(path)
    lines = gzip.open(os.path.join(ROOT, "tests", "golden", "dust", "masked2_by_reference.fa.gz"), "rb").read().split(b"\n")
    want = {lines[i][1:].decode(): lines[i + 1] for i in range(0, len(lines) - 1, 2)}
    return ids, b, o, want


def test_host_dust_twins_equal_the_second_set_of_reference_dumps():
    """tests/golden/dust set 2 (round 3, tests/golden/make_golden_dust.py set2): 4 300 reads the REFERENCE masked - runs of
    non-symbols of 1..200 at every offset 0..69 (a run of more than 64 closes a segment), runs at the ends of a read, low
    complexity directly beside a run, 60 long reads (1-8 kbp) of mixed stretches, 3 000 reads of the bench's 150 bp.  Both host
    twins reproduce every one of them; the oracle's literal scan is checked on the short ones."""
    ids, b, o, want = _dust_set2()
    assert len(ids) == len(want) > 4000
    assert sum(1 for i, rid in enumerate(ids) if bytes(b[int(o[i]):int(o[i + 1])]) != want[rid]) > 1500      # the set does get masked
    for literal in (False, True):
        got = b.copy()
        capi.dust_mask(got, o, threads=4, literal=literal)
        bad = [rid for i, rid in enumerate(ids) if bytes(got[int(o[i]):int(o[i + 1])]) != want[rid]]
        assert not bad, (literal, bad[:10])
    for i, rid in enumerate(ids[:1500]):
        s_ = bytes(b[int(o[i]):int(o[i + 1])])
        if len(s_) < 400:
            assert ora.dust_mask(s_) == want[rid], rid


def test_digest_is_stable_and_tells_indexes_apart():
    """cfr_index_digest: equal for every open of the same files, different between the golden index variants"""
    prot = os.path.join(GOLDEN, "prot")
    names = [os.path.join(GOLDEN, x) for x in ("f6", "f6_b1", "f6_b8", "f6_off3")] + [os.path.join(prot, x) for x in ("p2", "p2_b1_off2", "p3_b4")]
    first = [capi.Index(p).digest() for p in names]
    assert len(set(first)) == len(first)
    for _ in range(20):
        assert [capi.Index(p).digest() for p in names] == first


def test_pack_reads_equals_the_definition():
    """cfr_pack_reads: block b = characters [16 b, 16 b + 16): 2-bit codes (A 0, C 1, G 2, T 3) at bits 2j, validity of upper-case ACGT at bit
    32 + j, nothing else set; bytes past the end are not symbols; one thread and many give the same blocks."""
    rng = np.random.default_rng(3)
    for total in (0, 1, 15, 16, 17, 31, 33, 1000, 70001):
        b = rng.choice(np.frombuffer(b"ACGTNacgtRY-\x00\xff", dtype=np.uint8), size=total, p=[.22, .22, .22, .22] + [.12 / 10] * 10)
        got = capi.pack_reads(b, threads=1)
        assert np.array_equal(got, capi.pack_reads(b, threads=7))
        nblk = (total + 15) // 16
        assert len(got) == nblk
        pad = np.zeros(nblk * 16, dtype=np.uint8)
        pad[:total] = b
        code = np.full(256, 0, dtype=np.uint64)
        valid = np.zeros(256, dtype=np.uint64)
        for k, ch in enumerate(b"ACGT"):
            code[ch] = k
            valid[ch] = 1
        v = pad.reshape(nblk, 16)
        want = np.zeros(nblk, dtype=np.uint64)
        for j in range(16):
            want |= valid[v[:, j]] << np.uint64(32 + j)
        assert np.array_equal(got >> np.uint64(32), want >> np.uint64(32))
        for j in range(16):        # code bits only where the character is a symbol
            ok = valid[v[:, j]] == 1
            assert np.array_equal(((got >> np.uint64(2 * j)) & np.uint64(3))[ok], code[v[:, j]][ok])


def test_mapped_and_copied_opens_are_the_same_index(golden_dir, tmp_path):
    """Round 5: cfr_index_open leaves the bit strings in the read-only mapping of the .1.cfr file (the ranks of one node share the page
    cache's pages) where round 4 copied them into the process; CFR_INDEX_COPY=1 is the old form.  Same digest, same host tail; and the
    mapped index keeps working after the file's NAME is gone (the mapping holds the inode)."""
    import shutil
    for iname in ("f6", "f6_b1", "f10"):
        src = os.path.join(golden_dir, iname)
        a = capi.Index(src, capi.default_params())
        os.environ["CFR_INDEX_COPY"] = "1"
        try:
            b = capi.Index(src, capi.default_params())
        finally:
            del os.environ["CFR_INDEX_COPY"]
        assert a.digest() == b.digest()
        # the open REALLY maps: every bit string, at whatever offset the file has it (none of them is 8-byte aligned there - until the end of
        # round 5 only aligned ones were left in the mapping, i.e. none, and nothing noticed); the copying form holds them all
        ma, ca = a.mapped_bytes()
        mb, cb = b.mapped_bytes()
        assert ca == 0 and ma > 0 and mb == 0 and cb == ma, (ma, ca, mb, cb)
        a.close(); b.close()
    for k in (1, 2, 4):
        shutil.copy(os.path.join(golden_dir, f"f6.{k}.cfr"), tmp_path / f"gone.{k}.cfr")
    idx = capi.Index(str(tmp_path / "gone"), capi.default_params())
    want = idx.digest()
    for k in (1, 2, 4):
        os.unlink(tmp_path / f"gone.{k}.cfr")
    assert idx.digest() == want
    idx.close()


def write_bgzf(path, data, block=0xff00, level=1, eof_marker=True):
    """a BGZF file (bgzip's format, SAM specification 4.1): independent deflate blocks, each a gzip member whose extra field 'BC'
    carries the member's size"""
    import struct
    import zlib
    with open(path, "wb") as f:
        pieces = [data[o:o + block] for o in range(0, len(data), block)] + ([b""] if eof_marker else [])
        for piece in pieces:
            co = zlib.compressobj(level, zlib.DEFLATED, -15)
            comp = co.compress(piece) + co.flush()
            bsize = 12 + 6 + len(comp) + 8
            f.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1))
            f.write(comp + struct.pack("<II", zlib.crc32(piece) & 0xffffffff, len(piece)))


def test_bgzf_input_is_inflated_by_several_threads(tmp_path):
    """A bgzip'd read file: the blocks are inflated side by side (-t > 1) and reach the line splitter in file order; the same records
    as the plain file and as the same file through gzread (-t 1); a damaged block is an error, not a silent gap."""
    import gzip
    import subprocess
    cli = os.path.join(ROOT, "centrifuger_amd", "bin", "centrifuger")
    if not os.path.exists(cli):
        pytest.skip("CLI not built")
    rng = np.random.default_rng(11)
    recs = []
    for i in range(60000):
        n = int(rng.integers(30, 260))
        recs.append(b"@r%d/1 x\n%s\n+\n%s\n" % (i, bytes(rng.choice(list(b"ACGTN"), size=n).astype(np.uint8)), b"I" * n))
    data = b"".join(recs)                                  # ~19 MB: three groups of blocks
    (tmp_path / "plain.fq").write_bytes(data)
    write_bgzf(tmp_path / "reads.fq.gz", data)
    assert gzip.open(tmp_path / "reads.fq.gz", "rb").read() == data          # (what any gzip reader makes of it)
    env = dict(os.environ, CFR_CLI_PARSE_ONLY="1")
    run = lambda f, t: subprocess.run([cli, "-x", "unused", "-u", str(f), "-t", str(t)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    want = run(tmp_path / "plain.fq", 1).stdout
    assert want.count(b"\n") == 60000
    for t in (1, 2, 8, 64):
        r = run(tmp_path / "reads.fq.gz", t)
        assert r.returncode == 0 and r.stdout == want, t
    write_bgzf(tmp_path / "tiny.fq.gz", data[:1000], block=100, eof_marker=False)      # many small blocks, no end-of-file marker
    assert run(tmp_path / "tiny.fq.gz", 8).stdout == run(tmp_path / "tiny.fq.gz", 1).stdout != b""
    bad = bytearray((tmp_path / "reads.fq.gz").read_bytes())
    bad[len(bad) // 2] ^= 0x55
    (tmp_path / "bad.fq.gz").write_bytes(bytes(bad))
    r = run(tmp_path / "bad.fq.gz", 8)
    assert r.returncode != 0 and b"BGZF" in r.stderr
    # a file that continues as a plain gzip member (`cat a.bgz b.gz`): read like any .gz, as the reference's gzread does
    half = data[:len(data) // 2]
    half = half[:half.rfind(b"@r")]
    rest = data[len(half):]
    write_bgzf(tmp_path / "first.fq.gz", half, eof_marker=False)
    (tmp_path / "mixed.fq.gz").write_bytes((tmp_path / "first.fq.gz").read_bytes() + gzip.compress(rest, 1))
    for t in (1, 8):
        r = run(tmp_path / "mixed.fq.gz", t)
        assert r.returncode == 0 and r.stdout == want, t


def test_format_tsv_always_terminates_its_buffer(golden_dir):
    """cfr_format_tsv with a buffer that is too small: the return value says what is needed and what was written is a C string (ADVICE r5:
    the hand-rolled formatter left a truncated buffer unterminated)"""
    idx = capi.Index(os.path.join(golden_dir, "f6"))
    r = np.zeros(1, dtype=capi.RESULT_DTYPE)
    r["query_length"] = 150
    m = np.zeros(1, dtype=capi.MATCH_DTYPE)
    full = idx.format_tsv("read_with_a_long_name", r[0], m)
    assert full.endswith(b"\t150\t1\n")
    for cap in (1, 2, 7, len(full) - 1, len(full)):
        buf = C.create_string_buffer(b"\xff" * 64, 64)
        need = capi.lib().cfr_format_tsv(idx._h, b"read_with_a_long_name", r.ctypes.data_as(C.c_void_p), m.ctypes.data_as(C.c_void_p), buf, C.c_size_t(cap))
        assert need == len(full)
        assert buf.raw[:cap - 1] == full[:cap - 1] and buf.raw[cap - 1] == 0 and buf.raw[cap] == 0xff, cap
    buf = C.create_string_buffer(b"\xff" * 64, 64)
    capi.lib().cfr_format_tsv(idx._h, b"read_with_a_long_name", r.ctypes.data_as(C.c_void_p), m.ctypes.data_as(C.c_void_p), buf, C.c_size_t(len(full) + 1))
    assert buf.raw[:len(full) + 1] == full + b"\0"
