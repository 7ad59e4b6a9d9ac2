"""The native index writer (cfr_build_index: suffix array in HBM, csrc/cfr_build_sa.hip + cfr_build.cpp) against the indexes
the REAL reference's centrifuger-build wrote for the same genomes (tests/golden/*.cfr), against the Python restatement of
the writer (centrifuger_amd/indexbuild.py, CPU torch), and against a suffix array sorted naively.  -m gpu."""
import json
import os
import subprocess

import numpy as np
import pytest

from centrifuger_amd import capi, synth
from cfr_fields import parse_1cfr
from conftest import GOLDEN, REF_DIR, ROOT, have_ref

pytestmark = pytest.mark.gpu
MAN = json.load(open(os.path.join(GOLDEN, "manifest.json")))
VARIANTS = {"f6": dict(ftab_chars=6), "f6_b1": dict(ftab_chars=6, rbbwt_b=1), "f6_b8": dict(ftab_chars=6, rbbwt_b=8),
            "f6_off3": dict(ftab_chars=6, offrate=3), "f10": dict()}


@pytest.fixture(scope="module")
def genomes():
    return synth.make_genomes(n_species=5, n_strains=3, genome_len=20000, seed=MAN["seed"])


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_native_writer_equals_reference_index_field_by_field(name, genomes, golden_dir, tmp_path):
    g = genomes
    prefix = str(tmp_path / name)
    rep = capi.build_index(g.names, g.taxids, g.seqs, g.nodes, g.tax_names, prefix, **VARIANTS[name])
    assert rep["n"] == g.total_len
    mine = parse_1cfr(prefix + ".1.cfr")
    ref = parse_1cfr(os.path.join(golden_dir, name + ".1.cfr"))
    assert len(mine) == len(ref)
    for (na, va), (nb, vb) in zip(mine, ref):
        assert na == nb
        assert va == vb, f"field {na} differs"
    assert open(prefix + ".2.cfr", "rb").read() == open(os.path.join(golden_dir, name + ".2.cfr"), "rb").read()


BUILD_MODES = {"default": {}, "many_chunks": {"CFR_BUILD_CHUNK_LOG2": "12"}, "host_sa_many_spans": {"CFR_BUILD_HOST_SA": "1", "CFR_BUILD_CHUNK_LOG2": "13"}}


class _env:
    def __init__(self, kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = os.environ.get(k)
            os.environ[k] = v

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("mode", ["many_chunks", "host_sa_many_spans"])
def test_writer_with_many_chunks_and_with_the_suffix_array_on_the_host(mode, genomes, golden_dir, tmp_path):
    """The same 300 kbp text with 4096-row chunks (dozens of phase-1 chunks and phase-2 spans), and with the suffix array kept in
    host memory (the form texts beyond ~24 Gbp take): the files must not change."""
    g = genomes
    prefix = str(tmp_path / mode)
    with _env(BUILD_MODES[mode]):
        capi.build_index(g.names, g.taxids, g.seqs, g.nodes, g.tax_names, prefix, ftab_chars=6)
    mine, ref = parse_1cfr(prefix + ".1.cfr"), parse_1cfr(os.path.join(golden_dir, "f6.1.cfr"))
    assert len(mine) == len(ref)
    for (na, va), (nb, vb) in zip(mine, ref):
        assert na == nb and va == vb, f"field {na} differs"


def _naive_bwt(t):
    s = bytes(t.tolist())
    sa = sorted(range(len(s)), key=lambda i: s[i:])
    return sa


@pytest.mark.parametrize("mode", sorted(BUILD_MODES))
@pytest.mark.parametrize("kind", ["random", "long_repeat", "homopolymer_tail", "duplicate_genome", "periodic"])
def test_suffix_order_on_adversarial_texts(kind, mode, tmp_path):
    """Texts that stress the doubling rounds and the end-of-text rule (a proper prefix sorts first): the written BWT /
    firstISA must be those of the naively sorted suffixes."""
    rng = np.random.default_rng(7)
    n = 5000
    t = rng.integers(0, 4, size=n, dtype=np.uint8)
    if kind == "long_repeat":
        t[1000:3000] = t[3000:5000]
    elif kind == "homopolymer_tail":
        t[-700:] = 0                      # ...AAAA at the very end: every suffix of the tail is a prefix of the previous one
        t[2000:2600] = 0
    elif kind == "duplicate_genome":
        t[2500:] = t[:2500]
    elif kind == "periodic":
        t[:] = np.tile(np.array([0, 1, 0, 1, 1, 3], dtype=np.uint8), n // 6 + 1)[:n]
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = [acgt[t[:n // 2]], acgt[t[n // 2:]]]
    g = synth.Genomes(["a", "b"], [1000, 1001], seqs, [(1, 1, "no rank"), (1000, 1, "species"), (1001, 1, "species")], [(1, "root"), (1000, "x"), (1001, "y")])
    prefix = str(tmp_path / kind)
    with _env(BUILD_MODES[mode]):
        capi.build_index(g.names, g.taxids, g.seqs, g.nodes, g.tax_names, prefix, ftab_chars=4, rbbwt_b=1)
    sa = _naive_bwt(t)
    want_first_isa = sa.index(0)
    want_bwt = [int(t[p - 1]) if p else int(t[n - 1]) for p in sa]
    idx = capi.Index(prefix)
    assert idx.info().first_isa == want_first_isa
    dev = capi.DeviceIndex(idx)
    _, acc = dev.rank(np.full(n, ord("A"), dtype=np.uint8), np.arange(n, dtype=np.uint64), np.ones(n, dtype=np.uint8))
    assert bytes(acc) == bytes(acgt[np.array(want_bwt)])
    assert dev.selfcheck()["bad_sa_isa"] == 0
    dev.close()


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref (compiled reference) not present")
def test_native_writer_equals_reference_builder_on_a_fresh_3mbp_text(tmp_path):
    """centrifuger-build of the reference and the native writer on the same 3 Mbp input: .1.cfr fields and .2.cfr bytes equal."""
    g = synth.make_genomes(n_species=6, n_strains=4, genome_len=125000, seed=91)
    synth.write_reference_inputs(g, str(tmp_path))
    ref_prefix = str(tmp_path / "ref")
    subprocess.run([os.path.join(REF_DIR, "centrifuger-build"), "-t", "8", "-r", str(tmp_path / "ref.fa"), "--taxonomy-tree", str(tmp_path / "nodes.dmp"),
                    "--name-table", str(tmp_path / "names.dmp"), "--conversion-table", str(tmp_path / "seqid.map"), "-o", ref_prefix],
                   check=True, stderr=subprocess.DEVNULL)
    prefix = str(tmp_path / "own")
    capi.build_index(g.names, g.taxids, g.seqs, g.nodes, g.tax_names, prefix)
    # and through the drop-in command line (same files the reference's builder read, gz'd FASTA)
    cli_prefix = str(tmp_path / "cli")
    subprocess.run(["gzip", "-1", str(tmp_path / "ref.fa")], check=True)
    subprocess.run([os.path.join(ROOT, "centrifuger_amd", "bin", "centrifuger-build"), "-t", "8", "-r", str(tmp_path / "ref.fa.gz"), "--taxonomy-tree",
                    str(tmp_path / "nodes.dmp"), "--name-table", str(tmp_path / "names.dmp"), "--conversion-table", str(tmp_path / "seqid.map"),
                    "-o", cli_prefix, "--bmax", "1000000"], check=True, stderr=subprocess.DEVNULL)
    ref = parse_1cfr(ref_prefix + ".1.cfr")
    for pre in (prefix, cli_prefix):
        mine = parse_1cfr(pre + ".1.cfr")
        assert len(mine) == len(ref)
        for (na, va), (nb, vb) in zip(mine, ref):
            assert na == nb and va == vb, f"field {na} differs"
        assert open(pre + ".2.cfr", "rb").read() == open(ref_prefix + ".2.cfr", "rb").read()
        assert open(pre + ".3.cfr", "rb").read() == open(ref_prefix + ".3.cfr", "rb").read()


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref (compiled reference) not present")
@pytest.mark.parametrize("ignore_uncategorized", [False, True])
def test_cli_takes_the_inputs_the_reference_builder_takes(tmp_path, ignore_uncategorized):
    """Builder::Build (Builder.hpp:108-165): FASTA order != conversion-table order, a conversion table naming sequences the FASTA
    does not hold, a FASTA sequence missing from the table (extra name / skipped), a repeated id, a sequence that is too short
    after dropping non-ACGT characters, a name listed twice with different tax ids (LCA), a tax id outside the tree: the files of
    bin/centrifuger-build must equal the reference builder's, and so must a classification on them."""
    g = synth.make_genomes(n_species=4, n_strains=3, genome_len=30000, seed=17)
    synth.write_reference_inputs(g, str(tmp_path))
    rng = np.random.default_rng(5)
    order = rng.permutation(len(g.names))
    acgt = lambda k: "".join("ACGT"[x] for x in rng.integers(0, 4, size=k))
    with open(tmp_path / "messy.fa", "w") as f:
        def rec(name, s):
            f.write(f">{name} some description\n")
            for a in range(0, len(s), 70):
                f.write(s[a:a + 70] + "\n")
        for q, gi in enumerate(order):
            s = bytes(g.seqs[gi]).decode()
            if q == 2:
                rec("uncategorized_contig", acgt(5000))                   # not in the conversion table
            if q == 4:
                rec(g.names[order[0]], acgt(3000))                        # repeated id: ignored
                rec("short_one", "ACGTNNNNNNNNNNNNNNNNNNacgtacgtacgtACG")   # 7 symbols after compaction: filtered
            if q == 5:
                s = s[:10000] + "N" * 50 + s[10000:20000].lower() + s[20000:]   # dropped characters inside a genome
            rec(g.names[gi], s)
    lines = open(tmp_path / "seqid.map").read().splitlines()
    lines.insert(3, "absent_from_fasta_1\t%d" % g.taxids[0])
    lines.append("short_one\t%d" % g.taxids[1])
    lines.append("absent_from_fasta_2\t999999")                                 # tax id outside the tree
    lines.append("%s\t%d" % (g.names[order[1]], g.taxids[(order[1] + 1) % len(g.names)]))   # second tax id for a listed name
    open(tmp_path / "messy.map", "w").write("\n".join(lines) + "\n")
    common = ["-r", str(tmp_path / "messy.fa"), "--taxonomy-tree", str(tmp_path / "nodes.dmp"), "--name-table", str(tmp_path / "names.dmp"),
              "--conversion-table", str(tmp_path / "messy.map")] + (["--ignore-uncategorized-genome"] if ignore_uncategorized else [])
    ref_prefix, own_prefix = str(tmp_path / "ref"), str(tmp_path / "own")
    subprocess.run([os.path.join(REF_DIR, "centrifuger-build"), "-t", "4", "-o", ref_prefix] + common, check=True, stderr=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(ROOT, "centrifuger_amd", "bin", "centrifuger-build"), "-t", "4", "-o", own_prefix] + common, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    err = r.stderr.decode()
    assert "taxonomy id doesn't exist for uncategorized_contig" in err and "short_one is filtered" in err
    ref, mine = parse_1cfr(ref_prefix + ".1.cfr"), parse_1cfr(own_prefix + ".1.cfr")
    assert len(mine) == len(ref)
    for (na, va), (nb, vb) in zip(mine, ref):
        assert na == nb and va == vb, f"field {na} differs"
    for k in (2, 3):
        assert open(f"{own_prefix}.{k}.cfr", "rb").read() == open(f"{ref_prefix}.{k}.cfr", "rb").read(), f".{k}.cfr differs"
    # and the reference classifies identically on either index (reads from all genomes + the uncategorized contig)
    reads = synth.make_reads(g, 400, 120, seed=3)
    synth.write_fastq(reads, str(tmp_path / "r.fq"))
    out = []
    for pre in (ref_prefix, own_prefix):
        out.append(subprocess.run([os.path.join(REF_DIR, "centrifuger"), "-x", pre, "-u", str(tmp_path / "r.fq"), "-t", "2"], check=True,
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout)
    assert out[0] == out[1]


def test_build_entry_rejects_bad_genome_lists(tmp_path):
    """cfr_build_index validates before it touches the text: empty text, a genome shorter than --ftabchars + 1, an id twice."""
    g = synth.make_genomes(n_species=2, n_strains=1, genome_len=500, seed=3)
    for kw, lens in ((dict(), [0, 0]), (dict(), [500, 5]), (dict(genome_seq=[0, 0]), [500, 500])):
        text = np.concatenate([g.seqs[0][:lens[0]], g.seqs[1][:lens[1]]]) if sum(lens) else np.zeros(1, dtype=np.uint8)
        with pytest.raises(capi.CfrError):
            capi.build_index(g.names, g.taxids, (text, np.array(lens, dtype=np.uint64)), g.nodes, g.tax_names, str(tmp_path / "bad"), **kw)
