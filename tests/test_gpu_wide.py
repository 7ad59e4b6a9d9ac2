"""n >= 2^32: a 4.4 Gbp index written by the native writer on this box (220 species x 5 strains x 4 Mbp), 300 k reads.
The device image must use the 36-bit packed suffix array (WIDE search kernel), its derived tables must be
consistent with the BWT on every one of the 4.4 G rows, and the TSV must be byte-identical to what the REFERENCE binary
prints from the same index files and reads (dust on, default options).  -m gpu, about a minute."""
import os
import subprocess

import numpy as np
import pytest

from centrifuger_amd import capi, synth
from conftest import REF_DIR, have_ref

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref (compiled reference) not present")
def test_index_above_2_pow_32_against_the_reference_binary(tmp_path):
    g, cat = synth.make_genomes_fast(220, 5, 4_000_000, seed=4242, threads=min(os.cpu_count() or 1, 64))
    assert len(cat) > (1 << 32)
    prefix = str(tmp_path / "big")
    lens = np.array([len(s) for s in g.seqs], dtype=np.uint64)
    rep = capi.build_index(g.names, g.taxids, (cat, lens), g.nodes, g.tax_names, prefix)
    assert rep["n"] == len(cat)
    n = 300_000
    rs = synth.make_reads(g, n, 150, seed=4243, cat=cat)
    del cat
    synth.write_fasta(rs, str(tmp_path / "r.fa"))
    want = subprocess.run([os.path.join(REF_DIR, "centrifuger"), "-x", prefix, "-t", str(min(os.cpu_count() or 1, 64)), "-u", str(tmp_path / "r.fa")],
                          check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
    idx = capi.Index(prefix, capi.default_params(max_result=1))
    dev = capi.DeviceIndex(idx)
    assert dev.info().n == rep["n"] and dev.info().n >= (1 << 32)
    chk = dev.selfcheck()
    assert chk["text_tables"] and chk["memo"] == 1, chk
    assert (chk["bad_sa_isa"], chk["bad_text"], chk["bad_lf"], chk["bad_memo"]) == (0, 0, 0, 0), chk
    dev.set_dust(True)
    res, mat = dev.classify(rs.bases.copy(), rs.offsets)
    got = capi.tsv_header() + b"".join(idx.format_tsv(f"r{i}", res[i], mat) for i in range(n))
    assert got == want
    assert (res["n_match"] > 0).mean() > 0.99
    # the same reads in pairs (mate 2 = the next read): the paired WIDE kernel against the reference's -1/-2 run
    half = n // 2
    o = rs.offsets[:half + 1]
    b1, b2 = rs.bases[:int(o[-1])].copy(), rs.bases[int(o[-1]):2 * int(o[-1])].copy()
    synth.write_fasta(synth.ReadSet(b1, o.copy()), str(tmp_path / "m1.fa"))
    synth.write_fasta(synth.ReadSet(b2, o.copy()), str(tmp_path / "m2.fa"))
    want2 = subprocess.run([os.path.join(REF_DIR, "centrifuger"), "-x", prefix, "-t", str(min(os.cpu_count() or 1, 64)), "-k", "3",
                            "-1", str(tmp_path / "m1.fa"), "-2", str(tmp_path / "m2.fa")], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
    dev.close()
    idx3 = capi.Index(prefix, capi.default_params(max_result=3))
    dev3 = capi.DeviceIndex(idx3)
    dev3.set_dust(True)
    res2, mat2 = dev3.classify(b1, o, b2, o)
    got2 = capi.tsv_header() + b"".join(idx3.format_tsv(f"r{i}", res2[i], mat2) for i in range(half))
    assert got2 == want2
    dev3.close()


@pytest.mark.skipif(os.environ.get("CFR_SKIP_BIG_TESTS") == "1" or not have_ref(), reason="CFR_SKIP_BIG_TESTS=1, or oracle/_ref not present")
def test_16gbp_host_sa_build_and_lean_image_against_the_reference(tmp_path_factory):
    """The forms cfg4 / cfg5 run in, at a size a test can afford (set CFR_SKIP_BIG_TESTS=1 on a small box: needs ~120 GB of host
    memory, 60 GB of /tmp and ~3 minutes): a 16 Gbp index written with the suffix array kept in HOST memory (the builder's form above
    ~27 Gbp), loaded as the LEAN image (36-bit packed suffix array, 8-byte K-mer entries, no locate memo: what a 40 Gbp index gets
    on a 288 GB device) - every derived table checked against the BWT, and SE / PE reads classified like the reference binary."""
    import shutil
    import numpy as np
    from centrifuger_amd import capi, synth
    d = str(tmp_path_factory.mktemp("big16"))
    free_gb = shutil.disk_usage(d).free / 1e9
    if free_gb < 70:
        pytest.skip(f"only {free_gb:.0f} GB free under {d}")
    n_species = 800                                     # x 5 strains x 4 Mbp = 16 Gbp
    g, cat = synth.make_genomes_fast(n_species, 5, 4_000_000, seed=1601, threads=min(os.cpu_count() or 1, 64))
    lens = np.array([len(s) for s in g.seqs], dtype=np.uint64)
    g.seqs = [None] * len(g.seqs)
    prefix = os.path.join(d, "idx")
    os.environ["CFR_BUILD_HOST_SA"] = "1"
    try:
        rep = capi.build_index(g.names, g.taxids, (cat, lens), g.nodes, g.tax_names, prefix)
    finally:
        os.environ.pop("CFR_BUILD_HOST_SA", None)
    assert rep["n"] == int(lens.sum()) >= 1 << 33
    starts = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    rng = np.random.default_rng(3)
    nr = 200_000

    def draw(n, L):
        gi = rng.integers(0, len(lens), size=n)
        pos = (rng.random(n) * (lens[gi].astype(np.int64) - 600)).astype(np.int64)
        return starts[gi] + pos
    at = draw(nr, 150)
    ar = np.arange(150)
    reads = np.ascontiguousarray(cat[at[:, None] + ar[None, :]])
    mut = rng.random(reads.shape) < 0.01
    reads[mut] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(mut.sum()))]
    rc = rng.random(nr) < 0.5
    reads[rc] = synth.revcomp(reads[rc])
    m2 = synth.revcomp(np.ascontiguousarray(cat[(at + 300)[:, None] + ar[None, :]]))
    del cat
    offs = np.arange(nr + 1, dtype=np.uint64) * np.uint64(150)
    rs1, rs2 = synth.ReadSet(reads.reshape(-1), offs), synth.ReadSet(np.ascontiguousarray(m2).reshape(-1), offs.copy())
    synth.write_fasta(rs1, os.path.join(d, "r1.fa"))
    synth.write_fasta(rs2, os.path.join(d, "r2.fa"))
    env_keys = {"CFR_DEBUG_ENV": "1", "CFR_FTABX_E8": "1", "CFR_LOC_MEMO_GB": "0"}
    old = {k: os.environ.get(k) for k in env_keys}
    os.environ.update(env_keys)
    try:
        for k, files in ((1, ["-u", os.path.join(d, "r1.fa")]), (5, ["-1", os.path.join(d, "r1.fa"), "-2", os.path.join(d, "r2.fa")])):
            idx = capi.Index(prefix, capi.default_params(max_result=k))
            dev = capi.DeviceIndex(idx)
            if k == 1:
                chk = dev.selfcheck()
                assert chk["text_tables"] and chk["memo"] == 0 and (chk["bad_sa_isa"], chk["bad_text"], chk["bad_lf"], chk["bad_memo"]) == (0, 0, 0, 0), chk
            b1, b2 = rs1.bases.copy(), rs2.bases.copy()
            capi.dust_mask(b1, offs)
            capi.dust_mask(b2, offs)
            res, mat = dev.classify(b1, offs, b2 if k == 5 else None, offs if k == 5 else None)
            own = capi.tsv_header() + b"".join(idx.format_tsv(f"r{i}", res[i], mat) for i in range(nr))
            dev.close()
            want = subprocess.run([os.path.join(REF_DIR, "centrifuger"), "-x", prefix, "-t", str(min(os.cpu_count() or 1, 64)), "-k", str(k)] + files,
                                  check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
            assert own == want, f"-k {k}"
    finally:
        for k_, v in old.items():
            if v is None:
                os.environ.pop(k_, None)
            else:
                os.environ[k_] = v
