"""n >= 2^32: a 4.4 Gbp index written by the native writer on this box (220 species x 5 strains x 4 Mbp), 300 k reads.
The device image must use the 40-bit tables (5-byte SA / ISA entries, WIDE search kernel), its derived tables must be
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
