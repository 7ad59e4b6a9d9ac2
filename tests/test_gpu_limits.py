"""Limits are errors or slower paths, never wrong answers: every branch that refuses, or that drops a derived table
because of a size, is walked on the GPU box - on small inputs, through test hooks (behind CFR_DEBUG_ENV) that lower the size the branch
looks at.  -m gpu."""
import os
from contextlib import contextmanager

import numpy as np
import pytest

from centrifuger_amd import capi, synth
from conftest import GOLDEN
from test_gpu_parity import MAN, _case_kw, _load_case_reads

pytestmark = pytest.mark.gpu


@contextmanager
def env(**kw):
    old = {k: os.environ.get(k) for k in kw}
    os.environ.update({k: str(v) for k, v in kw.items()})
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _build(g, prefix, **kw):
    text = np.concatenate(g.seqs)
    lens = np.array([len(s) for s in g.seqs], dtype=np.uint64)
    return capi.build_index(g.names, g.taxids, (text, lens), g.nodes, g.tax_names, prefix, **kw)


def test_max_result_above_4096_is_refused_by_name(golden_dir):
    """-k has no cap in the reference (Classifier.hpp:17-38); here the match buffers bound it: 4097 is an error that says so, 4096 loads"""
    idx = capi.Index(os.path.join(golden_dir, "f6"), capi.default_params(max_result=4097))
    with pytest.raises(capi.CfrError, match="4096"):
        capi.DeviceIndex(idx)
    dev = capi.DeviceIndex(capi.Index(os.path.join(golden_dir, "f6"), capi.default_params(max_result=4096)))
    dev.close()


def test_writer_refuses_a_text_beyond_its_size(tmp_path):
    """cfr_build_sa.hip: texts of 2^36 symbols and more are beyond the single-GPU writer - the same comparison at 2^12 (test hook)"""
    g = synth.make_genomes(n_species=2, n_strains=1, genome_len=3000, seed=11)
    with env(CFR_DEBUG_ENV=1, CFR_BUILD_LIMIT_LOG2=12):
        with pytest.raises(capi.CfrError, match="beyond this single-GPU writer"):
            _build(g, str(tmp_path / "big"))
    with env(CFR_DEBUG_ENV=1, CFR_BUILD_LIMIT_LOG2=14):
        _build(g, str(tmp_path / "fits"))
    assert os.path.exists(str(tmp_path / "fits") + ".1.cfr")


def test_writer_refuses_a_group_of_equal_prefixes_larger_than_a_chunk(tmp_path):
    """a run of one character longer than the sorting chunk is more suffixes under one prefix than a chunk holds: refused by name (the chunk
    is 2^20 .. 2^28 rows by default, 2^12 under the test hook); the same text is written when the chunk holds the run"""
    g = synth.make_genomes(n_species=2, n_strains=1, genome_len=8000, seed=12)
    g.seqs[0][1000:7000] = ord("A")
    with env(CFR_DEBUG_ENV=1, CFR_BUILD_CHUNK_LOG2=12):
        with pytest.raises(capi.CfrError, match="too skewed for this writer|too repetitive for this writer"):      # (the first pass meets the run first)
            _build(g, str(tmp_path / "rep"))
    _build(g, str(tmp_path / "ok"))
    idx = capi.Index(str(tmp_path / "ok"))
    assert idx.info().n >= 16000
    dev = capi.DeviceIndex(idx)
    sc = dev.selfcheck()
    assert sc["text_tables"] and not any(sc[k] for k in sc if k.startswith("bad_")), sc
    dev.close()


def test_protein_writer_refuses_a_first_table_wider_than_six_characters(tmp_path):
    """5 bits per character: --ftabchars 7 of a protein index would need a 2 x 32^7-entry table (cfr_build.cpp)"""
    rng = np.random.default_rng(5)
    aa = np.frombuffer(b"ARNDCEQGHILKMFPSTWYV", dtype=np.uint8)
    seqs = [aa[rng.integers(0, 20, 400)] for _ in range(3)]
    names = [f"p{i}" for i in range(3)]
    nodes = [(1, 1, "no rank"), (10, 1, "species"), (11, 1, "species"), (12, 1, "species")]
    text = np.concatenate(seqs)
    lens = np.array([400, 400, 400], dtype=np.uint64)
    with pytest.raises(capi.CfrError, match="1..6"):
        capi.build_index(names, [10, 11, 12], (text, lens), nodes, [(1, "root"), (10, "a"), (11, "b"), (12, "c")], str(tmp_path / "p7"), ftab_chars=7, protein=True)


@pytest.mark.parametrize("case", ["f6.se_nodust", "f6.pe_default", "f10.edge_pe_k3"])
def test_image_past_the_row_limit_of_text_mode_loads_without_it_and_answers_the_same(case, golden_dir):
    """36-bit suffix-array entries end at 2^36 rows: from there the image is built WITHOUT the text-mode tables (searches stay on the BWT) -
    the comparison lowered to 2^10 rows (test hook): a smaller image, the reference's TSV"""
    c = MAN["cases"][case]
    kw = _case_kw(c["args"])
    full = capi.DeviceIndex(capi.Index(os.path.join(golden_dir, c["index"]), capi.default_params(**kw)))
    with env(CFR_DEBUG_ENV=1, CFR_TEXT_LIMIT_LOG2=10):
        idx = capi.Index(os.path.join(golden_dir, c["index"]), capi.default_params(**kw))
        lean = capi.DeviceIndex(idx)
    assert idx.info().n >= 1 << 10
    assert lean.info().device_bytes + 4 * idx.info().n <= full.info().device_bytes          # no suffix array (4 bytes per row) at least
    sc = lean.selfcheck()
    assert not sc["text_tables"] and full.selfcheck()["text_tables"] and not any(sc[k] for k in sc if k.startswith("bad_")), sc
    ids, b1, o1, b2, o2 = _load_case_reads(c["args"], golden_dir)
    if "--no-dust" not in c["args"]:
        capi.dust_mask(b1, o1)
        if b2 is not None:
            capi.dust_mask(b2, o2)
    results, matches = lean.classify(b1, o1, b2, o2)
    out = capi.tsv_header() + b"".join(idx.format_tsv(ids[i], results[i], matches) for i in range(len(ids)))
    assert out == open(os.path.join(GOLDEN, "tsv", case + ".tsv"), "rb").read()
    lean.close()
    full.close()
