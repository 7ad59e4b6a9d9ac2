"""`--expand-taxid` through the HIP path (Classifier.hpp:792-838, Taxonomy.hpp:733-973, ResultWriter.hpp:194-195, 226-227): the
library entry cfr_classify_batch_expanded and the command line against the TSVs the REAL reference wrote (tests/golden/expand),
under the switches that send a read through each form of the device tail, and against the C oracle on a many-strain workload
whose lists are long.  Bit-exact.  -m gpu."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ora
from centrifuger_amd import capi, synth
from conftest import ROOT
from test_host_expand_cpu import case_params, load_reads
from test_oracle_golden_expand import EXP, MAN, expand_args, expand_index

pytestmark = pytest.mark.gpu
CLI = os.path.join(ROOT, "centrifuger_amd", "bin", "centrifuger")


def open_with(prefix, env=None, **params):
    old = {}
    for key, val in (env or {}).items():
        old[key] = os.environ.get(key)
        os.environ[key] = val
    try:
        idx = capi.Index(prefix, capi.default_params(output_expanded=1, **params))
        dev = capi.DeviceIndex(idx)
    finally:
        for key, val in old.items():
            if val is None:
                os.environ.pop(key, None)
            else:
                os.environ[key] = val
    return idx, dev


def tsv_of(idx, dev, ids, b1, o1, b2, o2, **kw):
    results, matches, spans, xids = dev.classify_expanded(b1, o1, b2, o2, **kw)
    return capi.lib().cfr_tsv_header_expanded() + b"".join(idx.format_tsv_expanded(ids[i], results[i], matches, spans, xids) for i in range(len(ids)))


@pytest.mark.parametrize("case", sorted(c for c in MAN["cases"] if "--expand-taxid" in MAN["cases"][c]["args"]))
def test_library_entry_equals_reference_tsv(case, golden_dir):
    c = MAN["cases"][case]
    kw = case_params(c["args"])
    idx, dev = open_with(expand_index(case, golden_dir), max_result=kw.get("max_result", 1), max_result_per_hit_factor=kw.get("hitk_factor", 40))
    ids, b1, o1, b2, o2 = load_reads(c["args"], golden_dir)
    dev.set_dust("--no-dust" not in c["args"])
    want = open(os.path.join(EXP, "tsv", case + ".tsv"), "rb").read()
    assert tsv_of(idx, dev, ids, b1, o1, b2, o2) == want
    assert tsv_of(idx, dev, ids, b1, o1, b2, o2, ids_cap=3) == want          # CFR_ERR_CAPACITY names the size, the second call fits
    # the plain entry of the same image: same rows without the column
    r0, m0 = dev.classify(b1, o1, b2, o2)
    plain = capi.tsv_header() + b"".join(idx.format_tsv(ids[i], r0[i], m0) for i in range(len(ids)))
    assert plain.split(b"\n")[1:] == [ln.rsplit(b"\t", 1)[0] if ln else ln for ln in want.split(b"\n")[1:]]
    dev.close()


SWITCHES = {
    "two_kernel_post_stage": {"CFR_FUSED_POST": "0"},
    "single_lane_fold": {"CFR_TEAM_TAIL": "0"},
    "list_pool_grows": {"CFR_EXP_POOL_INIT": "8", "CFR_SUBBATCH": "37", "CFR_TAPER_FLOOR": "0"},
    "scratch_pool_redo": {"CFR_POOL_INIT": "3", "CFR_SUBBATCH": "50", "CFR_TAPER_FLOOR": "0"},
    "scratch_pool_pinned": {"CFR_POOL_CAP": "3", "CFR_SUBBATCH": "50", "CFR_TAPER_FLOOR": "0"},
    "no_derived_tables": {"CFR_FTABX_WIDTH": "0", "CFR_TEXT_MODE": "0", "CFR_LOC_MEMO_GB": "0"},
    "run_block_layout": {"CFR_LAYOUT": "rb"},
    "staged_inputs": {"CFR_STREAM_INPUTS": "0"},
}


@pytest.mark.parametrize("name", sorted(SWITCHES))
@pytest.mark.parametrize("case", ["x8.se_k1_expand", "x8.pe_k3_expand", "x8.se_k5_expand"])
def test_every_form_of_the_tail_keeps_the_lists(name, case, golden_dir):
    c = MAN["cases"][case]
    kw = case_params(c["args"])
    idx, dev = open_with(expand_index(case, golden_dir), SWITCHES[name], max_result=kw.get("max_result", 1))
    ids, b1, o1, b2, o2 = load_reads(c["args"], golden_dir)
    dev.set_dust(True)
    assert tsv_of(idx, dev, ids, b1, o1, b2, o2) == open(os.path.join(EXP, "tsv", case + ".tsv"), "rb").read()
    dev.close()


@pytest.mark.parametrize("case", sorted(c for c in MAN["cases"] if "--expand-taxid" in MAN["cases"][c]["args"]))
def test_cli_stdout_equals_reference_tsv(case, golden_dir):
    c = MAN["cases"][case]
    out = subprocess.run([CLI, "-x", expand_index(case, golden_dir), "-t", "3", "--gpu-batch", "97"] + expand_args(c["args"], golden_dir),
                         check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert out.stdout == open(os.path.join(EXP, "tsv", case + ".tsv"), "rb").read()
    assert b"can be classified." in out.stderr


def test_entry_refuses_an_index_opened_without_the_parameter(golden_dir):
    idx = capi.Index(os.path.join(golden_dir, "f6"), capi.default_params())
    dev = capi.DeviceIndex(idx)
    with pytest.raises(capi.CfrError) as e:
        dev.classify_expanded(np.frombuffer(b"ACGT" * 10, np.uint8), np.array([0, 40], np.uint64))
    assert e.value.status == capi.CFR_ERR_ARG
    dev.close()


@pytest.mark.parametrize("k", [1, 3])
def test_long_lists_of_a_many_strain_index_against_the_oracle(tmp_path, k):
    """40 near-identical strains per species: the best ids of a read run to dozens, the team folds hand such a read to the single-lane
    form (the lists need the ids as an array), and the list of a reported species holds up to 40 strains.  Every field and every
    list against the C oracle (which is pinned to the reference's TSVs by tests/test_oracle_golden_expand.py)."""
    g = synth.make_genomes(4, 40, 30_000, seed=2203, divergence_step=0.0004)
    prefix = str(tmp_path / "idx")
    capi.build_index(g.names, g.taxids, g.seqs, g.nodes, g.tax_names, prefix)
    n = 30_000
    rs = synth.make_reads(g, n, 150, seed=2204, sub_rate=0.004, n_rate=0.0005)
    r1, r2 = synth.make_pairs(g, 8_000, 125, seed=2205)
    idx, dev = open_with(prefix, {"CFR_EXP_POOL_INIT": "1000"}, max_result=k)
    o = ora.OracleIndex(prefix, max_result=k, expand=True)
    for (b1, o1, b2, o2, m) in ((rs.bases, rs.offsets, None, None, 2000), (r1.bases, r1.offsets, r2.bases, r2.offsets, 600)):
        results, matches, spans, xids = dev.classify_expanded(b1, o1, b2, o2)
        again = dev.classify_expanded(b1, o1, b2, o2)                 # the pool has grown by now: same answers
        assert all(a.tobytes() == b.tobytes() for a, b in zip((results, matches, spans, xids), again))
        ores = o.classify(b1[:int(o1[m])], o1[:m + 1], None if b2 is None else b2[:int(o2[m])], None if o2 is None else o2[:m + 1], threads=16)
        longest = 0
        for i in range(m):
            assert idx.format_tsv_expanded("r", results[i], matches, spans, xids) == o.format("r", ores[i]), i
            for q in range(int(results[i]["n_match"])):
                longest = max(longest, int(spans[int(results[i]["match_begin"]) + q]["count"]))
        assert longest >= (20 if k == 1 else 10), longest
    o.close()
    dev.close()
