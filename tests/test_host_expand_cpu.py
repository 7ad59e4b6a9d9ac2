"""`--expand-taxid` on the host side of the library (no GPU): cfr_classify_from_hits_expanded - the host twin of the device tail
(csrc/cfr_tail.cpp) - fed with the oracle's hits and located ids must reproduce the reference's TSVs with the expandedTaxIDs
column (tests/golden/expand, Classifier.hpp:792-838, Taxonomy.hpp:733-973, ResultWriter.hpp:194-195, 226-227)."""
import os

import numpy as np
import pytest

import oracle_lib as ora
from centrifuger_amd import capi
from conftest import GOLDEN
from test_host_cpu import _rows_for_hit
from test_oracle_golden_expand import EXP, MAN, expand_args, expand_index


def case_params(args):
    kw = {}
    if "-k" in args: kw["max_result"] = int(args[args.index("-k") + 1])
    if "--hitk-factor" in args: kw["hitk_factor"] = int(args[args.index("--hitk-factor") + 1])
    return kw


def load_reads(args, gd):
    a = expand_args(args, gd)
    if "-u" in a:
        ids, b1, o1 = ora.read_fastx(a[a.index("-u") + 1]); return ids, b1, o1, None, None
    ids, b1, o1 = ora.read_fastx(a[a.index("-1") + 1])
    _, b2, o2 = ora.read_fastx(a[a.index("-2") + 1])
    return ids, b1, o1, b2, o2


@pytest.mark.parametrize("case", ["x8.se_k1_expand", "x8.se_k2_expand", "x8.se_k3_expand", "x8.pe_k5_expand", "x8.pe_k1_expand",
                                  "x8.se_k1_expand_nodust_hitk2", "f6.se_k1_expand", "f6.edge_pe_k2_expand"])
def test_host_tail_keeps_the_promoted_ids(case, golden_dir):
    c = MAN["cases"][case]
    args = c["args"]
    kw = case_params(args)
    prefix = expand_index(case, golden_dir)
    o = ora.OracleIndex(prefix, **kw)
    params = capi.default_params(max_result=kw.get("max_result", 1), max_result_per_hit_factor=kw.get("hitk_factor", 40), output_expanded=1)
    idx = capi.Index(prefix, params)
    mhl = idx.info().min_hit_len
    ids, b1, o1, b2, o2 = load_reads(args, golden_dir)
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
            row_vals.extend(o.locate(r)[0] for r in _rows_for_hit(h, max_entries % (1 << 64), locate_all, mhl))
            row_begin.append(len(row_vals))
        hit_begin.append(len(hits))
        qlen.append(len(r1) + (len(r2) if r2 is not None else 0))
    results, matches, spans, xids = idx.classify_from_hits_expanded(np.array(hits, dtype=capi.HIT_DTYPE), hit_begin, row_begin,
                                                                     np.array(row_vals, dtype=np.uint64), qlen, threads=3)
    out = capi.lib().cfr_tsv_header_expanded() + b"".join(idx.format_tsv_expanded(ids[i], results[i], matches, spans, xids) for i in range(len(ids)))
    assert out == open(os.path.join(EXP, "tsv", case + ".tsv"), "rb").read()
    # the plain entry on the same index leaves the lists out and changes nothing else
    r0, m0 = idx.classify_from_hits(np.array(hits, dtype=capi.HIT_DTYPE), hit_begin, row_begin, np.array(row_vals, dtype=np.uint64), qlen)
    assert r0.tobytes() == results.tobytes() and m0.tobytes() == matches.tobytes()


def test_expanded_entry_needs_the_parameter(golden_dir):
    idx = capi.Index(os.path.join(golden_dir, "f6"), capi.default_params())
    with pytest.raises(capi.CfrError) as e:
        idx.classify_from_hits_expanded(np.zeros(0, dtype=capi.HIT_DTYPE), [0], [0], np.zeros(0, dtype=np.uint64), [])
    assert e.value.status == capi.CFR_ERR_ARG and b"output_expanded" in str(e.value).encode()
