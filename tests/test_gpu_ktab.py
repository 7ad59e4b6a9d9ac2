"""The K-mer COUNT table (cfr_kernels.hip.inc: k_build_ktab, the KTAB instantiations of k_search_chains_v2; profiles/HISTORY.md section 10,
profiles/r5z_ktab_model.txt) is OFF unless CFR_KTAB=1: written in the last hours of round 5, it ran on a GPU for 70 seconds in all - these
69 tests passed for each of its three encodings and the byte form gained 10 % on the scaled model of 40 Gbp - and has not seen a 40 Gbp
index.  The tests: the table against the search core on a sample of keys (CFR_KTAB_CHECK=1 makes a disagreement an error at load; in round
5's runs that check read its counters too early, so the TSVs are what those runs prove), then the reference's TSVs through the kernels that use it, on the 36-bit image forced on the golden indexes, for several table widths.
Skipped unless CFR_TEST_KTAB=1 (CFR_TEST_KTAB=1 python -m pytest tests/test_gpu_ktab.py -m gpu) until the path is switched on."""
import os

import pytest

from centrifuger_amd import capi
from conftest import GOLDEN
from test_gpu_limits import env
from test_gpu_parity import MAN, _case_kw, _load_case_reads

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.environ.get("CFR_TEST_KTAB"), reason="the K-mer count table has not been validated on a GPU yet: CFR_TEST_KTAB=1 runs these")]
KTAB_ENV = dict(CFR_DEBUG_ENV=1, CFR_FORCE_WIDE=1, CFR_KTAB=1, CFR_KTAB_CHECK=1)


def _open(golden_dir, iname, extra=None, **kw):
    with env(**dict(KTAB_ENV, **(extra or {}))):
        idx = capi.Index(os.path.join(golden_dir, iname), capi.default_params(**kw))
        return idx, capi.DeviceIndex(idx)


@pytest.mark.parametrize("e8", [0, 1])
@pytest.mark.parametrize("case", sorted(c for c in MAN["cases"] if "--expand-taxid" not in MAN["cases"][c]["args"]))
def test_tsv_equals_reference_with_the_count_table(case, e8, golden_dir):
    c = MAN["cases"][case]
    idx, dev = _open(golden_dir, c["index"], {"CFR_FTABX_E8": e8}, **_case_kw(c["args"]))
    ids, b1, o1, b2, o2 = _load_case_reads(c["args"], golden_dir)
    if "--no-dust" not in c["args"]:
        capi.dust_mask(b1, o1)
        if b2 is not None:
            capi.dust_mask(b2, o2)
    with env(CFR_DEBUG_ENV=1, CFR_SUBBATCH=37, CFR_TAPER_FLOOR=0):
        results, matches = dev.classify(b1, o1, b2, o2)
    out = capi.tsv_header() + b"".join(idx.format_tsv(ids[i], results[i], matches) for i in range(len(ids)))
    assert out == open(os.path.join(GOLDEN, "tsv", case + ".tsv"), "rb").read()
    dev.close()


@pytest.mark.parametrize("width", [7, 9, 12])
def test_tsv_equals_reference_for_several_table_widths(width, golden_dir):
    """the count table is one character wider than the derived K-mer table: K = 7, 9, 12 -> K + 1 = 8, 10, 13 on a 60 kbp text covers tables
    in which nearly every entry occurs, about a quarter, and almost none"""
    case = "f6.se_nodust"
    c = MAN["cases"][case]
    idx, dev = _open(golden_dir, c["index"], {"CFR_FTABX_WIDTH": width, "CFR_FTABX_E8": 1}, **_case_kw(c["args"]))
    ids, b1, o1, b2, o2 = _load_case_reads(c["args"], golden_dir)
    results, matches = dev.classify(b1, o1, b2, o2)
    out = capi.tsv_header() + b"".join(idx.format_tsv(ids[i], results[i], matches) for i in range(len(ids)))
    assert out == open(os.path.join(GOLDEN, "tsv", case + ".tsv"), "rb").read()
    dev.close()
