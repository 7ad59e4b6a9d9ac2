"""The K-mer COUNT table (cfr_kernels.hip.inc: k_build_ktab, the KTAB instantiations of k_search_chains_v2; profiles/HISTORY.md section 10,
profiles/r5z_ktab_model.txt).  Since round 6 it is the default of 36-bit images when it fits (cfr_device.hip: mode 1 keeps the K-mer table it
is built from, mode 2 frees it after the build - the 40 Gbp image, where both do not fit beside the batches).  The tests: the table against
the search core on a sample of keys (CFR_KTAB_CHECK=1 makes a disagreement an error at load), then the reference's TSVs through the kernels
that use it, on the 36-bit image forced on the golden indexes, with the K-mer table kept and freed, for both entry sizes and several widths."""
import os

import pytest

from centrifuger_amd import capi
from conftest import GOLDEN
from test_gpu_limits import env
from test_gpu_parity import MAN, _case_kw, _load_case_reads

pytestmark = pytest.mark.gpu
KTAB_ENV = dict(CFR_DEBUG_ENV=1, CFR_FORCE_WIDE=1, CFR_KTAB=1, CFR_KTAB_CHECK=1)


def _open(golden_dir, iname, extra=None, **kw):
    with env(**dict(KTAB_ENV, **(extra or {}))):
        idx = capi.Index(os.path.join(golden_dir, iname), capi.default_params(**kw))
        return idx, capi.DeviceIndex(idx)


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("e8", [0, 1])
@pytest.mark.parametrize("case", sorted(c for c in MAN["cases"] if "--expand-taxid" not in MAN["cases"][c]["args"]))
def test_tsv_equals_reference_with_the_count_table(case, e8, mode, golden_dir):
    c = MAN["cases"][case]
    idx, dev = _open(golden_dir, c["index"], {"CFR_FTABX_E8": e8, "CFR_KTAB": mode}, **_case_kw(c["args"]))
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


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("width", [7, 9, 12])
def test_tsv_equals_reference_for_several_table_widths(width, mode, golden_dir):
    """the count table is one character wider than the derived K-mer table: K = 7, 9, 12 -> K + 1 = 8, 10, 13 on a 60 kbp text covers tables
    in which nearly every entry occurs, about a quarter, and almost none"""
    case = "f6.se_nodust"
    c = MAN["cases"][case]
    idx, dev = _open(golden_dir, c["index"], {"CFR_FTABX_WIDTH": width, "CFR_FTABX_E8": 1, "CFR_KTAB": mode}, **_case_kw(c["args"]))
    ids, b1, o1, b2, o2 = _load_case_reads(c["args"], golden_dir)
    results, matches = dev.classify(b1, o1, b2, o2)
    out = capi.tsv_header() + b"".join(idx.format_tsv(ids[i], results[i], matches) for i in range(len(ids)))
    assert out == open(os.path.join(GOLDEN, "tsv", case + ".tsv"), "rb").read()
    dev.close()


def test_the_count_table_is_the_default_of_a_36_bit_image(golden_dir):
    """no CFR_KTAB: a 36-bit image builds the table when it fits (here it does: mode 1) - its bytes show in the image's size - and CFR_KTAB=0
    leaves it out; CFR_KTAB=2 frees the K-mer table instead"""
    sizes = {}
    base = {k: v for k, v in KTAB_ENV.items() if k not in ("CFR_KTAB", "CFR_KTAB_CHECK")}
    old = os.environ.pop("CFR_KTAB", None)
    try:
        for name, extra in (("default", {}), ("off", {"CFR_KTAB": 0}), ("dropped", {"CFR_KTAB": 2})):
            with env(**dict(base, CFR_FTABX_WIDTH=9, CFR_FTABX_E8=1, **extra)):
                idx = capi.Index(os.path.join(golden_dir, "f6"), capi.default_params())
                dev = capi.DeviceIndex(idx)
                sizes[name] = dev.info().device_bytes
                dev.close()
    finally:
        if old is not None:
            os.environ["CFR_KTAB"] = old
    lines = (4 ** 10 + 47) // 48
    assert sizes["default"] - sizes["off"] == lines * 64, sizes
    assert sizes["default"] - sizes["dropped"] == 8 * 4 ** 9 + 16, sizes
