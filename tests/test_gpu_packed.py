"""cfr_classify_batch_packed (the bases as 2-bit blocks + validity bits, half the bytes over PCIe) against cfr_classify_batch on the same
reads: golden read sets with N / lower case / short / empty reads, single-end and pairs, SDUST on the device on and off, one piece and
many (sub-batches that share a packed block); a protein index is refused.  -m gpu."""
import os

import numpy as np
import pytest

import oracle_lib as ora
from centrifuger_amd import capi
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _same(a, b, k):
    ra, ma = a
    rb, mb = b
    for f in ("score", "secondary_score", "hit_length", "query_length", "n_match"):
        assert np.array_equal(ra[f], rb[f]), f
    for i in np.nonzero(ra["n_match"] > 0)[0]:
        x = ma[int(ra[i]["match_begin"]):int(ra[i]["match_begin"]) + int(ra[i]["n_match"])]
        y = mb[int(rb[i]["match_begin"]):int(rb[i]["match_begin"]) + int(rb[i]["n_match"])]
        assert x.tobytes() == y.tobytes(), i


@pytest.mark.parametrize("dust", [False, True])
@pytest.mark.parametrize("k", [1, 5])
def test_packed_entry_equals_ascii_entry(golden_dir, dust, k):
    idx = capi.Index(os.path.join(golden_dir, "f10"), capi.default_params(max_result=k))
    dev = capi.DeviceIndex(idx)
    dev.set_dust(dust)
    for name in ("se.fq", "edge.fa", "long.fq"):
        _, b, o = ora.read_fastx(os.path.join(GOLDEN, name))
        _same(dev.classify(b.copy(), o), dev.classify_packed(capi.pack_reads(b, threads=3), o), k)
    _, b1, o1 = ora.read_fastx(os.path.join(GOLDEN, "pe_1.fq"))
    _, b2, o2 = ora.read_fastx(os.path.join(GOLDEN, "pe_2.fq"))
    _same(dev.classify(b1.copy(), o1, b2.copy(), o2), dev.classify_packed(capi.pack_reads(b1), o1, capi.pack_reads(b2), o2), k)
    _, e1, eo1 = ora.read_fastx(os.path.join(GOLDEN, "edge.fa"))
    _, e2, eo2 = ora.read_fastx(os.path.join(GOLDEN, "edge_2.fa"))
    _same(dev.classify(e1.copy(), eo1, e2.copy(), eo2), dev.classify_packed(capi.pack_reads(e1), eo1, capi.pack_reads(e2), eo2), k)
    dev.close()


@pytest.mark.parametrize("dust", [False, True])
def test_packed_entry_in_many_pieces(golden_dir, dust):
    """sub-batches of 37 reads: consecutive pieces share the packed block their boundary falls into"""
    idx = capi.Index(os.path.join(golden_dir, "f10"), capi.default_params(max_result=3))
    dev = capi.DeviceIndex(idx, 0, capi.default_device_options(sub_batch=37))
    dev.set_dust(dust)
    _, b, o = ora.read_fastx(os.path.join(GOLDEN, "se.fq"))
    rng = np.random.default_rng(5)
    b = b.copy()
    b[rng.random(len(b)) < 0.01] = ord("N")
    ref = capi.DeviceIndex(idx)
    ref.set_dust(dust)
    want = ref.classify(b.copy(), o)
    _same(want, dev.classify_packed(capi.pack_reads(b), o), 3)
    _same(want, dev.classify(b.copy(), o), 3)
    dev.close()
    ref.close()


def test_packed_entry_refuses_a_protein_index():
    """DnaToAa tells non-symbols apart (lower case goes down the ladder's last branch, N gives '?', Classifier.hpp:131-241): the packed
    form cannot carry that, so the entry says so instead of classifying something else"""
    prefix = os.path.join(GOLDEN, "prot", "p3_b4")
    idx = capi.Index(prefix, capi.default_params(max_result=3))
    dev = capi.DeviceIndex(idx)
    _, b, o = ora.read_fastx(os.path.join(GOLDEN, "prot", "se.fa"))
    with pytest.raises(capi.CfrError) as e:
        dev.classify_packed(capi.pack_reads(b), o)
    assert e.value.status == capi.CFR_ERR_ARG
    dev.close()
