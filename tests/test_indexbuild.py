"""centrifuger_amd/indexbuild.py (own .cfr writer: prefix-doubling suffix array, BWT, ftab, sampled SA,
run-block compression, taxonomy) against the indexes the REAL reference's centrifuger-build wrote for the
same genomes (tests/golden/*.cfr).  CPU only."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest
import torch

from centrifuger_amd import indexbuild, synth
from cfr_fields import parse_1cfr
from conftest import GOLDEN

MAN = json.load(open(os.path.join(GOLDEN, "manifest.json")))
VARIANTS = {"f6": dict(ftab_chars=6), "f6_b1": dict(ftab_chars=6, rbbwt_b=1), "f6_b8": dict(ftab_chars=6, rbbwt_b=8),
            "f6_off3": dict(ftab_chars=6, offrate=3), "f10": dict()}


@pytest.fixture(scope="module")
def genomes():
    return synth.make_genomes(n_species=5, n_strains=3, genome_len=20000, seed=MAN["seed"])


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_written_index_equals_reference_index_field_by_field(name, genomes, golden_dir, tmp_path):
    g = genomes
    prefix = str(tmp_path / name)
    indexbuild.build_index(g.names, g.taxids, g.seqs, g.nodes, g.tax_names, prefix, device=torch.device("cpu"), **VARIANTS[name])
    mine = parse_1cfr(prefix + ".1.cfr")
    ref = parse_1cfr(os.path.join(golden_dir, name + ".1.cfr"))
    assert len(mine) == len(ref)
    for (na, va), (nb, vb) in zip(mine, ref):
        assert na == nb
        assert va == vb, f"field {na} differs"
    assert open(prefix + ".2.cfr", "rb").read() == open(os.path.join(golden_dir, name + ".2.cfr"), "rb").read()


def test_classification_on_own_index_matches_reference_tsv(genomes, golden_dir, oracle_bin, tmp_path):
    g = genomes
    prefix = str(tmp_path / "own")
    indexbuild.build_index(g.names, g.taxids, g.seqs, g.nodes, g.tax_names, prefix, ftab_chars=6, device=torch.device("cpu"))
    for case in ("f6.se_default", "f6.pe_k5", "f6.edge_default"):
        c = MAN["cases"][case]
        args = [os.path.join(golden_dir, a) if a.endswith((".fq", ".fa")) else a for a in c["args"]]
        out = subprocess.run([oracle_bin, "classify", "-x", prefix] + args, check=True, stdout=subprocess.PIPE).stdout
        assert hashlib.md5(out).hexdigest() == c["md5"]


def test_suffix_array_against_naive():
    rng = np.random.default_rng(11)
    for n in (1, 2, 37, 500):
        t = rng.integers(0, 4, size=n, dtype=np.uint8)
        if n == 500:
            t[100:300] = t[300:500]          # a long repeat
        sa = indexbuild.suffix_array(torch.from_numpy(t)).numpy()
        s = bytes(t.tolist())
        want = sorted(range(n), key=lambda i: s[i:])
        assert list(sa) == want


def test_block_size_rule_on_repetitive_sequence():
    rng = np.random.default_rng(2)
    S = np.repeat(rng.integers(0, 4, size=3000, dtype=np.uint8), rng.integers(1, 30, size=3000))
    b = indexbuild.compute_block_size(S, len(S))
    assert b >= 2
    S2 = rng.integers(0, 4, size=50000, dtype=np.uint8)
    assert indexbuild.compute_block_size(S2, len(S2)) == 1      # no runs: "no compression"
