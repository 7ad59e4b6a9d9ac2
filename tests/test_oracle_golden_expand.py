"""`--expand-taxid` (Classifier.hpp:792-838, Taxonomy.hpp:733-973): the plain-C restatement against the TSVs the REAL reference
wrote (tests/golden/make_golden_expand.py).  CPU only."""
import hashlib
import json
import os
import subprocess

import pytest

from conftest import GOLDEN

EXP = os.path.join(GOLDEN, "expand")
MAN = json.load(open(os.path.join(EXP, "manifest.json")))


def expand_args(args, gd):
    return [os.path.join(GOLDEN if a.startswith("expand/") else gd, a) if a.endswith((".fq", ".fa")) else a for a in args]


def expand_index(case, gd):
    iname = MAN["cases"][case]["index"]
    return os.path.join(EXP if iname == "x8" else gd, iname)


@pytest.mark.parametrize("case", sorted(MAN["cases"]))
def test_expanded_tsv_matches_reference(case, oracle_bin, golden_dir):
    c = MAN["cases"][case]
    out = subprocess.run([oracle_bin, "classify", "-x", expand_index(case, golden_dir)] + expand_args(c["args"], golden_dir),
                         check=True, stdout=subprocess.PIPE).stdout
    want = open(os.path.join(EXP, "tsv", case + ".tsv"), "rb").read()
    assert hashlib.md5(want).hexdigest() == c["md5"]
    assert out == want


def test_the_fixtures_walk_every_branch():
    """what the golden set is for: lists at -k 1 (LCA children) and at -k > 1 (level sets), rows whose list stays empty, and the
    out-of-tree id (a FASTA sequence without a tax id) that hands back every input id"""
    rows = {c: v["rows_with_expanded_ids"] for c, v in MAN["cases"].items()}
    assert rows["x8.se_k1_expand"] > 100 and rows["x8.se_k3_expand"] > 100 and rows["x8.pe_k5_expand"] > 0 and rows["x8.se_k3"] == 0
    body = open(os.path.join(EXP, "tsv", "x8.se_k1_expand.tsv"), "rb").read().split(b"\n")[1:-1]
    lists = [ln.split(b"\t")[8] for ln in body]
    assert any(b"," in x for x in lists) and any(x == b"" for x in lists)
    # an id listed twice: only the out-of-tree branch copies its input as it is (Taxonomy.hpp:866-872)
    assert any(len(x.split(b",")) != len(set(x.split(b","))) for x in lists if x)
