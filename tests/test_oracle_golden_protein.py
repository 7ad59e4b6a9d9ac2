"""The plain-C restatement of the PROTEIN path (Sequence_RunBlockOneTree rank/access, the end-marker rows of
GetSampledSA, TranslatedSearch with its frame-score quirk) against what the REAL reference produced for the protein
fixtures (tests/golden/make_golden_protein.py -> tests/golden/prot).  CPU only."""
import gzip
import hashlib
import json
import os
import shutil
import subprocess

import pytest

from conftest import GOLDEN

PROT = os.path.join(GOLDEN, "prot")
MAN = json.load(open(os.path.join(PROT, "manifest.json")))


@pytest.fixture(scope="module")
def prot_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("prot")
    for f in os.listdir(PROT):
        src = os.path.join(PROT, f)
        if f.endswith(".cfr.gz"):
            with gzip.open(src, "rb") as fi, open(d / f[:-3], "wb") as fo:
                shutil.copyfileobj(fi, fo)
        elif os.path.isfile(src):
            os.symlink(src, d / f)
    return str(d)


@pytest.mark.parametrize("case", sorted(MAN["cases"]))
def test_protein_tsv_matches_reference(case, oracle_bin, prot_dir):
    c = MAN["cases"][case]
    args = [os.path.join(prot_dir, a) if a.endswith(".fa") else a for a in c["args"]]
    out = subprocess.run([oracle_bin, "classify", "-x", os.path.join(prot_dir, c["index"])] + args, check=True, stdout=subprocess.PIPE).stdout
    want = open(os.path.join(PROT, "tsv", case + ".tsv"), "rb").read()
    assert hashlib.md5(want).hexdigest() == c["md5"]
    assert out == want
    assert out.count(b"\n") > 10


@pytest.mark.parametrize("vec", sorted(MAN["vectors"]))
def test_protein_rank_access_locate_streams(vec, oracle_bin, prot_dir):
    v = MAN["vectors"][vec]
    iname, kind = vec.split(".")
    cmd = [oracle_bin, "dump-rank" if kind == "prank" else "dump-locate", "-x", os.path.join(prot_dir, iname), "--step", v["arg"]]
    out = subprocess.run(cmd, check=True, stdout=subprocess.PIPE).stdout
    assert out.count(b"\n") == v["lines"]
    assert hashlib.md5(out).hexdigest() == v["md5"]
    if "file" in v:
        assert gzip.open(os.path.join(PROT, v["file"]), "rb").read() == out
