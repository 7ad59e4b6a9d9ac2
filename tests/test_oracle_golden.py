"""The plain-C restatement (oracle/) against the golden vectors the REAL reference produced
(tests/golden/make_golden.py).  CPU only."""
import gzip
import hashlib
import json
import os
import subprocess

import pytest

from conftest import GOLDEN

MAN = json.load(open(os.path.join(GOLDEN, "manifest.json")))


def _args(args, gd):
    return [os.path.join(gd, a) if a.endswith((".fq", ".fa")) else a for a in args]


@pytest.mark.parametrize("case", sorted(MAN["cases"]))
def test_tsv_matches_reference(case, oracle_bin, golden_dir):
    c = MAN["cases"][case]
    out = subprocess.run([oracle_bin, "classify", "-x", os.path.join(golden_dir, c["index"])] + _args(c["args"], golden_dir),
                         check=True, stdout=subprocess.PIPE).stdout
    want = open(os.path.join(GOLDEN, "tsv", case + ".tsv"), "rb").read()
    assert hashlib.md5(want).hexdigest() == c["md5"]
    assert out == want


@pytest.mark.parametrize("threads", [2, 5])
def test_threads_do_not_change_output(threads, oracle_bin, golden_dir):
    c = MAN["cases"]["f6.pe_k5"]
    out = subprocess.run([oracle_bin, "classify", "-x", os.path.join(golden_dir, "f6"), "-t", str(threads)] + _args(c["args"], golden_dir),
                         check=True, stdout=subprocess.PIPE).stdout
    assert hashlib.md5(out).hexdigest() == c["md5"]


@pytest.mark.parametrize("vec", sorted(MAN["vectors"]))
def test_intermediate_vectors(vec, oracle_bin, golden_dir):
    v = MAN["vectors"][vec]
    iname, kind = vec.split(".")
    idx = os.path.join(golden_dir, iname)
    if kind == "rank":
        cmd = [oracle_bin, "dump-rank", "-x", idx, "--step", v["arg"]]
    elif kind == "locate":
        cmd = [oracle_bin, "dump-locate", "-x", idx, "--step", v["arg"]]
    else:
        cmd = [oracle_bin, "dump-bs", "-x", idx, "-u", os.path.join(golden_dir, v["arg"])]
    out = subprocess.run(cmd, check=True, stdout=subprocess.PIPE).stdout
    assert out.count(b"\n") == v["lines"]
    assert hashlib.md5(out).hexdigest() == v["md5"]
    if "file" in v:
        assert gzip.open(os.path.join(GOLDEN, v["file"]), "rb").read() == out
