"""The HOST half of the protein index writer (csrc/cfr_build.cpp: Sequence_RunBlockOneTree compression, 5-bit wavelet tree, endMarkerSA,
ftab, the four files) without a GPU: tools/dbg/prot_writer_host.cpp compiles cfr_build.cpp with g++ and puts a naive suffix sort in the
place of the device one (csrc/cfr_build_sa.hip build_sa_bytes).  Its files must equal, field by field, the ones the REAL reference's
`centrifuger-build --protein` wrote for the same inputs (tests/golden/prot).  The device half is covered by tests/test_gpu_build_protein.py."""
import gzip
import os
import shutil
import subprocess

import pytest

from cfr_fields import parse_1cfr
from conftest import GOLDEN, ROOT

PROT = os.path.join(GOLDEN, "prot")
INPUT = os.path.join(PROT, "input")
# name -> ftabchars, offrate, rbbwt_b
VARIANTS = {"p2": (2, 4, 0), "p3_b4": (3, 4, 4), "p2_b1_off2": (2, 2, 1), "p4": (4, 4, 0)}


@pytest.fixture(scope="module")
def writer(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("protw") / "prot_writer_host")
    csrc = os.path.join(ROOT, "centrifuger_amd", "csrc")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", csrc, "-o", exe, os.path.join(ROOT, "tools", "dbg", "prot_writer_host.cpp"),
                    os.path.join(csrc, "cfr_build.cpp"), "-lpthread"], check=True)
    return exe


@pytest.mark.parametrize("threads", [0, 5])
@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_host_half_writes_the_reference_builders_files(name, threads, writer, tmp_path):
    ftab, offrate, b = VARIANTS[name]
    prefix = str(tmp_path / name)
    subprocess.run([writer, os.path.join(INPUT, "prot.fa"), os.path.join(INPUT, "nodes.dmp"), os.path.join(INPUT, "names.dmp"), os.path.join(INPUT, "seqid.map"),
                    prefix, str(ftab), str(offrate), str(b), str(threads)], check=True, stderr=subprocess.DEVNULL)
    gold = os.path.join(PROT, name + ".1.cfr")
    if not os.path.exists(gold):
        gold = str(tmp_path / "gold.1.cfr")
        with gzip.open(os.path.join(PROT, name + ".1.cfr.gz"), "rb") as fi, open(gold, "wb") as fo:
            shutil.copyfileobj(fi, fo)
    mine, ref = parse_1cfr(prefix + ".1.cfr", protein=True), parse_1cfr(gold, protein=True)
    assert len(mine) == len(ref)
    for (na, va), (nb, vb) in zip(mine, ref):
        assert na == nb and va == vb, f"field {na} differs"
    assert open(prefix + ".2.cfr", "rb").read() == open(os.path.join(PROT, name + ".2.cfr"), "rb").read()
    assert open(prefix + ".4.cfr").read().split("\n")[:3] == open(os.path.join(PROT, name + ".4.cfr")).read().split("\n")[:3]
