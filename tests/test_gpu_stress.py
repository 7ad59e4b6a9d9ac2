"""cfr_index_open / cfr_index_destroy of every golden index, thousands of times, beside live device images with batches in flight
(csrc/cfr_stress.cpp through the C-ABI): every open must succeed, every digest must equal the first, every batch must reproduce
the first batch's results.  Round 3 saw the open of a golden protein index fail its decode check twice in ~110 suite runs; this is
the in-process stress VERDICT r3 asked for (the sanitizer variants of the same binary run from tools/stress_open.sh).  -m gpu."""
import gzip
import os
import shutil
import subprocess

import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "centrifuger_amd", "bin", "cfr_stress")


def _unpacked(tmp, sub, name):
    src = os.path.join(GOLDEN, sub)
    with gzip.open(os.path.join(src, name + ".1.cfr.gz"), "rb") as fi, open(tmp / (name + ".1.cfr"), "wb") as fo:
        shutil.copyfileobj(fi, fo)
    for ext in (".2.cfr", ".4.cfr"):
        shutil.copy(os.path.join(src, name + ext), tmp / (name + ext))
    return str(tmp / name)


def test_open_destroy_beside_live_images(tmp_path):
    f10 = _unpacked(tmp_path, "", "f10")
    p4 = _unpacked(tmp_path, "prot", "p4")
    prot = os.path.join(GOLDEN, "prot")
    opens = [os.path.join(GOLDEN, x) for x in ("f6", "f6_b1", "f6_b8", "f6_off3")] + [f10] + \
            [os.path.join(prot, x) for x in ("p2", "p2_b1_off2", "p3_b4")] + [p4]
    r = subprocess.run([BIN, "--load", f10 + "," + os.path.join(prot, "p3_b4"), "--open", ",".join(opens), "--reads", os.path.join(GOLDEN, "se.fq"),
                        "--opens", "2000", "--opener-threads", "2", "--devimg-every", "50"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0, out
    assert " 0 failures" in out and "18000 opens of 9 indexes" in out, out
