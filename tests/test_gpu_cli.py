"""The `centrifuger`-compatible command line (centrifuger_amd/bin/centrifuger) against the reference's TSV
and read dumps.  -m gpu."""
import gzip
import json
import os
import subprocess

import pytest

from conftest import GOLDEN, REF_DIR, ROOT, have_ref

pytestmark = pytest.mark.gpu
MAN = json.load(open(os.path.join(GOLDEN, "manifest.json")))
CLI = os.path.join(ROOT, "centrifuger_amd", "bin", "centrifuger")


def _args(args, gd):
    return [os.path.join(gd, a) if a.endswith((".fq", ".fa")) else a for a in args]


@pytest.mark.parametrize("case", sorted(MAN["cases"]))
def test_cli_stdout_equals_reference_tsv(case, golden_dir):
    c = MAN["cases"][case]
    out = subprocess.run([CLI, "-x", os.path.join(golden_dir, c["index"]), "-t", "3", "--gpu-batch", "97"] + _args(c["args"], golden_dir),
                         check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert out.stdout == open(os.path.join(GOLDEN, "tsv", case + ".tsv"), "rb").read()
    assert b"can be classified." in out.stderr and b"Centrifuger finishes." in out.stderr


def test_cli_gz_input_interleaved_and_batches(golden_dir, tmp_path):
    # interleave pe_1/pe_2 into one gz file; result must equal the -1/-2 golden
    l1 = open(os.path.join(golden_dir, "pe_1.fq"), "rb").read().split(b"\n")
    l2 = open(os.path.join(golden_dir, "pe_2.fq"), "rb").read().split(b"\n")
    inter = []
    for i in range(0, len(l1) - 1, 4):
        inter += l1[i:i + 4] + l2[i:i + 4]
    p = tmp_path / "inter.fq.gz"
    with gzip.open(p, "wb") as f:
        f.write(b"\n".join(inter) + b"\n")
    out = subprocess.run([CLI, "-x", os.path.join(golden_dir, "f6"), "-i", str(p), "-k", "5", "--gpu-batch", "33"],
                         check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
    assert out == open(os.path.join(GOLDEN, "tsv", "f6.pe_k5.tsv"), "rb").read()


def test_cli_several_device_workers_keep_input_order(golden_dir):
    """--gpu LIST starts one worker (and one device index) per entry; three workers on ordinal 0 race over 25-read batches and
    the writer must still emit the reference's rows in input order.  Also the throughput profile through the command line."""
    want = open(os.path.join(GOLDEN, "tsv", "f6.pe_k5.tsv"), "rb").read()
    base = [CLI, "-x", os.path.join(golden_dir, "f6"), "-1", os.path.join(golden_dir, "pe_1.fq"), "-2", os.path.join(golden_dir, "pe_2.fq"), "-k", "5"]
    out = subprocess.run(base + ["--gpu", "0,0,0", "--gpu-batch", "25", "-t", "4"], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
    assert out == want
    out = subprocess.run(base + ["--gpu", "all", "--gpu-throughput"], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
    assert out == want
    out = subprocess.run(base + ["--gpu-balanced"], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
    assert out == want


def test_cli_parallel_reader_of_plain_files(golden_dir, tmp_path):
    """Plain single-end files are cut at verified record starts and parsed by several threads; rows must come out in file
    order.  A 64 KB piece size cuts se.fq (400 reads) and a concatenation of it in several pieces; --parse-threads 1 is the
    sequential reader; both must print the reference's TSV."""
    want = open(os.path.join(GOLDEN, "tsv", "f6.se_default.tsv"), "rb").read()
    se = os.path.join(golden_dir, "se.fq")
    for extra in (["--parse-threads", "4", "--gpu-batch", "50"], ["--parse-threads", "1", "--gpu-batch", "50"], ["--gpu-batch", "60", "-t", "8"]):
        out = subprocess.run([CLI, "-x", os.path.join(golden_dir, "f6"), "-u", se] + extra, check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
        assert out == want, extra
    # five copies back to back (ids repeat; the reference would print the same rows five times)
    big = tmp_path / "big.fq"
    big.write_bytes(open(se, "rb").read() * 5)
    rows = want.split(b"\n")[1:-1]
    out = subprocess.run([CLI, "-x", os.path.join(golden_dir, "f6"), "-u", str(big), "--parse-threads", "6", "--gpu-batch", "70", "-t", "6"],
                         check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
    assert out.split(b"\n")[1:-1] == rows * 5


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref (compiled reference) not present")
def test_cli_read_dumps_match_reference(golden_dir, tmp_path):
    for tool, tag in ((os.path.join(REF_DIR, "centrifuger"), "ref"), (CLI, "gpu")):
        subprocess.run([tool, "-x", os.path.join(golden_dir, "f6"), "-u", os.path.join(golden_dir, "edge.fa"),
                        "--un", str(tmp_path / f"{tag}_un"), "--cl", str(tmp_path / f"{tag}_cl")],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        subprocess.run([tool, "-x", os.path.join(golden_dir, "f6"), "-1", os.path.join(golden_dir, "pe_1.fq"), "-2",
                        os.path.join(golden_dir, "pe_2.fq"), "--un", str(tmp_path / f"{tag}_pun"), "--cl", str(tmp_path / f"{tag}_pcl")],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for name in ("un.fq.gz", "cl.fq.gz", "pun_1.fq.gz", "pun_2.fq.gz", "pcl_1.fq.gz", "pcl_2.fq.gz"):
        assert gzip.open(tmp_path / f"gpu_{name}").read() == gzip.open(tmp_path / f"ref_{name}").read(), name
    # a record that does not fit zlib's 8192-byte gzprintf buffer is dropped by the reference's dumps (ResultWriter.hpp:254-265): same here
    import numpy as np
    rng = np.random.default_rng(1)
    long_read = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=9000)])
    (tmp_path / "long.fa").write_bytes(b">short\nACGTACGTACGTAAAAAAAAAAAAAAAAAAAAAACGT\n>long9k\n" + long_read + b"\n>tail\nGGGGCCCCAAAATTTTACGTACGT\n")
    for tool, tag in ((os.path.join(REF_DIR, "centrifuger"), "ref"), (CLI, "gpu")):
        subprocess.run([tool, "-x", os.path.join(golden_dir, "f6"), "-u", str(tmp_path / "long.fa"), "--un", str(tmp_path / f"{tag}_lun"), "--cl", str(tmp_path / f"{tag}_lcl")],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for name in ("lun.fq.gz", "lcl.fq.gz"):
        assert gzip.open(tmp_path / f"gpu_{name}").read() == gzip.open(tmp_path / f"ref_{name}").read(), name
    assert b"long9k" not in gzip.open(tmp_path / "ref_lun.fq.gz").read() + gzip.open(tmp_path / "ref_lcl.fq.gz").read()


OPTION_SETS = {
    "minhit10_k3": ["--min-hitlen", "10", "-k", "3"],            # hits shorter than the score offset: scores of 0 enter the fold
    "minhit12": ["--min-hitlen", "12"],
    "secondary_1_0.5_k2": ["--consider-secondary", "1,0.5", "-k", "2"],
    "secondary_30_0.9_k4": ["--consider-secondary", "30,0.9", "-k", "4"],
    "hitk1": ["--hitk-factor", "1"],
    "hitk_negative_k2": ["--hitk-factor", "-1", "-k", "2"],
    "k64": ["-k", "64"],
    "k200": ["-k", "200"],                                        # (the reference has no cap on -k; round 4 refused anything above 64)
}


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref (compiled reference) not present")
@pytest.mark.parametrize("name", sorted(OPTION_SETS))
@pytest.mark.parametrize("profile", ["fast-load", "throughput"])       # (fast-load is the command line's default)
def test_cli_option_corners_against_the_live_reference(name, profile, golden_dir):
    """Option values no committed fixture covers, checked against the reference binary run on the spot (both profiles of the
    command line: the multi-kernel path without derived tables, and the one-launch path with all of them)."""
    opts = OPTION_SETS[name]
    for reads in (["-u", os.path.join(golden_dir, "se.fq")], ["-1", os.path.join(golden_dir, "pe_1.fq"), "-2", os.path.join(golden_dir, "pe_2.fq")],
                  ["-u", os.path.join(golden_dir, "edge.fa")]):
        want = subprocess.run([os.path.join(REF_DIR, "centrifuger"), "-x", os.path.join(golden_dir, "f6"), "-t", "2"] + reads + opts,
                              check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
        got = subprocess.run([CLI, "-x", os.path.join(golden_dir, "f6"), "-t", "2"] + reads + opts + (["--gpu-throughput"] if profile == "throughput" else []),
                             check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
        assert got == want, (name, profile, reads[0])


def _device_count():
    from centrifuger_amd import capi
    import ctypes
    c = ctypes.c_int(0)
    capi.lib().cfr_device_count(ctypes.byref(c))
    return c.value


def test_cli_over_distinct_devices(golden_dir):
    """--gpu all / --gpu 0,1 on DISTINCT ordinals (CentrifugerClass.cpp:681-694: the per-batch fan-out this library replaces becomes
    one worker and one index image per GPU): rows in input order, equal to the reference's TSV, whichever device served a batch."""
    if _device_count() < 2:          # (asked at run time: a HIP call at import time would come before torch's, see conftest.py)
        pytest.skip("needs at least two MI355X in the box (the round-end 8-GPU node has them)")
    want = open(os.path.join(GOLDEN, "tsv", "f6.pe_k5.tsv"), "rb").read()
    base = [CLI, "-x", os.path.join(golden_dir, "f6"), "-1", os.path.join(golden_dir, "pe_1.fq"), "-2", os.path.join(golden_dir, "pe_2.fq"), "-k", "5", "--gpu-batch", "20", "-t", "4"]
    last = str(_device_count() - 1)
    print(f"distinct devices: {_device_count()} in the box; --gpu all, 0,1 and {last},0")
    for gpus in ("all", "0,1", last + ",0"):
        out = subprocess.run(base + ["--gpu", gpus], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
        assert out == want, gpus
    want_se = open(os.path.join(GOLDEN, "tsv", "f6.se_default.tsv"), "rb").read()
    out = subprocess.run([CLI, "-x", os.path.join(golden_dir, "f6"), "-u", os.path.join(golden_dir, "se.fq"), "--gpu", "all", "--gpu-batch", "30", "--gpu-throughput"],
                         check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
    assert out == want_se


def test_cli_parallel_reader_of_paired_and_interleaved_plain_files(golden_dir, tmp_path):
    """-1/-2 and -i from plain files: both mate files are cut at the same record numbers and parsed by several threads
    (ReadFiles.hpp:337 reads them with one thread); rows must equal the reference's TSV whatever the piece size and thread count,
    and --parse-threads 1 (the sequential reader) must agree."""
    want = open(os.path.join(GOLDEN, "tsv", "f6.pe_k5.tsv"), "rb").read()
    m1, m2 = os.path.join(golden_dir, "pe_1.fq"), os.path.join(golden_dir, "pe_2.fq")
    l1 = open(m1, "rb").read().split(b"\n")
    l2 = open(m2, "rb").read().split(b"\n")
    inter = []
    for i in range(0, len(l1) - 1, 4):
        inter += l1[i:i + 4] + l2[i:i + 4]
    p = tmp_path / "inter.fq"
    p.write_bytes(b"\n".join(inter) + b"\n")
    for extra in (["--parse-threads", "5", "--gpu-batch", "17"], ["--parse-threads", "1", "--gpu-batch", "17"], ["--gpu-batch", "64", "-t", "8"]):
        out = subprocess.run([CLI, "-x", os.path.join(golden_dir, "f6"), "-1", m1, "-2", m2, "-k", "5"] + extra, check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
        assert out == want, extra
        out = subprocess.run([CLI, "-x", os.path.join(golden_dir, "f6"), "-i", str(p), "-k", "5"] + extra, check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
        assert out == want, extra
    # mates of different counts are an error, as in the reference
    short = tmp_path / "short_2.fq"
    short.write_bytes(b"\n".join(l2[:4 * 150]) + b"\n")
    r = subprocess.run([CLI, "-x", os.path.join(golden_dir, "f6"), "-1", m1, "-2", str(short), "--gpu-batch", "17"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"different number of reads" in r.stderr


def test_cli_large_k_with_the_default_batch_size(golden_dir, tmp_path):
    """-k 4096 (the cap) with the default --gpu-batch: the match buffers have max_result slots per read, so the command line cuts its
    batches down (2 GB of slots) instead of sizing 262144 x 4096 of them; 200 000 reads, rows in input order and equal to
    the rows of a small-batch run of the same reads"""
    import time
    se = os.path.join(golden_dir, "se.fq")
    big = tmp_path / "big.fq"
    big.write_bytes(open(se, "rb").read() * 500)
    base = [CLI, "-x", os.path.join(golden_dir, "f6"), "-k", "4096", "-t", "8"]
    want = subprocess.run(base + ["-u", se, "--gpu-batch", "64"], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
    t0 = time.time()
    out = subprocess.run(base + ["-u", str(big)], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
    assert time.time() - t0 < 120
    assert out.split(b"\n")[1:-1] == want.split(b"\n")[1:-1] * 500
