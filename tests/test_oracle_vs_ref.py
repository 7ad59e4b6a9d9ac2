"""The plain-C restatement (oracle/) against the REAL reference run live (oracle/_ref, compiled from /root/reference by
oracle/Makefile) on freshly seeded inputs that are NOT among the committed fixtures.  Dev container only: skipped where
oracle/_ref is absent (the committed golden vectors in test_oracle_golden.py cover that case).  CPU only."""
import os
import subprocess

import numpy as np
import pytest

from conftest import REF_DIR, have_ref
from centrifuger_amd import synth

pytestmark = pytest.mark.skipif(not have_ref(), reason="oracle/_ref (compiled reference) not present")


@pytest.fixture(scope="module")
def fresh(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("fresh"))
    g = synth.make_genomes(3, 3, 30000, seed=771)
    synth.write_reference_inputs(g, d)
    subprocess.run([os.path.join(REF_DIR, "centrifuger-build"), "-t", "2", "-r", os.path.join(d, "ref.fa"),
                    "--taxonomy-tree", os.path.join(d, "nodes.dmp"), "--name-table", os.path.join(d, "names.dmp"),
                    "--conversion-table", os.path.join(d, "seqid.map"), "-o", os.path.join(d, "idx")],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    synth.write_fastq(synth.make_reads(g, 1500, 120, seed=772, sub_rate=0.02), os.path.join(d, "se.fq"))
    r1, r2 = synth.make_pairs(g, 800, 100, seed=773)
    synth.write_fastq(r1, os.path.join(d, "p1.fq"), suffix="/1")
    synth.write_fastq(r2, os.path.join(d, "p2.fq"), suffix="/2")
    synth.write_fastq(synth.make_long_reads(g, 60, 800, 4000, seed=774), os.path.join(d, "long.fq"))
    return d


CASES = {
    "se_k1": ["-u", "se.fq"],
    "se_k5": ["-u", "se.fq", "-k", "5"],
    "se_minhit30": ["-u", "se.fq", "--min-hitlen", "30", "-k", "3"],
    "pe_k1": ["-1", "p1.fq", "-2", "p2.fq"],
    "pe_k4": ["-1", "p1.fq", "-2", "p2.fq", "-k", "4"],
    "long_k3": ["-u", "long.fq", "-k", "3"],
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_oracle_tsv_equals_live_reference(case, fresh, oracle_bin):
    args = [os.path.join(fresh, a) if a.endswith(".fq") else a for a in CASES[case]]
    ref = subprocess.run([os.path.join(REF_DIR, "centrifuger"), "-x", os.path.join(fresh, "idx"), "-t", "2"] + args,
                         check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
    ora = subprocess.run([oracle_bin, "classify", "-x", os.path.join(fresh, "idx")] + args, check=True, stdout=subprocess.PIPE).stdout
    assert ref.count(b"\n") > 50
    assert ora == ref
