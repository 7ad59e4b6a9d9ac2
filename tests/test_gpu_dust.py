"""SDUST on the device (k_dust) against the host twin (cfr_dust_mask_batch, itself pinned to the oracle and through it to the
reference's Dustmasker): byte-identical masks on random, low-complexity and adversarial reads, and classification with the
device pre-step == host pre-step + classification.  -m gpu."""
import os

import numpy as np
import pytest

from centrifuger_amd import capi

pytestmark = pytest.mark.gpu


def _reads(rng, n):
    """A mix that exercises every part of the scan: random DNA, homopolymers, di/tri-nucleotide repeats, biased composition,
    N runs shorter and longer than the window, lower case, reads shorter than a triplet."""
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = []
    for i in range(n):
        L = int(rng.integers(0, 400))
        kind = i % 10
        if kind == 0:
            r = acgt[rng.integers(0, 4, size=L)]
        elif kind == 1:
            r = np.full(L, acgt[rng.integers(0, 4)], dtype=np.uint8)
        elif kind == 2:
            unit = acgt[rng.integers(0, 4, size=int(rng.integers(2, 7)))]
            r = np.resize(unit, L)
        elif kind == 3:
            r = acgt[rng.choice(4, size=L, p=[0.85, 0.05, 0.05, 0.05])]
        elif kind == 4:
            r = acgt[rng.integers(0, 4, size=L)].copy()
            if L > 40:
                a = int(rng.integers(0, L - 30))
                r[a:a + int(rng.integers(5, 30))] = ord("A")
        elif kind == 5:
            r = acgt[rng.integers(0, 4, size=L)].copy()
            if L > 100:
                a = int(rng.integers(0, L - 90))
                r[a:a + int(rng.integers(1, 90))] = ord("N")        # runs below and above the 64-base window
        elif kind == 6:
            r = np.frombuffer(bytes(acgt[rng.choice(4, size=L, p=[0.7, 0.1, 0.1, 0.1])]).lower(), dtype=np.uint8)
        elif kind == 7:
            r = acgt[rng.choice(2, size=L)]
        elif kind == 8:
            r = np.concatenate([np.full(L // 2, ord("T"), dtype=np.uint8), acgt[rng.integers(0, 4, size=L - L // 2)]])
        else:
            r = acgt[rng.integers(0, 4, size=min(L, 2))]
        out.append(np.ascontiguousarray(r, dtype=np.uint8))
    offs = np.concatenate([[0], np.cumsum([len(r) for r in out])]).astype(np.uint64)
    return np.concatenate(out) if out else np.zeros(0, dtype=np.uint8), offs


@pytest.fixture(scope="module")
def dev(golden_dir):
    idx = capi.Index(os.path.join(golden_dir, "f6"), capi.default_params(max_result=3))
    d = capi.DeviceIndex(idx)
    yield idx, d
    d.close()


def test_device_masks_equal_host_masks(dev):
    _, d = dev
    rng = np.random.default_rng(77)
    b, o = _reads(rng, 30_000)
    host = b.copy()
    capi.dust_mask(host, o, threads=8, literal=True)
    got = b.copy()
    d.dust_mask(got, o)
    assert np.array_equal(got, host)
    assert int((host != b).sum()) > 100_000          # the mix really has low-complexity sequence


def test_non_symbol_runs_at_every_place(dev):
    """Where the device scan takes its side paths: non-symbols at the first, second and third base, leading / trailing /
    inner runs around the 64-base limit (a longer run closes the segment, the next one starts from scratch), several runs
    in one read with low-complexity stretches between them, reads of non-symbols only - at every alignment of the buffer."""
    idx, d = dev
    rng = np.random.default_rng(77)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = []

    def lowc(L):
        kind = int(rng.integers(0, 3))
        if kind == 0:
            return np.full(L, acgt[rng.integers(0, 4)], dtype=np.uint8)
        if kind == 1:
            return np.resize(acgt[rng.integers(0, 4, size=int(rng.integers(2, 5)))], L)
        return acgt[rng.integers(0, 4, size=L)]

    for lead in (0, 1, 2, 3, 5, 63, 64, 65, 66, 130):
        for inner in (0, 1, 2, 63, 64, 65, 66, 100):
            for trail in (0, 1, 2, 64, 65, 70):
                parts = [np.full(lead, ord("N"), dtype=np.uint8), lowc(int(rng.integers(1, 120)))]
                if inner:
                    parts += [np.full(inner, ord("n" if inner % 2 else "N"), dtype=np.uint8), lowc(int(rng.integers(1, 120)))]
                    if inner in (2, 65):
                        parts += [np.full(inner + 1, ord("X"), dtype=np.uint8), lowc(int(rng.integers(3, 90)))]
                parts.append(np.full(trail, ord("N"), dtype=np.uint8))
                out.append(np.concatenate(parts))
    for L in (0, 1, 2, 3, 4, 63, 64, 65, 200):
        out.append(np.full(L, ord("N"), dtype=np.uint8))
        out.append(np.concatenate([np.array([ord("A")], dtype=np.uint8), np.full(L, ord("N"), dtype=np.uint8), lowc(40)]))
        out.append(np.concatenate([np.array([ord("A"), ord("C")], dtype=np.uint8), np.full(L, ord("N"), dtype=np.uint8), lowc(40)]))
    b = np.concatenate(out)
    o = np.concatenate([[0], np.cumsum([len(r) for r in out])]).astype(np.uint64)
    host = b.copy()
    capi.dust_mask(host, o, threads=1, literal=True)
    for shift in (0, 1, 2, 3, 5):                      # the buffer's alignment moves the 4-byte refills of the base window
        buf = np.concatenate([np.full(shift, ord("G"), dtype=np.uint8), b])
        oo = np.concatenate([[0], o + np.uint64(shift)]).astype(np.uint64)
        got = buf.copy()
        d.dust_mask(got, oo)
        assert np.array_equal(got[shift:], host), shift
        assert np.array_equal(got[:shift], buf[:shift])


def test_long_homopolymers_and_repeats(dev):
    """A 6 kbp homopolymer drives the reference's list of perfect intervals to its maximum (1711 entries, rescanned per window
    suffix); the device keeps one entry per start instead and must still produce the same masks."""
    _, d = dev
    rng = np.random.default_rng(78)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    reads = [np.full(6000, ord("A"), dtype=np.uint8), acgt[rng.integers(0, 4, size=300)], np.resize(np.frombuffer(b"AC", dtype=np.uint8), 5000),
             acgt[rng.integers(0, 4, size=150)], np.full(3000, ord("G"), dtype=np.uint8)]
    b = np.concatenate(reads)
    o = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.uint64)
    host = b.copy()
    capi.dust_mask(host, o, threads=1, literal=True)
    got = b.copy()
    d.dust_mask(got, o)
    assert np.array_equal(got, host)


def test_classification_with_the_device_pre_step(dev, golden_dir):
    """set_dust(1): unmasked reads in, same results as host masking + classification; the caller's buffer is not modified."""
    idx, d = dev
    recs = open(os.path.join(golden_dir, "se.fq"), "rb").read().split(b"\n")
    seqs = [recs[i + 1] for i in range(0, len(recs) - 1, 4)]
    rng = np.random.default_rng(3)
    seqs += [b"A" * 120 + bytes(rng.choice(list(b"ACGT"), size=60).astype(np.uint8)) for _ in range(50)] + [b"ACACACACAC" * 15] * 5
    b = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy()
    o = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.uint64)
    masked = b.copy()
    capi.dust_mask(masked, o, threads=4)
    want_r, want_m = d.classify(masked, o)
    d.set_dust(True)
    try:
        keep = b.copy()
        got_r, got_m = d.classify(b, o)
        assert np.array_equal(b, keep)
        assert got_r.tobytes() == want_r.tobytes() and got_m.tobytes() == want_m.tobytes()
        # paired entry with the same reads as mates
        gp_r, gp_m = d.classify(b, o, b, o)
    finally:
        d.set_dust(False)
    wp_r, wp_m = d.classify(masked, o, masked, o)
    assert gp_r.tobytes() == wp_r.tobytes() and gp_m.tobytes() == wp_m.tobytes()
    assert int((masked != b).sum()) > 1000


def test_resident_reads_at_an_unaligned_device_pointer(dev, golden_dir):
    """cfr_classify_batch_resident with SDUST on the device and a bases pointer that is not 4-byte aligned (a caller's slice of its own
    buffer): the screen and the scan fetch aligned words around the read, so the first read's words start BELOW the pointer; the first reads
    are homopolymers, which a screen that lost its first bases would let through unmasked."""
    import torch
    idx, d = dev
    rng = np.random.default_rng(9)
    seqs = [b"A" * 9 + bytes(rng.choice(list(b"ACGT"), size=141).astype(np.uint8)), b"ACGT" + b"C" * 8 + bytes(rng.choice(list(b"ACGT"), size=138).astype(np.uint8))]
    recs = open(os.path.join(golden_dir, "se.fq"), "rb").read().split(b"\n")
    seqs += [recs[i + 1] for i in range(0, len(recs) - 1, 4)][:200]
    b = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy()
    o = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.uint64)
    masked = b.copy()
    capi.dust_mask(masked, o, threads=2)
    assert (masked[:20] != b[:20]).any()
    want_r, want_m = d.classify(masked, o)
    dv = torch.device("cuda")
    do = torch.from_numpy(o.astype(np.int64)).to(dv)
    d.set_dust(True)
    try:
        for shift in (0, 1, 2, 3):
            big = torch.zeros(len(b) + 64, dtype=torch.uint8, device=dv)
            big[shift:shift + len(b)] = torch.from_numpy(b).to(dv)
            torch.cuda.synchronize()
            got_r, got_m = d.classify_resident(big.data_ptr() + shift, do.data_ptr(), len(seqs), int(o[-1]))
            assert got_r.tobytes() == want_r.tobytes() and got_m[:len(want_m)].tobytes() == want_m.tobytes(), shift
            assert bytes(big[shift:shift + len(b)].cpu().numpy()) == b.tobytes()          # the caller's reads stay as they were
    finally:
        d.set_dust(False)


def _dust_golden():
    import gzip
    import oracle_lib as ora
    from conftest import GOLDEN
    ids, b, o = ora.read_fastx(os.path.join(GOLDEN, "dust", "reads.fa"))
    want = {}
    lines = gzip.open(os.path.join(GOLDEN, "dust", "masked_by_reference.fa.gz"), "rb").read().split(b"\n")
    for i in range(0, len(lines) - 1, 2):
        want[lines[i][1:].decode()] = lines[i + 1]
    return ids, b, o, want


def test_device_masks_equal_the_reference_dumps(dev):
    """1580 adversarial reads (homopolymers around the window size and of 6 kbp, tandem repeats of period 1-13 with 3-8 copies,
    biased composition, N runs, scattered N, lower case) masked on the device against what the REFERENCE wrote into its
    --un/--cl dumps for them (tests/golden/dust, made by tests/golden/make_golden_dust.py)."""
    _, d = dev
    ids, b, o, want = _dust_golden()
    got = b.copy()
    d.dust_mask(got, o)
    bad = [ids[i] for i in range(len(ids)) if bytes(got[int(o[i]):int(o[i + 1])]) != want[ids[i]]]
    assert not bad, bad[:10]
    assert sum(1 for i in range(len(ids)) if bytes(b[int(o[i]):int(o[i + 1])]) != want[ids[i]]) > 500


def test_device_masks_equal_the_second_set_of_reference_dumps(dev):
    """Set 2 of tests/golden/dust (non-symbol runs of every length around the window at every offset, runs at the ends, low
    complexity beside a run, long reads of mixed stretches, 3 000 reads of 150 bp): the device scan against what the REFERENCE
    wrote for them."""
    from test_host_cpu import _dust_set2
    _, d = dev
    ids, b, o, want = _dust_set2()
    got = b.copy()
    d.dust_mask(got, o)
    bad = [ids[i] for i in range(len(ids)) if bytes(got[int(o[i]):int(o[i + 1])]) != want[ids[i]]]
    assert not bad, bad[:10]


def test_one_instantiation_for_all_reads():
    """The same tests with CFR_DUST_SPLIT=0: every read through the 125-triplet instantiation of k_dust (no flag pass, no
    ACGT-only instantiation) - the two forms must mask alike."""
    import subprocess
    import sys
    env = dict(os.environ, CFR_DEBUG_ENV="1", CFR_DUST_SPLIT="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-k", "not one_instantiation"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    tail = r.stdout.decode()[-1500:]
    assert r.returncode == 0, tail
    assert " passed" in tail


def test_reads_with_one_non_symbol_go_through_the_64_triplet_kernel(dev):
    """A read with exactly ONE character that is not A, C, G or T is scanned by the 64-triplet instantiation with that character's
    three triplets as uncounted placeholders (k_dust, cfr_kernels.hip.inc); two or more take the 125-triplet one.  20 000 reads of
    repeats and random stretches with one such character (N, n, X, a lower-case base) at any place - first, second, third, last
    base included - and 2 000 with two, against the literal host twin (which is pinned to the reference's own dumps)."""
    idx, d = dev
    rng = np.random.default_rng(1234)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    odd = np.frombuffer(b"NnXacgt-", dtype=np.uint8)
    out = []
    for i in range(22000):
        L = int(rng.integers(1, 320))
        parts, have = [], 0
        while have < L:
            k = int(rng.integers(1, 90))
            kind = int(rng.integers(0, 4))
            if kind == 0:
                p = np.full(k, acgt[rng.integers(0, 4)], dtype=np.uint8)
            elif kind == 1:
                p = np.resize(acgt[rng.integers(0, 4, size=int(rng.integers(2, 6)))], k)
            else:
                p = acgt[rng.integers(0, 4, size=k)]
            parts.append(p)
            have += k
        r = np.concatenate(parts)[:L].copy()
        for _ in range(1 if i < 20000 else 2):
            r[int(rng.integers(0, L))] = odd[rng.integers(0, len(odd))]
        out.append(r)
    for pos in (0, 1, 2, 3, 60, 61, 62, 63, 64, 65, 147, 148, 149):       # the places the window logic cares about, on a poly-A read
        r = np.full(150, ord("A"), dtype=np.uint8)
        r[pos] = ord("N")
        out.append(r)
    b = np.concatenate(out)
    o = np.concatenate([[0], np.cumsum([len(r) for r in out])]).astype(np.uint64)
    host = b.copy()
    capi.dust_mask(host, o, threads=8, literal=True)
    for shift in (0, 3):
        buf = np.concatenate([np.full(shift, ord("G"), dtype=np.uint8), b])
        oo = np.concatenate([[0], o + np.uint64(shift)]).astype(np.uint64)
        got = buf.copy()
        d.dust_mask(got, oo)
        assert np.array_equal(got[shift:], host), shift
    assert (host != b).sum() > 100000          # (the set does get masked)
