"""The criterion of k_dust_screen (csrc/cfr_kernels.hip.inc), restated step for step and checked on the CPU against the literal host twin of
the reference's SDUST (capi.dust_mask(literal=True), Dustmasker.hpp:106-243): a read the screen leaves unflagged must come out of the scan
unchanged - for the mix of reads tests/test_gpu_dust.py masks on the device (homopolymers, short tandem repeats, skewed compositions, one
non-symbol, every length from 0 to 400) and for a set of short periodic reads around the smallest perfect intervals (five equal triplets)."""
import numpy as np

from centrifuger_amd import capi
from test_gpu_dust import _reads

CODE = {ord("A"): 0, ord("C"): 1, ord("G"): 2, ord("T"): 3}


def screen_flags(read):
    """True = the read goes through the full scan.  The kernel's loop: a sliding window of 62 triplets; triplets with a non-symbol are not counted."""
    n = len(read)
    if n < 7:
        return False
    codes = [CODE.get(int(c), 4) for c in read]
    kinds = [None, None] + [None if 4 in codes[j - 2:j + 1] else codes[j - 2] * 16 + codes[j - 1] * 4 + codes[j] for j in range(2, n)]
    cnt = [0] * 64
    for j in range(n):
        if j >= 62 and kinds[j - 62] is not None:
            cnt[kinds[j - 62]] -= 1
        k = kinds[j]
        if k is None:
            continue
        before = cnt[k]
        cnt[k] += 1
        if before >= 5:
            return True
        if before >= 4:
            pk = kinds[j - 1]
            if pk is not None and cnt[pk] >= 5:
                return True
    return False


def _check(b, o):
    host = b.copy()
    capi.dust_mask(host, o, threads=4, literal=True)
    n = len(o) - 1
    pure = masked = skipped = 0
    for i in range(n):
        r = b[int(o[i]):int(o[i + 1])]
        if sum(1 for c in r if int(c) not in CODE) > 1:
            continue                                          # k_dust_flags sends it to the other instantiation: never screened
        pure += 1
        changed = bool((host[int(o[i]):int(o[i + 1])] != r).any())
        masked += changed
        if not screen_flags(r):
            skipped += 1
            assert not changed, (i, bytes(r))
    return pure, masked, skipped


def test_unflagged_reads_are_left_unchanged_by_the_scan():
    rng = np.random.default_rng(77)
    b, o = _reads(rng, 6000)
    pure, masked, skipped = _check(b, o)
    assert masked > 1000 and skipped > 1000, (pure, masked, skipped)


def test_smallest_perfect_intervals():
    """five equal triplets (a homopolymer of 7), two kinds five times each (a dinucleotide repeat of 12), three kinds (15 + 2) ... inside random flanks"""
    rng = np.random.default_rng(5)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = []
    for period in range(1, 13):
        for copies in range(2, 9):
            for extra in range(0, 3):
                for flank in (0, 3, 40):
                    unit = acgt[rng.integers(0, 4, size=period)]
                    rep = np.resize(unit, period * copies + extra)
                    out.append(np.concatenate([acgt[rng.integers(0, 4, size=flank)], rep, acgt[rng.integers(0, 4, size=flank)]]))
    o = np.concatenate([[0], np.cumsum([len(r) for r in out])]).astype(np.uint64)
    pure, masked, skipped = _check(np.concatenate(out), o)
    assert masked > 100 and skipped > 100, (pure, masked, skipped)
