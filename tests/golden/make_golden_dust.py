#!/usr/bin/env python3
"""Regenerate tests/golden/dust/*: an adversarial read set and the masked reads the REAL reference writes for it
(`centrifuger --un/--cl` dumps: the reference masks in place before Query, CentrifugerClass.cpp:276-316, so its dumps hold the
SDUST-masked sequences).  Dev container only:  make -C oracle ref && python tests/golden/make_golden_dust.py
Committed: data only (reads + the reference's masked reads)."""
import gzip
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "dust")
REF = os.path.join(ROOT, "oracle", "_ref")
SEED = 20260930
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def reads(rng):
    out = []

    def rnd(L):
        return ACGT[rng.integers(0, 4, size=L)]
    for L in (0, 1, 2, 3, 4, 7, 8, 9, 63, 64, 65, 66, 150):
        out.append(np.full(L, ord("A"), dtype=np.uint8))                       # homopolymers around the window size
    out.append(np.full(6000, ord("A"), dtype=np.uint8))                         # the reference's interval list at its maximum
    out.append(np.resize(np.frombuffer(b"AC", dtype=np.uint8), 5000))
    for period in range(1, 14):                                                  # tandem repeats, 3..8 copies
        unit = rnd(period)
        for copies in (3, 4, 5, 6, 8):
            out.append(np.concatenate([rnd(40), np.resize(unit, period * copies), rnd(40)]))
    for i in range(1500):
        L = int(rng.integers(20, 320))
        kind = i % 9
        if kind == 0:
            r = rnd(L)
        elif kind == 1:
            r = ACGT[rng.choice(4, size=L, p=[0.85, 0.05, 0.05, 0.05])]
        elif kind == 2:
            r = rnd(L).copy(); a = int(rng.integers(0, max(1, L - 30))); r[a:a + int(rng.integers(5, 30))] = ord("T")
        elif kind == 3:
            r = rnd(L).copy(); a = int(rng.integers(0, max(1, L - 90))); r[a:a + int(rng.integers(1, 90))] = ord("N")
        elif kind == 4:
            r = np.frombuffer(bytes(ACGT[rng.choice(4, size=L, p=[0.7, 0.1, 0.1, 0.1])]).lower(), dtype=np.uint8)
        elif kind == 5:
            r = ACGT[rng.choice(2, size=L)]
        elif kind == 6:
            r = np.concatenate([np.full(L // 2, ord("G"), dtype=np.uint8), rnd(L - L // 2)])
        elif kind == 7:
            r = rnd(L).copy(); r[rng.random(L) < 0.03] = ord("N")                # scattered N: two inside one window
        else:
            unit = rnd(int(rng.integers(2, 7))); r = np.resize(unit, L).copy(); r[rng.random(L) < 0.05] = ACGT[rng.integers(0, 4)]
        out.append(np.ascontiguousarray(r, dtype=np.uint8))
    return out


def main():
    rng = np.random.default_rng(SEED)
    os.makedirs(OUT, exist_ok=True)
    rs = [r for r in reads(rng)]
    fa = os.path.join(OUT, "reads.fa")
    with open(fa, "wb") as f:
        for i, r in enumerate(rs):
            f.write(b">d%d\n%s\n" % (i, r.tobytes()))
    tmp = tempfile.mkdtemp(prefix="cfr_dust_")
    # any index will do: the dumps hold the masked reads whether they classify or not
    prefix = os.path.join(HERE, "f6")
    subprocess.run([os.path.join(REF, "centrifuger"), "-x", prefix, "-u", fa, "--un", os.path.join(tmp, "un"), "--cl", os.path.join(tmp, "cl")],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    masked = {}
    for name in ("un.fq.gz", "cl.fq.gz"):
        lines = gzip.open(os.path.join(tmp, name), "rb").read().split(b"\n")
        i = 0
        while i < len(lines):
            if lines[i].startswith(b">"):
                masked[lines[i][1:]] = lines[i + 1]
                i += 2
            elif lines[i].startswith(b"@"):
                masked[lines[i][1:]] = lines[i + 1]
                i += 4
            else:
                i += 1
    with gzip.GzipFile(os.path.join(OUT, "masked_by_reference.fa.gz"), "wb", mtime=0) as f:
        for i in range(len(rs)):
            f.write(b">d%d\n%s\n" % (i, masked.get(b"d%d" % i, b"")))
    print(len(rs), "reads;", sum(1 for i, r in enumerate(rs) if masked.get(b"d%d" % i, b"") != r.tobytes()), "masked somewhere")


if __name__ == "__main__":
    main()
