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


def reads2(rng):
    """Second set (round 3): what the first one is thin on - runs of non-symbols of every length around the window size at every
    offset of a read (a run of more than 64 closes a segment, Dustmasker.hpp / CentrifugerClass.cpp:283-289), runs at the ends,
    low complexity right behind a run, long reads (1-8 kbp) made of random, biased, tandem and homopolymer stretches."""
    out = []

    def rnd(L):
        return ACGT[rng.integers(0, 4, size=L)]
    base = rnd(300)
    for run in (1, 2, 3, 62, 63, 64, 65, 66, 67, 128, 129, 200):
        for off in range(0, 70):
            r = base.copy()
            r[off:off + run] = ord("N")
            out.append(r[:max(off + run + 5, 150 + (off % 7))].copy() if off % 3 else r)
    for run in (1, 5, 64, 65, 70):                                               # runs at the very start / end, and the whole read
        r = rnd(200).copy(); r[:run] = ord("N"); out.append(r)
        r = rnd(200).copy(); r[-run:] = ord("N"); out.append(r)
        out.append(np.full(run, ord("N"), dtype=np.uint8))
    for k in range(400):                                                         # low complexity directly behind / in front of a run
        L = int(rng.integers(100, 400))
        r = rnd(L).copy()
        a = int(rng.integers(0, L - 80))
        run = int(rng.choice([1, 3, 10, 63, 64, 65, 66]))
        r[a:a + run] = ord("N")
        b = min(L, a + run)
        fill = int(rng.integers(8, 60))
        r[b:b + fill] = np.resize(rnd(int(rng.integers(1, 4))), min(fill, L - b))
        if k % 2:
            r[max(0, a - fill):a] = np.resize(rnd(int(rng.integers(1, 4))), a - max(0, a - fill))
        out.append(r)
    for k in range(60):                                                          # long reads of mixed stretches
        parts = []
        total = int(rng.integers(1000, 7000))               # (the reference's gzprintf drops dump records of 8192 bytes and more)
        while sum(len(p) for p in parts) < total:
            kind = int(rng.integers(0, 6))
            L = int(rng.integers(20, 600))
            if kind == 0:
                parts.append(rnd(L))
            elif kind == 1:
                parts.append(ACGT[rng.choice(4, size=L, p=[0.8, 0.1, 0.05, 0.05])])
            elif kind == 2:
                parts.append(np.resize(rnd(int(rng.integers(1, 8))), L))
            elif kind == 3:
                parts.append(np.full(int(rng.integers(5, 200)), ACGT[int(rng.integers(0, 4))], dtype=np.uint8))
            elif kind == 4:
                parts.append(np.full(int(rng.choice([1, 2, 30, 64, 65, 90])), ord("N"), dtype=np.uint8))
            else:
                parts.append(np.frombuffer(bytes(rnd(L)).lower(), dtype=np.uint8))
        out.append(np.ascontiguousarray(np.concatenate(parts), dtype=np.uint8)[:8100])
    for k in range(3000):                                                        # 150 bp reads, the bench's length
        kind = k % 5
        if kind == 0:
            r = ACGT[rng.choice(4, size=150, p=[0.6, 0.2, 0.1, 0.1])]
        elif kind == 1:
            r = rnd(150).copy(); a = int(rng.integers(0, 120)); r[a:a + int(rng.integers(6, 30))] = ACGT[int(rng.integers(0, 4))]
        elif kind == 2:
            r = rnd(150).copy(); a = int(rng.integers(0, 100)); u = rnd(int(rng.integers(2, 5))); n_ = int(rng.integers(10, 50)); r[a:a + n_] = np.resize(u, min(n_, 150 - a))
        elif kind == 3:
            r = rnd(150).copy(); r[rng.random(150) < 0.01] = ord("N")
        else:
            r = rnd(150)
        out.append(np.ascontiguousarray(r, dtype=np.uint8))
    return out


def run_reference(rs, fa):
    tmp = tempfile.mkdtemp(prefix="cfr_dust_")
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
    return masked


def main2():
    rng = np.random.default_rng(SEED + 1)
    rs = reads2(rng)
    tmp = tempfile.mkdtemp(prefix="cfr_dust2_")
    fa = os.path.join(tmp, "reads2.fa")
    with open(fa, "wb") as f:
        for i, r in enumerate(rs):
            f.write(b">e%d\n%s\n" % (i, r.tobytes()))
    masked = run_reference(rs, fa)
    with gzip.GzipFile(os.path.join(OUT, "reads2.fa.gz"), "wb", mtime=0) as f:
        f.write(open(fa, "rb").read())
    with gzip.GzipFile(os.path.join(OUT, "masked2_by_reference.fa.gz"), "wb", mtime=0) as f:
        for i in range(len(rs)):
            f.write(b">e%d\n%s\n" % (i, masked.get(b"e%d" % i, b"")))
    print("set 2:", len(rs), "reads;", sum(1 for i, r in enumerate(rs) if masked.get(b"e%d" % i, b"") != r.tobytes()), "masked somewhere")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "set2":
        return main2()
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
