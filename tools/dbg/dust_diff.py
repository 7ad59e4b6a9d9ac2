#!/usr/bin/env python3
"""device SDUST against the literal host twin on single-read cases with one non-symbol; prints the reads that differ"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import torch  # initialises HIP first (see tests/conftest.py)
torch.cuda.init()
from centrifuger_amd import capi
gold = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")
idx = capi.Index(os.path.join(gold, "f6"))
d = capi.DeviceIndex(idx)
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
reads = out
b = np.concatenate(reads); o = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.uint64)
host = b.copy(); capi.dust_mask(host, o, threads=1, literal=True)
got = b.copy(); d.dust_mask(got, o)
bad = 0
for i, r in enumerate(reads):
    a, e = int(o[i]), int(o[i + 1])
    if not np.array_equal(host[a:e], got[a:e]):
        bad += 1
        if bad <= 12:
            print("read", i, "len", e - a); print(" in  ", bytes(b[a:e]).decode()); print(" host", bytes(host[a:e]).decode()); print(" dev ", bytes(got[a:e]).decode())
print("differing reads:", bad, "of", len(reads))
