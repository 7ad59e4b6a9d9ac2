#!/usr/bin/env python3
"""k_dust alone on the device: 10 M x 150 bp reads (random; --lowc P mixes in a fraction P of poly-A / dinucleotide reads).
Run under `rocprofv3 --kernel-trace --stats` for the kernel time; prints the wall time of cfr_dust_mask_device (with copies)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from centrifuger_amd import capi

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=10_000_000)
ap.add_argument("--lowc", type=float, default=0.0)
ap.add_argument("--n-rate", type=float, default=0.0)
ap.add_argument("--check", type=int, default=20000, help="compare this many reads with the host twin")
a = ap.parse_args()
rng = np.random.default_rng(5)
n = a.reads
b = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n * 150)].copy()
if a.n_rate:
    b[rng.random(n * 150) < a.n_rate] = ord("N")
if a.lowc:
    v = b.reshape(n, 150)
    pick = np.nonzero(rng.random(n) < a.lowc)[0]
    half = len(pick) // 2
    v[pick[:half], 20:120] = ord("A")
    v[pick[half:], 10:140:2] = ord("C")
    v[pick[half:], 11:140:2] = ord("A")
o = np.arange(n + 1, dtype=np.uint64) * np.uint64(150)
idx = capi.Index(os.path.join(ROOT, "tests", "golden", "f6"), capi.default_params(max_result=3))
d = capi.DeviceIndex(idx)
for rep in range(3):
    x = b.copy()
    t0 = time.perf_counter()
    d.dust_mask(x, o)
    print(f"rep {rep}: cfr_dust_mask_device wall {time.perf_counter() - t0:.3f} s, masked bases {int((x != b).sum())}")
k = min(a.check, n)
h = b[: k * 150].copy()
capi.dust_mask(h, o[: k + 1].copy(), threads=8)
print("equals host twin on", k, "reads:", bool((h == x[: k * 150]).all()))
