#!/usr/bin/env python3
"""Host SDUST throughput on this box (cfr_dust_mask_batch), random 150 bp reads with 0.1 % N."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centrifuger_amd import capi

rng = np.random.default_rng(3)
n = 2_000_000
b = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n * 150)].copy()
b[rng.random(n * 150) < 0.001] = ord("N")
o = (np.arange(n + 1, dtype=np.uint64) * np.uint64(150))
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for thr in (1, 4, 16, 64, 128):
    x = b.copy()
    t0 = time.perf_counter()
    capi.dust_mask(x, o, threads=thr)
    dt = time.perf_counter() - t0
    print(f"threads {thr:3d}: {n/dt/1e6:7.2f} M reads/s, {dt*1e9/(n*150)*thr:6.1f} ns/base/thread, wall {dt:.3f} s")
