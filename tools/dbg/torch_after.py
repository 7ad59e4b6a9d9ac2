import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["CFR_DEBUG_ENV"] = "1"
import torch
from centrifuger_amd import capi
G = os.path.join(ROOT, "tests", "golden")
mode = sys.argv[1]
idx = capi.Index(os.path.join(G, "f6"))
dev = capi.DeviceIndex(idx)
import numpy as np
import oracle_lib as ora
ids, b, o = ora.read_fastx(os.path.join(G, "se.fq"))
if mode == "classify":
    dev.classify(b, o)
if mode == "selfcheck":
    print(dev.selfcheck())
if mode == "rank":
    dev.rank(np.full(10, ord("A"), dtype=np.uint8), np.arange(10, dtype=np.uint64), np.ones(10, dtype=np.uint8))
print([l.split()[-1] for l in open("/proc/self/maps") if "amdhip" in l and "r-xp" in l])
try:
    print(torch.zeros(3).cuda().sum().item(), "torch ok after", mode)
except Exception as e:
    print("torch FAILED after", mode, repr(e)[:200])
