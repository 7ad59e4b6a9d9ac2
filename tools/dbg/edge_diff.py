import os, sys, gzip, shutil, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["CFR_DEBUG_ENV"] = "1"
import numpy as np
import oracle_lib as ora
from centrifuger_amd import capi
G = os.path.join(ROOT, "tests", "golden")
tmp = tempfile.mkdtemp()
for f in ("f10.2.cfr", "f10.4.cfr"): shutil.copy(os.path.join(G, f), tmp)
with gzip.open(os.path.join(G, "f10.1.cfr.gz"), "rb") as fi, open(os.path.join(tmp, "f10.1.cfr"), "wb") as fo: shutil.copyfileobj(fi, fo)
prefix = os.path.join(tmp, "f10")
ids, b, o = ora.read_fastx(os.path.join(G, "edge.fa"))
capi.dust_mask(b, o)
idx = capi.Index(prefix); dev = capi.DeviceIndex(idx)
res, mat = dev.classify(b, o)
got = (capi.tsv_header() + b"".join(idx.format_tsv(ids[i], res[i], mat) for i in range(len(ids)))).split(b"\n")
want = open(os.path.join(G, "tsv", "f10.edge_default.tsv"), "rb").read().split(b"\n")
for a, w in zip(got, want):
    if a != w: print("GOT ", a); print("WANT", w)
hits, hb = dev.search(b, o)
oi = ora.OracleIndex(prefix)
for i in range(len(ids)):
    r = bytes(b[int(o[i]):int(o[i+1])])
    w = oi.query_hits(r)
    g = hits[int(hb[i]):int(hb[i+1])]
    if len(g) != len(w) or any(not np.array_equal(g[f], w[f]) for f in ("sp","ep","l","strand","offset")):
        print(ids[i], "hits differ"); print(" got ", [(int(h["sp"]),int(h["ep"]),int(h["l"]),int(h["strand"]),int(h["offset"])) for h in g]); print(" want", [(int(h["sp"]),int(h["ep"]),int(h["l"]),int(h["strand"]),int(h["offset"])) for h in w])
print(dev.info().n, dev.info().first_isa, os.environ.get("CFR_WIDE_ROWS"))
