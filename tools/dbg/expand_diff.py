"""which rows of an --expand-taxid golden case differ on the device (debug helper): python tools/dbg/expand_diff.py x8.pe_k1_expand"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
os.environ.setdefault("CFR_DEBUG_ENV", "1")
from centrifuger_amd import capi
from test_host_expand_cpu import case_params, load_reads
from test_oracle_golden_expand import EXP, MAN
from conftest import GOLDEN
case = sys.argv[1]
c = MAN["cases"][case]
kw = case_params(c["args"])
gd = GOLDEN
prefix = os.path.join(EXP if c["index"] == "x8" else gd, c["index"])
idx = capi.Index(prefix, capi.default_params(output_expanded=1, max_result=kw.get("max_result", 1)))
dev = capi.DeviceIndex(idx)
ids, b1, o1, b2, o2 = load_reads(c["args"], gd)
dev.set_dust("--no-dust" not in c["args"])
r, m, s, x = dev.classify_expanded(b1, o1, b2, o2)
got = (capi.lib().cfr_tsv_header_expanded() + b"".join(idx.format_tsv_expanded(ids[i], r[i], m, s, x) for i in range(len(ids)))).split(b"\n")
want = open(os.path.join(EXP, "tsv", case + ".tsv"), "rb").read().split(b"\n")
nd = 0
for a, b in zip(got, want):
    if a != b:
        print("got ", a.decode()); print("want", b.decode()); nd += 1
        if nd > 12: break
print(len(got), len(want), nd)
