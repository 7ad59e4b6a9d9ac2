"""debug: the fresh-index PE workload of tests/test_gpu_parity.py under different switch sets; prints reads whose results differ"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["CFR_DEBUG_ENV"] = "1"
from centrifuger_amd import capi, synth
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
tmp = tempfile.mkdtemp()
g = synth.make_genomes(n_species=6, n_strains=4, genome_len=125000, seed=77)
synth.write_reference_inputs(g, tmp)
prefix = os.path.join(tmp, "idx")
subprocess.run([os.path.join(REF_DIR, "centrifuger-build"), "-t", "8", "-r", os.path.join(tmp, "ref.fa"), "--taxonomy-tree", os.path.join(tmp, "nodes.dmp"),
                "--name-table", os.path.join(tmp, "names.dmp"), "--conversion-table", os.path.join(tmp, "seqid.map"), "-o", prefix], check=True, stderr=subprocess.DEVNULL)
n = 20000
r1, r2 = synth.make_pairs(g, n, 150, seed=78)
b1, o1, b2, o2 = r1.bases.copy(), r1.offsets, r2.bases.copy(), r2.offsets
capi.dust_mask(b1, o1, threads=4); capi.dust_mask(b2, o2, threads=4)
def run(env):
    for k, v in env.items(): os.environ[k] = v
    idx = capi.Index(prefix, capi.default_params(max_result=5))
    d = capi.DeviceIndex(idx)
    res, mat = d.classify(b1, o1, b2, o2)
    res = res.copy(); mat = mat.copy()
    d.close()
    for k in env: os.environ.pop(k)
    return res, mat
base = run({"CFR_TEXT_MODE": "0"})
for name, env in [("default", {}), ("default again", {}), ("v1", {"CFR_SEARCH_V1": "1"}), ("wide", {"CFR_FORCE_WIDE": "1"})]:
    res, mat = run(env)
    bad = [i for i in range(n) if res[i].tobytes() != base[0][i].tobytes() or mat[5*i:5*i+5].tobytes() != base[1][5*i:5*i+5].tobytes()]
    print(name, "differing reads:", len(bad), bad[:10])
    for i in bad[:3]:
        print("  base", base[0][i], base[1][5*i:5*i+max(1,base[0][i]["n_match"])])
        print("  this", res[i], mat[5*i:5*i+max(1,res[i]["n_match"])])
