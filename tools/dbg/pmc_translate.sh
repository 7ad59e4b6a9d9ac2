#!/bin/bash
# SQ counters of k_translate_prot / k_search_prot_sm (one protein bench step under rocprofv3 --pmc, two passes)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp; export TMPDIR=/tmp
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"; do
  rm -rf /tmp/pmct
  rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmct -o out --output-format csv -- python $ROOT/bench.py --mode protein --inner --no-pmc --no-cpu-baseline --steps 1 --warmup 0 > /dev/null 2>/tmp/pmct.err
  f=$(find /tmp/pmct -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in rows:
    k=r["Kernel_Name"]
    if "k_translate_prot" in k or "k_search_prot" in k:
        name=k.split("(")[0][-40:]
        acc[name][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in acc.items():
    print(k, {c: "%.4g"%x for c,x in v.items()})
PY
done
