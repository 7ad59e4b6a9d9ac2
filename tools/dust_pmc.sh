#!/bin/bash
# PMC passes over k_dust alone (tools/dust_device_timing.py): tools/dust_pmc.sh <outdir> [script args]
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv --kernel-include-regex "k_dust" -d "$OUT/pmc$i" -- python $ROOT/tools/dust_device_timing.py --check 0 "$@" > /dev/null 2> "$OUT/pmc$i.log" || tail -3 "$OUT/pmc$i.log"
done
python $ROOT/tools/pmc_summary.py "$OUT" | grep -v "^#"
