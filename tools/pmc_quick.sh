#!/bin/bash
# two quick PMC passes (fabric requests, VALU/issue) for A/B of kernel variants: tools/pmc_quick.sh <outdir> [env assignments...]
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--reads 2000000 --steps 1 --warmup 1 --no-cpu-baseline"
i=0
for grp in "TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  env "$@" rocprofv3 --pmc $grp --kernel-trace --output-format csv --kernel-include-regex "k_search_chains" -d "$OUT/pmc$i" -- python $ROOT/bench.py $ARGS > /dev/null 2> "$OUT/pmc$i.log"
done
python $ROOT/tools/pmc_summary.py "$OUT" | grep -v "^#"
