#!/bin/bash
# PMC counter passes for the path kernels (separate rocprofv3 runs per counter group, kernel-trace only).
# usage: tools/pmc_passes.sh <outdir> [bench args...]
set -u
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--reads 2000000 --steps 1 --warmup 1 --no-cpu-baseline $*"
# warm the index cache (not profiled)
python $ROOT/bench.py $ARGS > "$OUT/bench_plain.json" 2> "$OUT/bench_plain.log"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv --kernel-include-regex "cfr::" -d "$OUT/pmc$i" -- python $ROOT/bench.py $ARGS > /dev/null 2> "$OUT/pmc$i.log" || echo "pass $i ($grp) failed" >> "$OUT/failed.txt"
done
find "$OUT" -name "*counter_collection.csv" | head
