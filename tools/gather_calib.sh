#!/bin/bash
# Calibrates what one fabric read request of a RANDOM gather carries on this chip (MI355X_MICROARCH.md: FETCH_SIZE is
# calibrated for streaming reads only).  gather_bench touches a known number of 64-byte records (lanes x steps) of a table
# far larger than L2 + Infinity Cache; the PMC pass counts the requests that left L2.
# usage: tools/gather_calib.sh <outdir> [table_MB=32768]
set -u
OUT=$1; MB=${2:-32768}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
[ -x $ROOT/tools/gather_bench ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $ROOT/tools/gather_bench $ROOT/tools/gather_bench.hip
LANES=$((256*2048*8)); STEPS=50
for mode in 1 4 5 6 2; do
  $ROOT/tools/gather_bench $MB $mode $STEPS $LANES > "$OUT/mode$mode.txt" 2>&1
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_mode$mode" -- \
     $ROOT/tools/gather_bench $MB $mode $STEPS $LANES > "$OUT/pmc_mode$mode.log" 2>&1 || echo "mode $mode pmc failed" >> "$OUT/failed.txt"
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d "$OUT/pmc2_mode$mode" -- \
     $ROOT/tools/gather_bench $MB $mode $STEPS $LANES > "$OUT/pmc2_mode$mode.log" 2>&1 || true
done
python3 $ROOT/tools/gather_calib.py "$OUT" $LANES $STEPS | tee "$OUT/summary.txt"
