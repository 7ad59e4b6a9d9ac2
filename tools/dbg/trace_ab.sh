#!/bin/bash
# kernel trace of one workload under two settings: tools/dbg/trace_ab.sh <tag> "<bench args>" "SET1" "SET2" ...  -> gpurun_out/<tag>_<n>.txt
export CFR_DEBUG_ENV=1
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$ROOT/gpurun_out; mkdir -p $O
TAG=$1; ARGS=$2; shift 2
n=0
for S in "$@"; do
  n=$((n+1))
  ( cd /tmp && export TMPDIR=/tmp && env $S rocprofv3 --kernel-trace --stats -d /tmp/tr_$TAG_$n -- python $ROOT/bench.py --inner --no-cpu-baseline --no-pmc --no-extra-configs --steps 4 --warmup 1 $ARGS > /tmp/tr_$n.json 2> /tmp/tr_$n.log )
  db=$(find /tmp/tr_$TAG_$n -name "*.db" | head -1)
  echo "== $S" > $O/${TAG}_$n.txt
  python $ROOT/tools/rocpd_summary.py $db | head -16 >> $O/${TAG}_$n.txt
  cat $O/${TAG}_$n.txt | cut -c1-230
  rm -rf /tmp/tr_$TAG_$n
done
