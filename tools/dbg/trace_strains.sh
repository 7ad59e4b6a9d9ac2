#!/bin/bash
# kernel trace of one strain workload's steps under a few switch settings: tools/dbg/trace_strains.sh [workload]
W=${1:-strains200}
mkdir -p gpurun_out/r4n; O=$PWD/gpurun_out/r4n; R=$PWD
export CFR_DEBUG_ENV=1
python bench.py --workload $W --inner --no-cpu-baseline --no-pmc --no-extra-configs --steps 1 --warmup 1 > /dev/null 2>&1   # builds the index cache
cd /tmp && export TMPDIR=/tmp
one() {   # tag, env...
  tag=$1; shift
  env "$@" rocprofv3 --kernel-trace --stats -d $O/trace_$tag -- python $R/bench.py --workload $W --inner --no-cpu-baseline --no-pmc --no-extra-configs --steps 3 --warmup 2 > $O/trace_$tag.json 2> $O/trace_$tag.log
  db=$(find $O/trace_$tag -name "*.db" | head -1)
  echo "=== $W $tag ($*)"
  python $R/tools/trace_reconcile.py $db $O/trace_$tag.json $O/${W}_trace_$tag.txt | grep -E "^#|k_search|k_tail|k_adjust" | cut -c1-150
  rm -rf $O/trace_$tag
}
one behind CFR_TAIL_STREAM=0
one beside CFR_TAIL_STREAM=1
one behind_direct64 CFR_TAIL_STREAM=0 CFR_HEAVY_DIRECT_ROWS=64
one beside_direct64 CFR_TAIL_STREAM=1 CFR_HEAVY_DIRECT_ROWS=64
