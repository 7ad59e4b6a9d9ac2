#!/bin/bash
export CFR_DEBUG_ENV=1   # the gate behind which the library reads its CFR_* A/B switches
# One evidence pass on the GPU box: bench lines (SE with the CPU baseline + parity, PE, long), the kernel trace of the
# default bench command, and the PMC passes.  usage: tools/evidence.sh <tag>     -> gpurun_out/<tag>_*
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out
mkdir -p $O
cd $ROOT
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.log
python bench.py --mode pe --no-cpu-baseline > $O/${TAG}_bench_pe.json 2> $O/${TAG}_bench_pe.log
python bench.py --mode long --no-cpu-baseline > $O/${TAG}_bench_long.json 2> $O/${TAG}_bench_long.log
# kernel trace of the default command (no CPU baseline leg: it only adds host time)
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $O/${TAG}_trace -- python $ROOT/bench.py --no-cpu-baseline > $O/${TAG}_trace.json 2> $O/${TAG}_trace.log )
db=$(find $O/${TAG}_trace -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocpd_summary.py $db $O/${TAG}_kernel_trace_stats.txt
# PMC passes: one 2 M-read launch per kernel (CFR_SUBBATCH=2000000 CFR_TAPER_FLOOR=0: a single sub-batch, so per-launch counters divide by 2 M reads)
CFR_SUBBATCH=2000000 CFR_TAPER_FLOOR=0 tools/pmc_passes.sh $O/${TAG}_pmc > $O/${TAG}_pmc.log 2>&1
python tools/pmc_summary.py $O/${TAG}_pmc > $O/${TAG}_pmc_summary.txt 2>> $O/${TAG}_pmc.log
python tools/pmc_latest.py $O/${TAG}_pmc $TAG $O/${TAG}_pmc_latest.json >> $O/${TAG}_pmc.log 2>&1
tail -c 600 $O/${TAG}_bench.json; echo; head -12 $O/${TAG}_kernel_trace_stats.txt; cat $O/${TAG}_pmc_latest.json
