#!/bin/bash
# does a rocprofv3 --pmc pass filtered on the post-stage kernels finish?  (it hung for 900 s inside bench.py, three times)
export CFR_DEBUG_ENV=1 CFR_SUBBATCH=2000000 CFR_TAPER_FLOOR=0 TMPDIR=/tmp
python bench.py --steps 1 --warmup 1 --no-pmc --no-extra-configs --no-40gbp --no-cpu-baseline > /dev/null 2>&1     # fills the cache
INNER="python $PWD/bench.py --inner --reads 2000000 --steps 1 --warmup 1 --no-cpu-baseline"
cd /tmp
try() {
  local label=$1; shift
  local t0=$(date +%s)
  rm -rf /tmp/pp
  timeout 240 env "$@" > /tmp/pp.log 2>&1
  echo "$label: rc $? in $(( $(date +%s) - t0 )) s; $(find /tmp/pp -name '*counter_collection.csv' 2>/dev/null | head -1 | xargs -r wc -l | cut -d' ' -f1) counter rows"
}
try "tail stream, 4 counters" X=1 rocprofv3 --pmc TCC_EA0_RDREQ_sum WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv --kernel-include-regex 'k_adjust_tail|k_post_fast|k_tail_heavy' -d /tmp/pp -- $INNER
try "main stream, 4 counters" CFR_TAIL_STREAM=0 rocprofv3 --pmc TCC_EA0_RDREQ_sum WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv --kernel-include-regex 'k_adjust_tail|k_post_fast|k_tail_heavy' -d /tmp/pp -- $INNER
try "main stream, 2 counters" CFR_TAIL_STREAM=0 rocprofv3 --pmc TCC_EA0_RDREQ_sum WRITE_SIZE --kernel-trace --output-format csv --kernel-include-regex 'k_adjust_tail|k_post_fast|k_tail_heavy' -d /tmp/pp -- $INNER
try "main stream, 2 counters, one kernel" CFR_TAIL_STREAM=0 rocprofv3 --pmc TCC_EA0_RDREQ_sum WRITE_SIZE --kernel-trace --output-format csv --kernel-include-regex 'k_adjust_tail' -d /tmp/pp -- $INNER
try "main stream, SQ counters" CFR_TAIL_STREAM=0 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv --kernel-include-regex 'k_adjust_tail|k_post_fast|k_tail_heavy' -d /tmp/pp -- $INNER
tail -3 /tmp/pp.log
