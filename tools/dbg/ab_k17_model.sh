#!/bin/bash
# VERDICT r4 #5, the benefit side of "K = 17 at 40 Gbp" measured on a scaled model: the lean image of 40 Gbp (36-bit suffix array, 8-byte
# K-mer entries, no locate memo) forced on a 2.5 Gbp index, where K = 14 leaves 2.5e9 / 4^14 = 9.3 expected rows per random K-mer - what
# K = 16 leaves at 40 Gbp - and K = 15 leaves 2.3 - what K = 17 would.  150 bp reads (cfg4's) and long reads (cfg5's); iteration mix of both.
export CFR_DEBUG_ENV=1 CFR_FORCE_WIDE=1 CFR_FTABX_E8=1 CFR_LOC_MEMO_GB=0
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
run() { python bench.py "$@" --index-gbp 2.5 --no-cpu-baseline --no-pmc --no-extra-configs --steps 6 --warmup 2 2>/tmp/k17_err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g /s  search %.2f ms  post %.2f ms  step %.2f ms' % (d['value'], d['stage_ms']['search_ms'], d['stage_ms']['tail_ms'], d['ms_per_step']))" || tail -3 /tmp/k17_err.txt; }
for rep in 1 2; do
  for K in 14 15; do
    echo -n "K = $K | 150 bp: "; CFR_FTABX_WIDTH=$K bash -c "$(declare -f run); run"
    echo -n "K = $K | long reads: "; CFR_FTABX_WIDTH=$K bash -c "$(declare -f run); run --mode long"
  done
done
for K in 14 15; do
  echo "== iteration mix, K = $K"
  CFR_FTABX_WIDTH=$K CFR_SEARCH_PROF=1 python bench.py --index-gbp 2.5 --no-cpu-baseline --no-pmc --no-extra-configs --steps 1 --warmup 0 2>&1 >/dev/null | grep "search prof" | head -1
done
