#!/bin/bash
# pairs -k 5: the post stage (23.6 ms summed inside the step) is longer than the searches (16.8 ms) - how the wave slots are split between them
export CFR_DEBUG_ENV=1
run() { CFR_BENCH_FULL_LINE=1 python bench.py --mode pe "$@" --no-cpu-baseline --no-pmc --no-extra-configs --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('%.4g pairs/s  step %.2f ms  search %.2f  post %.2f  device total %.2f  oracle %s' % (d['value'], d['ms_per_step'], s['search_ms'], s['tail_ms'], s['total_ms'], d['parity_oracle']['equals_oracle']))"; }
echo -n "default: "; run
for b in 3 4 5; do echo -n "CFR_BLOCKS_PER_CU=$b: "; CFR_BLOCKS_PER_CU=$b run; done
echo -n "CFR_TAIL_PRIO=1: "; CFR_TAIL_PRIO=1 run
echo -n "CFR_TAIL_PRIO=1 CFR_BLOCKS_PER_CU=3: "; CFR_TAIL_PRIO=1 CFR_BLOCKS_PER_CU=3 run
echo -n "CFR_TAIL_STREAM=0 (post after its search, no overlap): "; CFR_TAIL_STREAM=0 run
echo -n "CFR_TAIL_STREAM=0 CFR_BLOCKS_PER_CU=5: "; CFR_TAIL_STREAM=0 CFR_BLOCKS_PER_CU=5 run
echo -n "CFR_SUBBATCH=625000: "; CFR_SUBBATCH=625000 run
echo -n "CFR_SUBBATCH=2500000: "; CFR_SUBBATCH=2500000 run
echo -n "default again: "; run
