#!/bin/bash
# round 5: k_post_fast (the common read's post stage in registers, under the search) against k_adjust_tail over all reads (CFR_POST_FAST=0),
# on one box, alternating; cfg2, pairs -k 5, 20 / 200 strains.  SETS="A=1 B=2|A=0" overrides the settings compared; WORK="cfg2 pe" the workloads.
export CFR_DEBUG_ENV=1
SETS=${SETS:-"CFR_POST_FAST=1|CFR_POST_FAST=0"}
WORK=${WORK:-"cfg2 pe strains20 strains200"}
run() { python bench.py "$@" --no-cpu-baseline --no-pmc --no-extra-configs --steps 8 --warmup 2 2>/tmp/ab_err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g /s  search %.2f ms  post %.2f ms  total %.2f ms  step %.2f ms' % (d['value'], d['stage_ms']['search_ms'], d['stage_ms']['tail_ms'], d['stage_ms']['total_ms'], d['ms_per_step']))" || tail -3 /tmp/ab_err.txt; }
for rep in 1 2; do
  IFS='|' read -ra LIST <<< "$SETS"
  for S in "${LIST[@]}"; do
    for w in $WORK; do
      case $w in cfg2) a="";; pe) a="--mode pe";; *) a="--workload $w";; esac
      echo -n "$S | $w: "
      env $S bash -c "$(declare -f run); run $a"
    done
  done
done
