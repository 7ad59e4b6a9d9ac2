#!/bin/bash
# kernel trace of the SDUST pre-step (bench.py --inner-dust) with the screen off / on
export CFR_DEBUG_ENV=1
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for m in 0 1; do
  ( cd /tmp && export TMPDIR=/tmp && CFR_DUST_SCREEN=$m rocprofv3 --kernel-trace --stats -d /tmp/trd_$m -- python $ROOT/bench.py --inner --inner-dust --no-cpu-baseline --no-pmc --no-extra-configs --steps 4 --warmup 1 > /tmp/trd_$m.json 2> /tmp/trd_$m.log )
  db=$(find /tmp/trd_$m -name "*.db" | head -1)
  echo "== CFR_DUST_SCREEN=$m"
  python $ROOT/tools/rocpd_summary.py $db | grep -E "calls|k_dust|k_search_chains|k_pack|k_adjust" | cut -c1-200
  rm -rf /tmp/trd_$m
done
