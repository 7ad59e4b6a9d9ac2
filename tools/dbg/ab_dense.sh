#!/bin/bash
# NOTE: the switches below (CFR_DENSE, CFR_PRM) exist in the library of commit e774bae only - the shipped kernel has neither form (profiles/HISTORY.md section 9)
# round 5: the dense read form (64 characters per 16-byte load) and the chains' parameter records (one gather per chain start instead of two)
# against the build without them, on one box, alternating: CFR_DENSE=1 (both, the default) | CFR_DENSE=0 (records only) | CFR_PRM=0 (neither).
# Also the iteration mix of the search (CFR_SEARCH_PROF=1: block loads per read) for the first and the last setting.
export CFR_DEBUG_ENV=1
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
for S in "CFR_DENSE=1" "CFR_PRM=0"; do
  echo "== iteration mix, $S"
  env $S CFR_SEARCH_PROF=1 python bench.py --no-cpu-baseline --no-pmc --no-extra-configs --steps 1 --warmup 0 2>&1 >/dev/null | grep "search prof" | head -2
done
SETS="CFR_DENSE=1|CFR_DENSE=0|CFR_PRM=0" WORK="${WORK:-cfg2 pe strains20}" tools/dbg/ab_post_fast.sh
