#!/bin/bash
# NOTE: the switches below (CFR_PRM, CFR_PRM_CHARS) exist in the library of commit e774bae only - the shipped kernel reads its chains' offsets from the arrays (profiles/HISTORY.md section 9)
# round 5: the chains' parameter records with the strands' first 32 characters (a chain looks up its first K-mer in the iteration it is
# taken in) | records without the characters (CFR_PRM_CHARS=0) | no records (CFR_PRM=0: offsets from the arrays), alternating on one box;
# the iteration mix of the search for the first and the last setting.
export CFR_DEBUG_ENV=1
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
for S in "CFR_PRM=1" "CFR_PRM=0"; do
  echo "== iteration mix, $S"
  env $S CFR_SEARCH_PROF=1 python bench.py --no-cpu-baseline --no-pmc --no-extra-configs --steps 1 --warmup 0 2>&1 >/dev/null | grep "search prof" | head -1
done
SETS="CFR_PRM=1|CFR_PRM_CHARS=0|CFR_PRM=0" WORK="${WORK:-cfg2 pe strains20}" tools/dbg/ab_post_fast.sh
