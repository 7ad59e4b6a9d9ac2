#!/bin/bash
# The post stage of sub-batch k beside the search of k + 1 (CFR_TAIL_STREAM=1), how many blocks the search keeps per CU, how many the post
# stage gets, and whether handing the search's chains out dynamically (eight per draw) helps: value, step, search and post-stage ms per
# 10 M reads, cfg2 and pairs.
run() { python bench.py "$@" --no-cpu-baseline --no-pmc --no-extra-configs --steps 8 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g reads/s  step %.2f ms  search %.2f  tail %.2f' % (d['value'], d['ms_per_step'], d['stage_ms']['search_ms'], d['stage_ms']['tail_ms']))"; }
export CFR_DEBUG_ENV=1
for mode in "" "--mode pe"; do
  echo "== $mode"
  echo -n "behind (TAIL_STREAM=0):              "; CFR_TAIL_STREAM=0 run $mode
  echo -n "beside, 5 blocks/CU:                 "; CFR_TAIL_STREAM=1 CFR_BLOCKS_PER_CU=5 run $mode
  echo -n "beside, 4 blocks/CU:                 "; CFR_TAIL_STREAM=1 CFR_BLOCKS_PER_CU=4 run $mode
  echo -n "beside, 3 blocks/CU:                 "; CFR_TAIL_STREAM=1 CFR_BLOCKS_PER_CU=3 run $mode
  echo -n "beside, 4 blocks/CU, tail 1 block/CU: "; CFR_TAIL_STREAM=1 CFR_BLOCKS_PER_CU=4 CFR_TAIL_BLOCKS=1 run $mode
  echo -n "beside, 4 blocks/CU, tail 2 blocks/CU:"; CFR_TAIL_STREAM=1 CFR_BLOCKS_PER_CU=4 CFR_TAIL_BLOCKS=2 run $mode
  echo -n "beside, 4 blocks/CU, dyn (8 per draw):"; CFR_TAIL_STREAM=1 CFR_BLOCKS_PER_CU=4 CFR_SEARCH_DYN=1 run $mode
  echo -n "behind, dyn (8 per draw):            "; CFR_TAIL_STREAM=0 CFR_SEARCH_DYN=1 run $mode
done
