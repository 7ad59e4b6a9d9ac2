#!/bin/bash
# strain workloads: search blocks per CU beside the post stage (4 = default when overlapping, 3 leaves room for a post-stage wave per SIMD)
run() { python bench.py "$@" --no-cpu-baseline --no-pmc --no-extra-configs --steps 5 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g reads/s  step %.2f ms  search %.2f  tail %.2f' % (d['value'], d['ms_per_step'], d['stage_ms']['search_ms'], d['stage_ms']['tail_ms']))"; }
export CFR_DEBUG_ENV=1
for w in strains20 strains200; do
  echo "== $w"
  echo -n "default (4 blocks/CU beside): "; run --workload $w
  echo -n "3 blocks/CU beside:           "; CFR_BLOCKS_PER_CU=3 run --workload $w
  echo -n "2 blocks/CU beside:           "; CFR_BLOCKS_PER_CU=2 run --workload $w
  echo -n "behind:                       "; CFR_TAIL_STREAM=0 run --workload $w
done
