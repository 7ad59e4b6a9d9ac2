#!/bin/bash
# A/B of the K-mer count table on a 36-bit image: CFR_KTAB=0 (none) / 1 (beside the K-mer table) / 2 (the K-mer table freed after the build); same library, same box
# usage: tools/dbg/ab_ktab.sh <index-gbp> [extra bench args]
export CFR_DEBUG_ENV=1 CFR_LOAD_TIMING=1
GBP=$1; shift
run() { python bench.py --index-gbp $GBP "$@" --no-cpu-baseline --no-pmc --no-extra-configs --steps 6 --warmup 2 2>/tmp/ab_ktab.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g reads/s  search %.2f ms  total %.2f ms  oracle %s' % (d['value'], d['stage_ms']['search_ms'], d['stage_ms']['total_ms'], d['parity'].get('equals_oracle')))"; grep -E "\[ktab\]|count table|device image" /tmp/ab_ktab.err | head -4; }
for rep in 1 2; do
  for m in 0 1 2; do
    echo "== CFR_KTAB=$m  se 150 bp"; CFR_KTAB=$m CFR_KTAB_CHECK=$((rep==1)) run "$@"
    echo "== CFR_KTAB=$m  long reads"; CFR_KTAB=$m run --mode long "$@"
  done
done
