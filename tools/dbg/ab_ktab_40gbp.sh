#!/bin/bash
# BASELINE configs[3] / configs[4] on one GPU (40 Gbp index, written once by the native writer) with the K-mer count table off (CFR_KTAB=0) / beside the
# K-mer table (1) / with the K-mer table freed after the build (2) - same library, same box
export CFR_DEBUG_ENV=1 CFR_LOAD_TIMING=1
run() { python bench.py "$@" --no-cpu-baseline --no-pmc 2>/tmp/ab_ktab.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g reads/s  step %.2f ms  search %.2f ms  oracle %s' % (d['value'], d['ms_per_step'], d['stage_ms']['search_ms'], d['parity'].get('equals_oracle')))" || tail -5 /tmp/ab_ktab.err; grep -E "\[ktab\]|count table:|device image|no room" /tmp/ab_ktab.err | head -5; }
for m in 0 1 2 0 2; do
  echo "== CFR_KTAB=$m cfg4"; CFR_KTAB=$m CFR_KTAB_CHECK=1 run --config cfg4
done
for m in 0 1 2; do
  echo "== CFR_KTAB=$m cfg5"; CFR_KTAB=$m run --config cfg5
done
