#!/bin/bash
# A/B of the two-launch search (k_search_chains_v2 STAGE 1 without wide text mode + STAGE 2 over the list, CFR_SEARCH_SPLIT=1) against the one-launch form, same library, same box, alternating
export CFR_DEBUG_ENV=1
run() { python bench.py "$@" --no-cpu-baseline --no-pmc --no-extra-configs --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g reads/s  search %.2f ms  total %.2f ms  oracle %s' % (d['value'], d['stage_ms']['search_ms'], d['stage_ms']['total_ms'], d['parity'].get('equals_oracle')))"; }
legs() {
  echo -n "cfg2 se: "; run; echo -n "cfg3 pe: "; run --mode pe; echo -n "long: "; run --mode long
  echo -n "strains20: "; run --workload strains20
  echo -n "8 Gbp (wide kernel): "; run --index-gbp 8
}
for rep in 1 2; do
  echo "== one launch"; CFR_SEARCH_SPLIT=0 legs
  echo "== two launches (stage 1 without wide text mode + stage 2 over the list)"; CFR_SEARCH_SPLIT=1 legs
done
echo "== iteration mix, one launch"; CFR_SEARCH_SPLIT=0 CFR_SEARCH_PROF=1 python bench.py --no-cpu-baseline --no-pmc --no-extra-configs --steps 1 --warmup 0 2>&1 | grep "search prof" | head -3
echo "== iteration mix, two launches"; CFR_SEARCH_SPLIT=1 CFR_SEARCH_PROF=1 python bench.py --no-cpu-baseline --no-pmc --no-extra-configs --steps 1 --warmup 0 2>&1 | grep "search prof" | head -6
