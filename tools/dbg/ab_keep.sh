#!/bin/bash
# A/B of the text step's survivor selection in k_search_chains_v2: the shipped build (selects, no scratch memory) against the form before
# (tools/dbg/libcfr_hip_keepchain.so, -DCFR_TEXT_KEEP_CHAIN=1: an if-chain the compiler turned into a four-entry scratch array, 12 scratch ops per text step)
run() { python bench.py "$@" --no-cpu-baseline --no-pmc --no-extra-configs --steps 6 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g reads/s  search %.2f ms  total %.2f ms' % (d['value'], d['stage_ms']['search_ms'], d['stage_ms']['total_ms']))"; }
legs() {
  echo -n "cfg2 se: "; run; echo -n "cfg3 pe: "; run --mode pe; echo -n "long: "; run --mode long
  echo -n "strains20: "; run --workload strains20
  echo -n "8 Gbp (wide kernel): "; run --index-gbp 8
}
echo "== shipped (selects)"; legs
cp centrifuger_amd/libcfr_hip.so /tmp/shipped.so; cp tools/dbg/libcfr_hip_keepchain.so centrifuger_amd/libcfr_hip.so
echo "== before (if-chain -> scratch array)"; legs
cp /tmp/shipped.so centrifuger_amd/libcfr_hip.so
echo "== shipped once more"; echo -n "cfg2 se: "; run; echo -n "8 Gbp: "; run --index-gbp 8
