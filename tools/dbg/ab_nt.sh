#!/bin/bash
# A/B of the search kernel's gather loads with the non-temporal hint (tools/dbg/libcfr_hip_nt.so, -DCFR_GATHER_NT=1) against the shipped build
set -e
run() { python bench.py "$@" --no-cpu-baseline --no-pmc --no-extra-configs --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms']['search_ms'], d['stage_ms']['tail_ms'])"; }
pmc() { python bench.py "$@" --no-cpu-baseline --no-extra-configs --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], r.get('fabric_read_requests_per_read'), r.get('frac'), r.get('kernel_ms'))"; }
legs() {
  echo -n "cfg2 se: "; run; echo -n "cfg3 pe: "; run --mode pe
  if [ -z "$QUICK" ]; then echo -n "strains20: "; run --workload strains20; echo -n "long: "; run --mode long; fi
  echo -n "8 Gbp lean: "; CFR_DEBUG_ENV=1 CFR_FTABX_E8=1 CFR_LOC_MEMO_GB=0 run --index-gbp 8
  echo -n "cfg2 se with PMC (value, requests/read, frac, kernel ms): "; pmc
}
if [ -z "$SKIP_SHIPPED" ]; then echo "== shipped"; legs; fi
cp centrifuger_amd/libcfr_hip.so /tmp/shipped.so; cp tools/dbg/libcfr_hip_nt.so centrifuger_amd/libcfr_hip.so
echo "== gathers non-temporal"; legs
cp /tmp/shipped.so centrifuger_amd/libcfr_hip.so
