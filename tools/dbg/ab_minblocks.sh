#!/bin/bash
# A/B of the search kernel built for 5 blocks per CU (shipped) against the unconstrained build (tools/dbg/libcfr_hip_mb1.so)
set -e
run() { python bench.py "$@" --no-cpu-baseline --no-pmc --no-extra-configs --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms']['search_ms'], d['stage_ms']['tail_ms'])"; }
echo "== shipped (min blocks 5)"; echo -n "cfg2 se: "; run; echo -n "cfg3 pe: "; run --mode pe; echo -n "8 Gbp lean: "; CFR_DEBUG_ENV=1 CFR_FTABX_E8=1 CFR_LOC_MEMO_GB=0 run --index-gbp 8; echo -n "long: "; run --mode long
cp centrifuger_amd/libcfr_hip.so /tmp/shipped.so; cp tools/dbg/libcfr_hip_mb1.so centrifuger_amd/libcfr_hip.so
echo "== unconstrained (min blocks 1)"; echo -n "cfg2 se: "; run; echo -n "cfg3 pe: "; run --mode pe; echo -n "8 Gbp lean: "; CFR_DEBUG_ENV=1 CFR_FTABX_E8=1 CFR_LOC_MEMO_GB=0 run --index-gbp 8; echo -n "long: "; run --mode long
cp /tmp/shipped.so centrifuger_amd/libcfr_hip.so
