#!/bin/bash
# A/B of k_tail_heavy's best / second bookkeeping: selects (shipped, no scratch instructions) against the branches the compiler had
# turned into two-entry scratch arrays (tools/dbg/libcfr_hip_teambr.so, -DCFR_TEAM_BEST_BRANCHES=1), on the strain workloads
run() { python bench.py "$@" --no-cpu-baseline --no-pmc --no-extra-configs --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g reads/s  search %.2f ms  post %.2f ms  total %.2f ms' % (d['value'], d['stage_ms']['search_ms'], d['stage_ms']['tail_ms'], d['stage_ms']['total_ms']))"; }
legs() { echo -n "strains20: "; run --workload strains20; echo -n "strains200: "; run --workload strains200; echo -n "cfg2: "; run; }
echo "== shipped (selects)"; legs
cp centrifuger_amd/libcfr_hip.so /tmp/shipped.so; cp tools/dbg/libcfr_hip_teambr.so centrifuger_amd/libcfr_hip.so
echo "== before (branches -> scratch arrays)"; legs
cp /tmp/shipped.so centrifuger_amd/libcfr_hip.so
echo "== shipped once more"; legs
