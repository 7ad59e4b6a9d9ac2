#!/bin/bash
# profiles/HISTORY.md section 8 item 0: the cheap probes of rounds 3 / 4 once more on the search kernel WITHOUT its scratch array (cfg2; "pe" = pairs -k 5)
export CFR_DEBUG_ENV=1
run() { python bench.py "$@" --no-cpu-baseline --no-pmc --no-extra-configs --steps 6 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g /s  search %.2f ms  post %.2f ms  total %.2f ms  step %.2f ms' % (d['value'], d['stage_ms']['search_ms'], d['stage_ms']['tail_ms'], d['stage_ms']['total_ms'], d['ms_per_step']))"; }
probe() { echo -n "$1 | cfg2: "; env $1 bash -c "$(declare -f run); run"; }
probe "X=0"
probe "CFR_BLOCKS_PER_CU=5"
probe "CFR_BLOCKS_PER_CU=3"
probe "CFR_TAIL_STREAM=0"
probe "CFR_TAIL_STREAM=0 CFR_BLOCKS_PER_CU=5"
probe "CFR_TAIL_BLOCKS=1"
probe "CFR_TAIL_BLOCKS=2"
probe "CFR_TAIL_BLOCKS=4"
probe "CFR_SUBBATCH=2500000"
probe "CFR_SUBBATCH=833334"
probe "X=0"
cp centrifuger_amd/libcfr_hip.so /tmp/shipped.so
for v in nt nt2; do
  cp tools/dbg/libcfr_hip_$v.so centrifuger_amd/libcfr_hip.so
  echo -n "gathers non-temporal ($v: 1 = all four slots, 2 = the table gather only) | cfg2: "; run
  echo -n "   the same with PMC (value, requests/read, frac): "; python bench.py --no-cpu-baseline --no-extra-configs --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], r.get('fabric_read_requests_per_read'), r.get('frac'), r.get('kernel_ms_alone'))"
done
cp /tmp/shipped.so centrifuger_amd/libcfr_hip.so
echo -n "shipped with PMC (value, requests/read, frac, alone ms): "; python bench.py --no-cpu-baseline --no-extra-configs --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], r.get('fabric_read_requests_per_read'), r.get('frac'), r.get('kernel_ms_alone'))"
