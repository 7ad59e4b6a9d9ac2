#!/bin/bash
export CFR_DEBUG_ENV=1
for m in 0 1; do
  echo "== CFR_KTAB=$m cfg4 with PMC"
  CFR_KTAB=$m CFR_BENCH_FULL_LINE=1 python bench.py --config cfg4 --no-cpu-baseline 2>/tmp/pmc40.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value %.4g  step %.2f ms  search %.2f ms' % (d['value'], d['ms_per_step'], d['stage_ms']['search_ms']))
print({k: r.get(k) for k in ('fabric_read_requests_per_read','read_bytes_per_read','l2_hit','kernel_ms_alone','frac_counter_traffic','fetched_over_useful')})
print('gather', r.get('gather'))
print('instr', {k:v for k,v in (r.get('instruction_stream') or {}).items() if k!='note'})
print('mix', r.get('iteration_mix_per_read'))
"
  tail -3 /tmp/pmc40.err | cut -c1-300
done
