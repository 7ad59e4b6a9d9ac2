#!/bin/bash
# the K = 17 policy at the sizes around its memory limit: 8 Gbp (K = 17), 12 and 16 Gbp (which table is chosen, and that a 10 M-read batch still has its scratch)
export CFR_DEBUG_ENV=1
run() { python bench.py "$@" --no-cpu-baseline --no-pmc --no-extra-configs --steps 5 --warmup 2 2>/tmp/k.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g reads/s  step %.2f ms  search %.2f ms  oracle %s' % (d['value'], d['ms_per_step'], d['stage_ms']['search_ms'], d['parity'].get('equals_oracle')))" || tail -2 /tmp/k.err; grep "device image" /tmp/k.err | cut -c1-120; }
for g in 8 12 16; do echo "== $g Gbp, policy:"; run --index-gbp $g; done
echo "== 16 Gbp long reads, policy:"; run --index-gbp 16 --mode long
echo "== 12 Gbp K = 16:"; CFR_K17=0 run --index-gbp 12
