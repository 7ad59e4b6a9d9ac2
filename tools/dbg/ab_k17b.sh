#!/bin/bash
# K = 17 (8-byte entries) against the shipped K = 16 on the other workloads: long reads, 20 / 200 strains, 8 Gbp (36-bit image); same library, same box
export CFR_DEBUG_ENV=1
run() { python bench.py "$@" --no-cpu-baseline --no-pmc --no-extra-configs --steps 5 --warmup 2 2>/tmp/k17.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g reads/s  search %.2f ms  total %.2f ms  oracle %s' % (d['value'], d['stage_ms']['search_ms'], d['stage_ms']['total_ms'], d['parity'].get('equals_oracle')))" || tail -3 /tmp/k17.err; }
for w in "--mode long" "--workload strains20" "--workload strains200" "--index-gbp 8" "--index-gbp 8 --mode long" "--index-gbp 2.5"; do
  echo -n "K = 16 $w: "; run $w
  echo -n "K = 17 $w: "; CFR_FTABX_WIDTH=17 CFR_FTABX_E8=1 run $w
done
