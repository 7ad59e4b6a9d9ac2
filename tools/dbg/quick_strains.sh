#!/bin/bash
# the two strain workloads and cfg2, plain steps only: value, step, search and post-stage ms per 10 M reads
run() { python bench.py "$@" --no-cpu-baseline --no-pmc --no-extra-configs --steps 5 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g reads/s  step %.2f ms  search %.2f  tail %.2f  oracle %s' % (d['value'], d['ms_per_step'], d['stage_ms']['search_ms'], d['stage_ms']['tail_ms'], d['parity_oracle']['equals_oracle']))"; }
echo -n "cfg2:       "; run
echo -n "strains20:  "; run --workload strains20
echo -n "strains200: "; run --workload strains200
echo -n "pairs:      "; run --mode pe
