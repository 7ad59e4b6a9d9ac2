#!/bin/bash
run() { python bench.py "$@" --no-cpu-baseline --no-pmc --no-extra-configs --steps 8 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g reads/s  step %.2f ms  search %.2f  device window %.2f' % (d['value'], d['ms_per_step'], d['stage_ms']['search_ms'], d['stage_ms']['total_ms']))"; }
export CFR_DEBUG_ENV=1
echo -n "packed up front:      "; run
echo -n "packed per sub-batch: "; CFR_PACK_PIECES=1 run
echo -n "packed up front:      "; run
echo -n "packed per sub-batch: "; CFR_PACK_PIECES=1 run
echo -n "pairs up front:       "; run --mode pe
echo -n "pairs per sub-batch:  "; CFR_PACK_PIECES=1 run --mode pe
