#!/bin/bash
run() { python bench.py "$@" --no-cpu-baseline --no-pmc --no-extra-configs --steps 8 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g reads/s  step %.2f ms  search %.2f  device window %.2f' % (d['value'], d['ms_per_step'], d['stage_ms']['search_ms'], d['stage_ms']['total_ms']))"; }
export CFR_DEBUG_ENV=1
echo -n "split (CFR_PACK_SPLIT=1): "; CFR_PACK_SPLIT=1 run
echo -n "all up front (default): "; run
echo -n "split (CFR_PACK_SPLIT=1): "; CFR_PACK_SPLIT=1 run
echo -n "all up front (default): "; run
echo -n "pairs split:          "; CFR_PACK_SPLIT=1 run --mode pe
echo -n "pairs all up front:   "; run --mode pe
