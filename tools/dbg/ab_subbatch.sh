#!/bin/bash
# sub-batch size with the post stage beside the next search (cfg2, pairs)
run() { python bench.py "$@" --no-cpu-baseline --no-pmc --no-extra-configs --steps 8 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g reads/s  step %.2f ms  search %.2f  device window %.2f' % (d['value'], d['ms_per_step'], d['stage_ms']['search_ms'], d['stage_ms']['total_ms']))"; }
export CFR_DEBUG_ENV=1
for sb in 500000 833334 1250000 2000000 2500000; do
  echo -n "sub-batch $sb: "; CFR_SUBBATCH=$sb run
done
echo -n "1250000, no taper: "; CFR_TAPER_FLOOR=0 run
echo -n "pairs 625000:  "; CFR_SUBBATCH=625000 run --mode pe
echo -n "pairs 1250000: "; run --mode pe
echo -n "pairs 2500000: "; CFR_SUBBATCH=2500000 run --mode pe
