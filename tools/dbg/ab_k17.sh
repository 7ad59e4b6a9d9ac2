#!/bin/bash
# cfg2 with the derived K-mer table at K = 16 (16-byte entries with text positions: shipped), K = 16 with 8-byte entries, K = 17 with 8-byte entries (137 GB)
export CFR_DEBUG_ENV=1 CFR_LOAD_TIMING=1
run() { python bench.py "$@" --no-cpu-baseline --no-pmc --no-extra-configs --steps 6 --warmup 2 2>/tmp/k17.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g reads/s  search %.2f ms  total %.2f ms  oracle %s' % (d['value'], d['stage_ms']['search_ms'], d['stage_ms']['total_ms'], d['parity'].get('equals_oracle')))" || tail -3 /tmp/k17.err; grep -E "device image" /tmp/k17.err | cut -c1-120; }
for rep in 1 2; do
  echo "== K = 16, 16-byte entries (shipped)"; run
  echo "== K = 16, 8-byte entries"; CFR_FTABX_E8=1 run
  echo "== K = 17, 8-byte entries"; CFR_FTABX_WIDTH=17 CFR_FTABX_E8=1 run
done
echo "== pairs, shipped / K = 17"; run --mode pe; CFR_FTABX_WIDTH=17 CFR_FTABX_E8=1 run --mode pe
echo "== iteration mix K = 17"; CFR_FTABX_WIDTH=17 CFR_FTABX_E8=1 CFR_SEARCH_PROF=1 python bench.py --no-cpu-baseline --no-pmc --no-extra-configs --steps 1 --warmup 0 2>&1 | grep "search prof" | head -1
